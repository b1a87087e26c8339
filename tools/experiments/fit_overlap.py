"""Experiment: X^T X of batch i on a second stream while batch i + 1 is featurized (two row buffers).
   python tools/experiments/fit_overlap.py [c4|w] [frames]"""
import ctypes as C, sys, time
import numpy as np, torch
from uf3_amd import _lib, synthetic, pipeline
from uf3_amd.regression import least_squares as ls
from uf3_amd.representation import process
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
els, zs = ((['Mo', 'W'], [42, 74]) if wl == "c4" else (['W'], [74]))
basis = synthetic.notebook_basis(els)
frames = [synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, zs, 3000 + k) for k in range(B)]
batch = _lib.FrameBatch(frames)
fz = process.BasisFeaturizer(basis, device=0)
ctx, db = fz._dev()
F = db.n_feat
model = ls.WeightedLinearModel(basis, regularizer=basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0))
acc = pipeline.DeviceFitAccumulator(model, fz, with_forces=True)
d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
d_counts = torch.from_numpy(np.diff(batch.offsets).astype(np.float64)).to(dev)
g = torch.Generator(device=dev).manual_seed(7)
d_ye = torch.randn((B,), dtype=torch.float64, device=dev, generator=g)
d_yf = torch.randn((3 * batch.n_atoms,), dtype=torch.float64, device=dev, generator=g)
bufs = [(torch.empty((B, F), dtype=torch.float64, device=dev), torch.empty((batch.n_atoms * 3, F), dtype=torch.float64, device=dev)) for _ in range(2)]
sA = torch.cuda.current_stream(dev); sB = torch.cuda.Stream(dev)
free = [torch.cuda.Event(), torch.cuda.Event()]
for e in free: e.record(sB)

def serial(k):
    ctx.set_stream(sA.cuda_stream)
    acc.add_device_batch(batch.struct, B, batch.n_atoms, d_pos, d_z, d_counts, d_ye, d_yf, x_e=bufs[0][0], x_f=bufs[0][1])

def overlapped(k):
    xe, xf = bufs[k % 2]
    sA.wait_event(free[k % 2])
    ctx.set_stream(sA.cuda_stream)
    fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), xe.data_ptr(), xf.data_ptr())
    ready = torch.cuda.Event(); ready.record(sA)
    sB.wait_event(ready)
    ctx.set_stream(sB.cuda_stream)
    n_fro = int(acc._frozen.numel())
    ctx.check(ctx.lib.uf3_fit_rows_dev(ctx.handle, B, F, xe.data_ptr(), d_counts.data_ptr(), d_ye.data_ptr(), d_yf.data_ptr(), int(d_yf.numel()),
                                       acc._frozen.data_ptr() if n_fro else None, acc._frozen_c.data_ptr() if n_fro else None, n_fro, acc.m_e.data_ptr()))
    acc._gram(xe, d_ye, acc.gram_e, acc.ord_e)
    acc._gram(xf, d_yf, acc.gram_f, acc.ord_f)
    free[k % 2].record(sB)
    ctx.set_stream(sA.cuda_stream)

for name, fn in (("serial", serial), ("overlapped", overlapped)):
    for k in range(3): fn(k)
    torch.cuda.synchronize()
    acc.reset()
    n = 8
    t0 = time.perf_counter()
    for k in range(n): fn(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{wl} {name}: {dt * 1e3:.2f} ms per {B}-frame step = {B / dt:.0f} frames/s; G_f checksum {float(acc.gram_f.sum()):.10e}")
    acc.reset()
