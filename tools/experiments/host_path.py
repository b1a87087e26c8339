import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from uf3_amd import synthetic, _lib
from uf3_amd.representation import process
basis = synthetic.notebook_basis(['Mo', 'W'])
frames = [synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, [42, 74], 3000 + k) for k in range(8)]
fz = process.BasisFeaturizer(basis)
ctx, db = fz._dev()
batch = _lib.FrameBatch(frames)
F = db.n_feat
def run(x_e, x_f, tag):
    t0 = time.perf_counter()
    ctx.check(ctx.lib.uf3_featurize(db.handle, C.byref(batch.struct), _lib._p(batch.pos), _lib._p(batch.z), _lib._p(x_e), _lib._p(x_f)))
    dt = time.perf_counter() - t0
    print(f"{tag}: {dt*1e3:.1f} ms = {8/dt:.1f} frames/s ({x_f.nbytes/dt/1e9:.1f} GB/s)")
xe = np.empty((8, F)); xf = np.empty((batch.n_atoms, 3, F))
run(xe, xf, "pageable fresh"); run(xe, xf, "pageable reused")
t0 = time.perf_counter()
pf = torch.empty((batch.n_atoms, 3, F), dtype=torch.float64, pin_memory=True); pe = torch.empty((8, F), dtype=torch.float64, pin_memory=True)
print(f"pinned alloc {1e3*(time.perf_counter()-t0):.1f} ms")
run(pe.numpy(), pf.numpy(), "pinned fresh"); run(pe.numpy(), pf.numpy(), "pinned reused")
