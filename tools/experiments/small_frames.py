import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from uf3_amd import _lib, synthetic
from uf3_amd.representation import process
dev = torch.device("cuda", 0)
for (reps, nfr, els) in [((4, 4, 4), 512, ['W']), ((8, 8, 8), 64, ['W']), ((4, 4, 4), 512, ['Mo', 'W'])]:
    basis = synthetic.notebook_basis(els)
    zs = [74] if els == ['W'] else [42, 74]
    frames = [synthetic.lattice_frame("bcc", reps, 3.165, zs, 100 + k) for k in range(nfr)]
    batch = _lib.FrameBatch(frames)
    fz = process.BasisFeaturizer(basis, device=0)
    ctx, db = fz._dev()
    F = db.n_feat
    d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
    d_xe = torch.empty((nfr, F), dtype=torch.float64, device=dev)
    d_xf = torch.empty((batch.n_atoms, 3, F), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    step = lambda: fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), d_xe.data_ptr(), d_xf.data_ptr())
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{els} {len(frames[0])} atoms x {nfr} frames, F={F}: {dt*1e3:.2f} ms/step, {nfr/dt:.0f} frames/s, {batch.n_atoms/dt/1e6:.2f} M atoms/s")
