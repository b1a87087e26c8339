"""Featurizer rate on the 10k-atom W/Mo cells for 3-body leading trims 0 / 1 / 2 / 3 (windows of 6 / 5 / 4 / 3 rows on the
centre legs: k_featurize3<6,3> / <5,3> / <4,2> / <3,1>) -- what a window narrowed to the rows a batch touches would run at."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from uf3_amd import synthetic, _lib
from uf3_amd.representation import process
dev = torch.device("cuda:0")
B = 24
frames = [synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, [42, 74], 3000 + k) for k in range(B)]
batch = _lib.FrameBatch(frames)
d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
for lead in (0, 1, 2, 3):
    basis = synthetic.notebook_basis(['Mo', 'W'], lead3=lead)
    fz = process.BasisFeaturizer(basis, device=0)
    ctx, db = fz._dev()
    F = db.n_feat
    xe = torch.empty((B, F), dtype=torch.float64, device=dev)
    xf = torch.empty((batch.n_atoms, 3, F), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    def step(): fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), xe.data_ptr(), xf.data_ptr())
    for _ in range(3): step()
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    ctx.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"lead {lead}: F = {F}, {B / dt:.0f} frames/s, {dt * 1e3:.2f} ms per {B} frames")
    del xf, xe
