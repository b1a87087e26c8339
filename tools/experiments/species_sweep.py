"""Featurizer throughput on 10k-atom frames with 1 / 2 / 3 species (notebook basis): same cells and records per atom."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np, torch
from uf3_amd import _lib, synthetic
from uf3_amd.representation import process
for els, zs in ((['W'], [74]), (['Mo', 'W'], [42, 74]), (['V', 'Mo', 'W'], [23, 42, 74])):
    basis = synthetic.notebook_basis(els)
    B = 16
    frames = [synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, zs, 3000 + k) for k in range(B)]
    batch = _lib.FrameBatch(frames)
    fz = process.BasisFeaturizer(basis, device=0)
    ctx, db = fz._dev()
    F = db.n_feat
    dev = torch.device("cuda", 0)
    d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
    d_xe = torch.empty((B, F), dtype=torch.float64, device=dev)
    d_xf = torch.empty((batch.n_atoms, 3, F), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    step = lambda: fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), d_xe.data_ptr(), d_xf.data_ptr())
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(els, "F", F, "ms/step", round(dt * 1e3, 3), "frames/s", round(B / dt, 1), "modes", hex(db.featurizer_modes))
    del d_xf
