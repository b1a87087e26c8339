import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from uf3_amd import _lib, synthetic
from uf3_amd.data import composition
from uf3_amd.representation import process, bspline
dev = torch.device("cuda", 0)
def basis_res(res3):
    cs = composition.ChemicalSystem(['Mo', 'W'], 3)
    pairs, trios = cs.interactions_map[2], cs.interactions_map[3]
    return bspline.BSplineBasis(cs, r_min_map={**{p: 0.001 for p in pairs}, **{t: [1.5, 1.5, 1.5] for t in trios}},
        r_max_map={**{p: 5.5 for p in pairs}, **{t: [3.5, 3.5, 7.0] for t in trios}},
        resolution_map={**{p: 15 for p in pairs}, **{t: res3 for t in trios}}, leading_trim={2: 0, 3: 3}, trailing_trim={2: 3, 3: 3})
for res3 in ([7, 7, 14], [8, 8, 16], [9, 9, 19]):
    basis = basis_res(res3)
    frames = [synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, [42, 74], 3000 + k) for k in range(2)]
    batch = _lib.FrameBatch(frames)
    fz = process.BasisFeaturizer(basis, device=0)
    ctx, db = fz._dev()
    F = db.n_feat
    d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
    d_xe = torch.empty((2, F), dtype=torch.float64, device=dev)
    d_xf = torch.empty((batch.n_atoms, 3, F), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    step = lambda: fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), d_xe.data_ptr(), d_xf.data_ptr())
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"res={res3}: F={F}, modes={db.featurizer_modes:#x}, {dt*1e3:.2f} ms per 2 frames, {2/dt:.0f} frames/s")
