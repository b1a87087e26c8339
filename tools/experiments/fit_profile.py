import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from uf3_amd import _lib, synthetic, pipeline
from uf3_amd.regression import least_squares as ls
from uf3_amd.representation import process
basis = synthetic.notebook_basis(['W'])
frames = [synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, [74], 5000 + k) for k in range(8)]
rng = np.random.default_rng(3)
energies = rng.normal(-8.9e4, 5.0, 8); forces = [rng.normal(0, 0.5, (len(f), 3)) for f in frames]
model = ls.WeightedLinearModel(basis); fz = process.BasisFeaturizer(basis, device=0)
acc = pipeline.DeviceFitAccumulator(model, fz)
acc.add_frames(frames, energies, forces)
torch.cuda.synchronize()
dev = acc.dev
T = {}
def lap(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t0; return time.perf_counter()
for rep in range(3):
    t = time.perf_counter()
    batch = _lib.FrameBatch(frames); t = lap("pack", t)
    d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev); t = lap("h2d", t)
    x_e = torch.empty((8, acc.n_feat), dtype=torch.float64, device=dev); x_f = torch.empty((batch.n_atoms * 3, acc.n_feat), dtype=torch.float64, device=dev); t = lap("alloc", t)
    fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), x_e.data_ptr(), x_f.data_ptr()); t = lap("featurize", t)
    n_atoms = x_e[:, :acc.n_el].sum(dim=1); x_e2 = (x_e / n_atoms[:, None]).contiguous()
    y_e = torch.from_numpy(np.asarray(energies)).to(dev) / n_atoms; t = lap("norm", t)
    acc._gram(x_e2, y_e, acc.gram_e, acc.ord_e); t = lap("gram_e", t)
    m = ls.moments(y_e.cpu().numpy()); t = lap("moments_e", t)
    y_host = np.concatenate([np.asarray(f).reshape(-1, 3) for f in forces]).reshape(-1); t = lap("y_pack", t)
    y_f = torch.from_numpy(y_host).to(dev); t = lap("y_h2d", t)
    acc._gram(x_f, y_f, acc.gram_f, acc.ord_f); t = lap("gram_f", t)
    m = ls.moments(y_host); t = lap("moments_f", t)
for k, v in T.items(): print(f"{k:10s} {v/3*1e3:7.3f} ms per 8-frame chunk")
print("total", sum(T.values())/3*1e3)
