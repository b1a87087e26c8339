"""Does the featurizer's rate depend on the order of the atoms in the input?  (lattice order vs a random permutation)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from uf3_amd import _lib, synthetic
from uf3_amd.data.atoms import Atoms
from uf3_amd.representation import process
dev = torch.device("cuda", 0)
basis = synthetic.notebook_basis(['Mo', 'W'])
fz = process.BasisFeaturizer(basis, device=0)
ctx, db = fz._dev()
F = db.n_feat
B = 32
for mode in ("lattice order", "shuffled"):
    frames = []
    for k in range(B):
        a = synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, [42, 74], 3000 + k)
        if mode == "shuffled":
            p = np.random.default_rng(k).permutation(len(a))
            a = Atoms(numbers=a.get_atomic_numbers()[p], positions=a.get_positions()[p], cell=np.array(a.get_cell()), pbc=True)
        frames.append(a)
    batch = _lib.FrameBatch(frames)
    d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
    d_xe = torch.empty((B, F), dtype=torch.float64, device=dev)
    d_xf = torch.empty((batch.n_atoms, 3, F), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    step = lambda: fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), d_xe.data_ptr(), d_xf.data_ptr())
    for _ in range(3): step()
    torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f"{mode}: {B / dt:.0f} frames/s ({dt * 1e3:.2f} ms per {B} frames); row checksum {float(d_xf.abs().sum()):.6e}")
