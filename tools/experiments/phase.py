"""Experiment: per-phase cycle counters of the MFMA featurizer (library built with -DUF3_PHASE_TIMING)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["UF3_LIB_PATH"] = os.path.join(ROOT, "exp", "libuf3hip_phase.so")
import torch
from uf3_amd import _lib, synthetic
from uf3_amd.representation import process
dev = torch.device("cuda", 0)
basis = synthetic.notebook_basis(['Mo', 'W'], lead3=0 if 'lead0' in sys.argv else 3)
frames = [synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, [42, 74], 3000 + k) for k in range(8)]
batch = _lib.FrameBatch(frames)
fz = process.BasisFeaturizer(basis, device=0)
ctx, db = fz._dev()
F = db.n_feat
d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
d_xe = torch.empty((8, F), dtype=torch.float64, device=dev)
d_xf = torch.empty((batch.n_atoms, 3, F), dtype=torch.float64, device=dev)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
step = lambda: fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), d_xe.data_ptr(), d_xf.data_ptr())
step(); step()
buf = (ctypes.c_ulonglong * 16)()
ctx.lib.uf3_debug_phase(buf)
step()
ctx.lib.uf3_debug_phase(buf)
names = ["list", "setup", "walk", "eval", "stage", "mfma", "fold: energy adds + end", "fold: tile dump", "fold: reads + sums", "fold: row stores (incl. their completion)"]
tot = sum(buf[i] for i in range(len(names)))
for i, n in enumerate(names):
    print(f"{n:44s} {buf[i]:>14d} {100.0 * buf[i] / tot:5.1f}%")
