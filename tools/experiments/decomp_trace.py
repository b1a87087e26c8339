"""One rank's share of a decomposed 50 000-atom frame (uf3_eval_centres_dev), for kernel traces: python decomp_trace.py [world]"""
import ctypes as C, sys, time
import numpy as np, torch
from uf3_amd import _lib, synthetic
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
basis = synthetic.notebook_basis(['V', 'Mo', 'W'])
atoms = synthetic.lattice_frame("bcc", (25, 25, 40), 3.165, [23, 42, 74], 4000)
model = ls.WeightedLinearModel(basis)
coeff = np.random.default_rng(11).normal(0, 0.05, basis.n_feats); coeff[basis.col_idx] = 0.0
model.coefficients = coeff
calc = calculator.UFCalculator(model)
ctx = _lib.get_context(0); ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
db = _lib.device_basis(basis, ctx)
batch = _lib.FrameBatch([atoms])
d_pos = torch.from_numpy(batch.pos).to(dev); d_z = torch.from_numpy(batch.z).to(dev)
d_e = torch.empty((1,), dtype=torch.float64, device=dev); d_f = torch.empty((batch.n_atoms, 3), dtype=torch.float64, device=dev)
d_v = torch.empty((6,), dtype=torch.float64, device=dev)
lo, hi = 0, (batch.n_atoms + world - 1) // world
call = lambda: ctx.check(ctx.lib.uf3_eval_centres_dev(db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()), C.c_void_p(d_z.data_ptr()),
        _lib._p(calc._c1), _lib._p(calc._c2), _lib._p(calc._c3), lo, hi, C.c_void_p(d_e.data_ptr()), C.c_void_p(d_f.data_ptr()), C.c_void_p(d_v.data_ptr())))
for _ in range(5): call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): call()
torch.cuda.synchronize()
print(f"world {world}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per share")
