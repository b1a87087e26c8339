import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from uf3_amd import synthetic, _lib
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls
basis = synthetic.notebook_basis(['W'])
model = ls.WeightedLinearModel(basis)
coeff = np.random.default_rng(1).normal(0, 0.05, basis.n_feats); coeff[basis.col_idx] = 0.0
model.coefficients = coeff
calc = calculator.UFCalculator(model)
atoms = synthetic.lattice_frame("bcc", (4, 4, 4), 3.165, [74], seed=3)
for _ in range(50): calc.evaluate_frames([atoms])
N = 500
t0 = time.perf_counter()
for _ in range(N): calc.evaluate_frames([atoms])
t_all = (time.perf_counter() - t0) / N
ctx = _lib.get_context(None); db = _lib.device_basis(basis, ctx)
batch = _lib.FrameBatch([atoms]); e = np.empty(1); f = np.empty((batch.n_atoms, 3)); addr = _lib._addr
args = (db.handle, C.byref(batch.struct), addr(batch.pos), addr(batch.z), calc._pc[0], calc._pc[1], calc._pc[2], addr(e), addr(f))
t0 = time.perf_counter()
for _ in range(N): ctx.lib.uf3_eval(*args)
t_c = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N): _lib.FrameBatch([atoms])
t_fb = (time.perf_counter() - t0) / N
print(f"evaluate_frames {t_all*1e6:.1f} us; C entry alone {t_c*1e6:.1f} us; FrameBatch {t_fb*1e6:.1f} us")
