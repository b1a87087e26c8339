"""Timeline of one small evaluator call (under `rocprofv3 --kernel-trace --memory-copy-trace`): 300 calls on a 128-atom W frame."""
import sys, time
import numpy as np
from uf3_amd import synthetic
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
basis = synthetic.notebook_basis(['W'])
model = ls.WeightedLinearModel(basis)
coeff = np.random.default_rng(1).normal(0, 0.05, basis.n_feats)
coeff[basis.col_idx] = 0.0
model.coefficients = coeff
calc = calculator.UFCalculator(model)
atoms = synthetic.lattice_frame("bcc", (n, n, n), 3.165, [74], seed=3)
for _ in range(20):
    calc.evaluate_frames([atoms])
t0 = time.perf_counter()
for _ in range(300):
    calc.evaluate_frames([atoms])
print(f"{len(atoms)} atoms: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call")
