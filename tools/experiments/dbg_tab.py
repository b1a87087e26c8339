import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from uf3_amd import synthetic
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls
from oracle import oracle as O
basis = synthetic.notebook_basis(['Mo', 'W'])
model = ls.WeightedLinearModel(basis)
coeff = np.random.default_rng(5).normal(0, 0.05, basis.n_feats); coeff[basis.col_idx] = 0.0
model.coefficients = coeff
calc = calculator.UFCalculator(model)
atoms = synthetic.lattice_frame("bcc", (7, 6, 5), 3.165, [42, 74], seed=2)
for _ in range(3): e, f, _, v = calc.evaluate_frames([atoms], virial=True)
def run(**env):
    os.environ.update(env)
    try: return calc.evaluate_frames([atoms], virial=True)
    finally:
        for k in env: del os.environ[k]
eo, fo = O.evaluate(O.OracleBasis(basis), atoms, coeff)
for name, env in (("gather", dict(UF3_EVAL_GATHER="1")), ("notab", dict(UF3_EVAL_NO_TAB="1")), ("again", {})):
    e2, f2, _, v2 = run(**env)
    print(name, "e eq", e2[0] == e[0], "v eq", np.array_equal(v2, v), "f maxdiff", np.abs(f2 - f).max(), "v maxrel", np.abs(v2 - v).max() / np.abs(v).max(), "e diff", e2[0] - e[0])
print("oracle: e", abs(e[0] - eo) / abs(eo), "f", np.abs(f - fo).max() / np.abs(fo).max())
