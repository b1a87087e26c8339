import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from uf3_amd import synthetic, _lib
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls
from uf3_amd.data.atoms import Atoms
basis = synthetic.notebook_basis(['W'])
model = ls.WeightedLinearModel(basis)
coeff = np.random.default_rng(1).normal(0, 0.05, basis.n_feats); coeff[basis.col_idx] = 0.0
model.coefficients = coeff
calc = calculator.UFCalculator(model, md_skin=0.5)
atoms = synthetic.lattice_frame("bcc", (4, 4, 4), 3.165, [74], seed=3)
rng = np.random.default_rng(0)
noise = rng.uniform(-0.01, 0.01, (512,) + atoms.positions.shape)
ctx = _lib.get_context(None)
for _ in range(50): calc.evaluate_frames([atoms])
ts = []
b0 = ctx.md_stats()["builds"]
builds_at = []
for i in range(3000):
    t0 = time.perf_counter()
    atoms.positions += noise[i & 511]
    calc.evaluate_frames([atoms])
    ts.append(time.perf_counter() - t0)
    b = ctx.md_stats()["builds"]
    if b != b0: builds_at.append(i); b0 = b
ts = np.array(ts) * 1e6
mask = np.zeros(len(ts), bool); mask[builds_at] = True
print(f"mean {ts.mean():.1f} us; median {np.median(ts):.1f}; rebuild calls {mask.sum()} mean {ts[mask].mean():.1f} us; others mean {ts[~mask].mean():.1f}; p99 {np.percentile(ts, 99):.1f}")
nxt = np.zeros(len(ts), bool); nxt[[i + 1 for i in builds_at if i + 1 < len(ts)]] = True
print(f"calls right after a rebuild: {ts[nxt].mean():.1f} us; calls before a rebuild (soft flag): {ts[[i - 1 for i in builds_at if i > 0]].mean():.1f}")
