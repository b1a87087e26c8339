"""Randomised parity sweep: random bases (species, resolutions, trims, cut-offs) x random small cells (pbc patterns,
densities) through the featurizer and the evaluator against the oracle.  Prints the worst relative error and the
featurizer modes seen.   python tools/experiments/random_parity.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from uf3_amd.data import composition
from uf3_amd.data.atoms import Atoms
from uf3_amd.representation import bspline, process
from uf3_amd.regression import least_squares as ls
from uf3_amd.forcefield import calculator

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ELS = ['Al', 'Cu', 'Mo', 'W', 'Zr']
worst, modes_seen = 0.0, {}
for case in range(n_cases):
    S = int(rng.integers(1, 4))
    els = sorted(rng.choice(ELS, S, replace=False).tolist())
    cs = composition.ChemicalSystem(els, 3)
    pairs, trios = cs.interactions_map[2], cs.interactions_map[3]
    r3 = float(rng.uniform(3.0, 4.2))
    res_l = int(rng.integers(4, 10)); res_n = int(rng.integers(8, 20))
    lead3 = int(rng.choice([0, 3])); trail3 = int(rng.choice([3, 3, 2]))
    kw = dict(r_min_map={**{p: float(rng.uniform(0.2, 1.0)) for p in pairs}, **{t: [float(rng.uniform(0.8, 1.6))] * 3 for t in trios}},
              r_max_map={**{p: float(rng.uniform(4.0, 6.0)) for p in pairs}, **{t: [r3, r3, 2 * r3] for t in trios}},
              resolution_map={**{p: int(rng.integers(6, 18)) for p in pairs}, **{t: [res_l, res_l, res_n] for t in trios}},
              leading_trim={2: 0, 3: lead3}, trailing_trim={2: 3, 3: trail3})
    basis = bspline.BSplineBasis(cs, **kw)
    # cell: bcc-like lattice with random strain, rattle, pbc pattern; big enough for the reference's supercell logic
    reps = tuple(int(x) for x in rng.integers(3, 5, 3))
    a = float(rng.uniform(2.9, 3.4))
    base = np.array([[0, 0, 0], [.5, .5, .5]])
    grid = np.array([[i, j, k] for i in range(reps[0]) for j in range(reps[1]) for k in range(reps[2])], float)
    frac = (grid[:, None, :] + base[None, :, :]).reshape(-1, 3)
    cell = np.diag(np.array(reps, float) * a) @ (np.eye(3) + rng.normal(0, 0.02, (3, 3)))
    pos = (frac / np.array(reps)) @ cell + rng.normal(0, 0.1, (len(frac), 3))
    from uf3_amd.data.composition import ChemicalSystem  # noqa
    numbers = rng.choice([{'Al': 13, 'Cu': 29, 'Mo': 42, 'W': 74, 'Zr': 40}[e] for e in els], len(pos))
    pbc = [True, True, True] if rng.random() < 0.7 else [bool(b) for b in rng.integers(0, 2, 3)]
    atoms = Atoms(numbers=numbers, positions=pos, cell=cell, pbc=pbc)
    fz = process.BasisFeaturizer(basis)
    modes = fz._dev()[1].featurizer_modes
    modes_seen[modes] = modes_seen.get(modes, 0) + 1
    try:
        x_e, x_f, _ = fz.featurize_frames([atoms])
    except Exception as exc:
        print(f"case {case:3d}: S={S} res=({res_l},{res_l},{res_n}) lead3={lead3} trail3={trail3} N={len(pos)} F={basis.n_feats} modes={modes:#x} ERROR {exc}", flush=True)
        continue
    ref = O.featurize(O.OracleBasis(basis), atoms)
    err_e = np.abs(x_e[0] - ref["xe"]).max() / max(1.0, np.abs(ref["xe"]).max())
    err_f = np.abs(x_f - ref["xf"]).max() / max(1.0, np.abs(ref["xf"]).max())
    coeff = rng.normal(0, 0.05, basis.n_feats); coeff[basis.col_idx] = 0.0
    model = ls.WeightedLinearModel(basis); model.coefficients = coeff
    e, f, _ = calculator.UFCalculator(model).evaluate_frames([atoms])
    e_ref, f_ref = O.evaluate(O.OracleBasis(basis), atoms, coeff)
    err_v = max(abs(e[0] - e_ref) / max(1.0, abs(e_ref)), np.abs(f - f_ref).max() / max(1.0, np.abs(f_ref).max()))
    worst = max(worst, err_e, err_f, err_v)
    flag = "" if max(err_e, err_f, err_v) < 1e-9 else "   <-- MISMATCH"
    print(f"case {case:3d}: S={S} res=({res_l},{res_l},{res_n}) lead3={lead3} trail3={trail3} N={len(pos)} pbc={pbc} modes={modes:#x} "
          f"rows {max(err_e, err_f):.1e} eval {err_v:.1e}{flag}", flush=True)
print("worst", worst, "modes", {hex(k): v for k, v in modes_seen.items()})
