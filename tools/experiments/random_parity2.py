"""Harder randomised parity sweep: up to 5 species, batches of ragged frames, slab / wire / cluster boundary
conditions, leading trims 0..3, per-trio resolutions (asymmetric l / m legs on mixed trios), pair cut-offs shorter than
the 3-body range, 2-body-only bases.   python tools/experiments/random_parity2.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from uf3_amd.data import composition
from uf3_amd.data.atoms import Atoms
from uf3_amd.representation import bspline, process

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
Z = {'Al': 13, 'Cu': 29, 'Mo': 42, 'W': 74, 'Zr': 40, 'Ni': 28, 'Ti': 22}
worst, modes_seen = 0.0, {}
for case in range(n_cases):
    S = int(rng.integers(1, 6))
    els = sorted(rng.choice(list(Z), S, replace=False).tolist())
    degree = 2 if rng.random() < 0.15 else 3
    cs = composition.ChemicalSystem(els, degree)
    pairs = cs.interactions_map[2]
    trios = cs.interactions_map[3] if degree == 3 else []
    r3 = float(rng.uniform(3.0, 4.0))
    rmin, rmax, res = {}, {}, {}
    for p in pairs:
        rmin[p] = float(rng.uniform(0.2, 1.0)); rmax[p] = float(rng.uniform(3.2, 6.0)); res[p] = int(rng.integers(5, 18))
    sym_res = [int(rng.integers(4, 9)), int(rng.integers(8, 18))]
    for t in trios:
        rmin[t] = [float(rng.uniform(0.8, 1.6))] * 3
        rmax[t] = [r3, r3, 2 * r3]
        if t[1] != t[2] and rng.random() < 0.5:
            res[t] = [int(rng.integers(4, 9)), int(rng.integers(4, 9)), int(rng.integers(8, 18))]
        else:
            res[t] = [sym_res[0], sym_res[0], sym_res[1]]
    basis = bspline.BSplineBasis(cs, r_min_map=rmin, r_max_map=rmax, resolution_map=res,
                                 leading_trim={2: int(rng.choice([0, 0, 1])), 3: int(rng.integers(0, 4))},
                                 trailing_trim={2: 3, 3: int(rng.choice([3, 3, 2]))})
    frames = []
    for _ in range(int(rng.integers(1, 4))):
        reps = tuple(int(x) for x in rng.integers(int(os.environ.get("REPS_LO", 3)), int(os.environ.get("REPS_HI", 6)), 3))
        a = float(rng.uniform(2.9, 3.4))
        grid = np.array([[i, j, k] for i in range(reps[0]) for j in range(reps[1]) for k in range(reps[2])], float)
        frac = (grid[:, None, :] + np.array([[0, 0, 0], [.5, .5, .5]])[None]).reshape(-1, 3)
        keep = rng.random(len(frac)) > 0.1                                      # vacancies: ragged neighbour counts
        frac = frac[keep]
        cell = np.diag(np.array(reps, float) * a) @ (np.eye(3) + rng.normal(0, 0.03, (3, 3)))
        pos = (frac / np.array(reps)) @ cell + rng.normal(0, 0.1, (len(frac), 3))
        pos += rng.integers(-2, 3, 3) @ cell if rng.random() < 0.3 else 0.0     # atoms outside the box
        pbc = [True, True, True] if rng.random() < 0.5 else [bool(b) for b in rng.integers(0, 2, 3)]
        frames.append(Atoms(numbers=rng.choice([Z[e] for e in els], len(pos)), positions=pos, cell=cell, pbc=pbc))
    fz = process.BasisFeaturizer(basis)
    modes = fz._dev()[1].featurizer_modes
    modes_seen[modes] = modes_seen.get(modes, 0) + 1
    try:
        x_e, x_f, off = fz.featurize_frames(frames)
    except Exception as exc:
        print(f"case {case:3d}: S={S} degree={degree} F={basis.n_feats} modes={modes:#x} ERROR {exc}", flush=True)
        continue
    err = 0.0
    ob = O.OracleBasis(basis)
    for k, fr in enumerate(frames):
        ref = O.featurize(ob, fr)
        err = max(err, np.abs(x_e[k] - ref["xe"]).max() / max(1.0, np.abs(ref["xe"]).max()),
                  np.abs(x_f[off[k]:off[k + 1]] - ref["xf"]).max() / max(1.0, np.abs(ref["xf"]).max()))
    worst = max(worst, err)
    print(f"case {case:3d}: S={S} degree={degree} F={basis.n_feats} frames={[len(f) for f in frames]} modes={modes:#x} rows {err:.1e}"
          + ("" if err < 1e-9 else "   <-- MISMATCH"), flush=True)
print("worst", worst, "modes", {hex(k): v for k, v in modes_seen.items()})
