#!/usr/bin/env python3
"""
Golden capture of the reference's host-side helpers around a fitted model that the mirrored classes lacked until round 5
(runs only in the build container, like make_golden.py: same stand-ins, reference at /root/reference):

    python tests/golden/make_postprocess_golden.py

    uf3.regression.least_squares.find_pair_potential_well, postprocess_coefficients_2b, get_spline_taylor_expansion,
    WeightedLinearModel.fix_repulsion_2b; uf3.representation.process.dataframe_to_training_tuples

on seeded coefficient vectors / a seeded feature table.  Outputs: tests/golden/postprocess.npz.
"""
import contextlib
import io
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, REF)
warnings.simplefilter("ignore")

from uf3.data import composition as rc  # noqa: E402
from uf3.regression import least_squares as rls  # noqa: E402
from uf3.representation import bspline as rb, process as rp  # noqa: E402

out = {}
rng = np.random.default_rng(2024)
# ---- pair-potential shapes: a well with a core, a well left of the peak, a plateau, nearly flat, monotone --------------
r = np.linspace(0.0, 1.0, 18)
shapes = [
    6.0 * np.exp(-6 * r) - 1.5 * np.exp(-((r - 0.45) / 0.15) ** 2),
    np.concatenate([[0.02, 0.01, 0.015, 0.4, 1.2, 0.8], -0.6 * np.exp(-((r[6:] - 0.6) / 0.2) ** 2)]),
    np.concatenate([np.full(5, 1e-4), 0.3 * np.sin(6 * r[5:])]),
    1e-5 * rng.normal(size=18),
    np.linspace(3.0, -0.2, 18),
    rng.normal(0, 0.5, 18),
    np.concatenate([[-0.5, -0.2, 0.1], 2.0 * np.exp(-4 * r[3:]) - 0.8]),
]
out["shapes"] = np.array(shapes)
for rf in (3, 2):
    out[f"well_rf{rf}"] = np.array([rls.find_pair_potential_well(np.array(c), rf) for c in shapes])
settings = [dict(), dict(core_hardness=3.0, min_core=1.0, min_slope=0.05, rounding_factor=2), dict(smooth_cutoff=True),
            dict(min_core=10.0, rounding_factor=4)]
out["n_settings"] = np.array(len(settings))
for k, kw in enumerate(settings):
    out[f"post{k}"] = np.array([rls.postprocess_coefficients_2b(np.array(c), **kw) for c in shapes])
keep = np.array(shapes[0])
same = rls.postprocess_coefficients_2b(keep, in_place=True)
out["in_place_is_same_object"] = np.array(same is keep)

# ---- Taylor continuation of a pair spline towards the core, and the model method that uses it --------------------------
cs = rc.ChemicalSystem(["W"], 2)
basis = rb.BSplineBasis(cs, r_min_map={("W", "W"): 0.5}, r_max_map={("W", "W"): 5.5}, resolution_map={("W", "W"): 15},
                        leading_trim=0, trailing_trim=3)
knots = basis.knots_map[("W", "W")]
coeff = 4.0 * np.exp(-1.2 * np.linspace(0, 4, 18)) - 0.9 * np.exp(-((np.linspace(0, 4, 18) - 2.0) / 0.6) ** 2)
out["knots"] = np.array(knots)
out["taylor_coeff"] = coeff
targets = np.array([1.9, 2.4, 3.1])
out["taylor_targets"] = targets
out["taylor_r"] = np.array(knots[2:8])
for k, rt in enumerate(targets):
    for tag, mc in (("c2", 2.0), ("none", None), ("c0", 0.0)):
        out[f"taylor_{k}_{tag}"] = np.array(rls.get_spline_taylor_expansion(rt, out["taylor_r"], coeff, knots, min_curvature=mc), dtype=float)
model = rls.WeightedLinearModel(basis)
flat = np.concatenate([[-3.25], coeff])
cover = np.ones(len(flat), dtype=bool)
cover[1:1 + 5] = False                       # the first five pair functions saw no data
for tag, kw in (("default", {}), ("target", dict(r_target=2.2, min_curvature=0.5))):
    model.coefficients = flat.copy()
    model.data_coverage = cover.copy()
    with contextlib.redirect_stdout(io.StringIO()):
        model.fix_repulsion_2b(("W", "W"), **kw)
    out[f"fix_{tag}"] = np.array(model.coefficients)
out["fix_input"] = flat
out["fix_coverage"] = cover

# ---- the deprecated weights-per-row form of the training tuples -----------------------------------------------------------
n_e, n_f, n_x = 7, 23, 9
names = [f"s{k}" for k in range(n_e)]
idx, rows = [], []
for k, nm in enumerate(names):
    idx.append((nm, "energy")); rows.append(np.concatenate([[rng.normal(-20, 3)], rng.normal(size=n_x)]))
    for j in range(3 + (k % 2)):
        idx.append((nm, f"fx_{j}")); rows.append(np.concatenate([[rng.normal(0, 0.7)], rng.normal(size=n_x)]))
df = pd.DataFrame(np.array(rows), index=pd.MultiIndex.from_tuples(idx), columns=["y"] + [f"x{k}" for k in range(n_x)])
out["tt_table"] = df.to_numpy()
out["tt_names"] = np.array([i[0] for i in idx])
out["tt_keys"] = np.array([i[1] for i in idx])
for k, kappa in enumerate((0.5, 0.0, 0.85)):
    x, y, w = rp.dataframe_to_training_tuples(df, kappa=kappa, energy_key="energy")
    out[f"tt_x{k}"], out[f"tt_y{k}"], out[f"tt_w{k}"] = x, y, w
out["tt_kappas"] = np.array([0.5, 0.0, 0.85])
np.savez_compressed(os.path.join(HERE, "postprocess.npz"), **out)
print("wrote postprocess.npz:", len(out), "arrays")
