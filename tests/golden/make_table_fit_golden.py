#!/usr/bin/env python3
"""
Golden vectors for the from-file fit workflow (build container only).  The reference's ``fit_from_file``
(uf3/regression/least_squares.py:355-424) needs PyTables to open its HDF5 input, which this image lacks; its
arithmetic is the per-table call of ``gram_from_df`` (:435-483) plus the accumulation written out below in the same
order, so the reference's own ``gram_from_df``, ``VarianceRecorder``, ``calc_E_F_weights``,
``combine_weighted_gram``, ``fit_with_gram`` and ``subset_prediction`` are run here on three seeded feature tables
(W / Mo, 2-body basis, layout of ``BasisFeaturizer.evaluate``).  Writes tests/golden/table_fit.npz.
"""
import json
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, "/root/reference")
warnings.simplefilter("ignore")
from uf3.data import composition as rc  # noqa: E402
from uf3.representation import bspline as rb  # noqa: E402
from uf3.regression import least_squares as rl  # noqa: E402

els = ['Mo', 'W']
kw = dict(r_min_map={('Mo', 'Mo'): 0.5, ('Mo', 'W'): 0.6, ('W', 'W'): 0.7},
          r_max_map={('Mo', 'Mo'): 5.0, ('Mo', 'W'): 5.5, ('W', 'W'): 6.0},
          resolution_map={('Mo', 'Mo'): 6, ('Mo', 'W'): 8, ('W', 'W'): 7})
basis = rb.BSplineBasis(rc.ChemicalSystem(els, 2), **kw)
nf = int(np.sum(basis.partition_sizes))
columns = basis.get_column_names()                      # y | n_Mo n_W | pair columns
assert len(columns) == nf + 1
rng = np.random.default_rng(2024)
c_true = rng.normal(0, 1, nf)
c_true[basis.col_idx] = 0.0


def make_table(names, sizes):
    index, rows = [], []
    for name, n in zip(names, sizes):
        n_mo = int(rng.integers(0, n + 1))
        x = np.concatenate([[n_mo, n - n_mo], rng.uniform(0, 4, nf - 2) * n])
        index.append((name, "energy"))
        rows.append(np.concatenate([[x @ c_true + rng.normal(0, 1e-2)], x]))
        for c in "xyz":
            for i in range(n):
                x = np.concatenate([[0.0, 0.0], rng.normal(0, 1, nf - 2)])
                index.append((name, f"f{c}_{i}"))
                rows.append(np.concatenate([[x @ c_true + rng.normal(0, 1e-2)], x]))
    return pd.DataFrame(np.array(rows), index=pd.MultiIndex.from_tuples(index), columns=columns)


layout = [(["a0", "a1", "a2"], [4, 7, 2]), (["b0", "b1"], [5, 3]), (["c0", "c1", "c2", "c3"], [2, 6, 3, 4])]
tables = [make_table(*spec) for spec in layout]
subset = ["a0", "a2", "b1", "c0", "c1", "c3", "not_there"]
sample_weights = {"a2": 0.25, "c1": 3.0}
reg = basis.get_regularization_matrix(ridge_1b=1e-6, ridge_2b=1e-5, curvature_2b=1e-4)
out = dict(meta=json.dumps(dict(element_list=els, degree=2,
                                basis_kwargs={k: {"-".join(p): v for p, v in m.items()} for k, m in kw.items()})),
           regularizer=reg, subset=np.array(subset), weight_names=np.array(list(sample_weights)),
           weight_values=np.array(list(sample_weights.values())), kappa=np.array([0.35]))
for t, df in enumerate(tables):
    out[f"table{t}"] = df.values
    out[f"table{t}_names"] = np.array([i[0] for i in df.index])
    out[f"table{t}_keys"] = np.array([i[1] for i in df.index])

model = rl.WeightedLinearModel(basis, regularizer=reg)
gram_e, gram_f, ord_e, ord_f = model.initialize_gram_ordinate()
e_var, f_var = rl.VarianceRecorder(), rl.VarianceRecorder()
for t, df in enumerate(tables):                              # the loop of fit_from_file, :391-412
    keys = df.index.unique(level=0).intersection(subset)
    g_e, g_f, o_e, o_f = model.gram_from_df(df, keys, e_variance=e_var, f_variance=f_var,
                                            sample_weights=sample_weights, energy_key="energy", batch_size=7)
    out.update({f"gram_e{t}": g_e, f"gram_f{t}": g_f, f"ord_e{t}": o_e, f"ord_f{t}": o_f})
    gram_e += g_e
    gram_f += g_f
    ord_e += o_e
    ord_f += o_f
w_e, w_f = rl.calc_E_F_weights(e_var.n, f_var.n, e_var.std, f_var.std)
gram, ordinate = model.combine_weighted_gram(gram_e, gram_f, ord_e, ord_f, w_e, w_f, 0.35)
model.fit_with_gram(gram, ordinate)
out.update(e_stats=np.array([e_var.mean, e_var.std, e_var.n]), f_stats=np.array([f_var.mean, f_var.std, f_var.n]),
           weights=np.array([w_e, w_f]), coefficients=model.coefficients, data_coverage=model.data_coverage)

n_el = len(els)
parts = ([], [], [], [])
for df in tables:                                            # the loop of batched_prediction, :998-1014
    for dst, piece in zip(parts, rl.subset_prediction(df, model, subset_keys=["a1", "b0", "c2", "c3"], n_elements=n_el)):
        dst.append(piece)
for name, p in zip(("pred_y_e", "pred_p_e", "pred_y_f", "pred_p_f"), parts):
    out[name] = np.concatenate(p)

# VarianceRecorder.update_with_components (:55-67) on a table with fx / fy / fz columns, one row holding NaN
vr = rl.VarianceRecorder()
fdf = pd.DataFrame({"fx": [[0.1, -0.2], [0.3], np.nan], "fy": [[0.0, 0.5], [-0.7], [1.0]], "fz": [[0.2, 0.2], [0.9], [2.0]]})
vr.update_with_components(fdf)
vr.update_with_components(pd.DataFrame({"fx": [[1.5, -1.0, 0.2]], "fy": [[0.4, 0.1, 0.0]], "fz": [[-0.3, 0.8, 0.6]]}))
out["components_stats"] = np.array([vr.mean, vr.std, vr.n])
np.savez_compressed(os.path.join(HERE, "table_fit.npz"), **out)
print({k: np.shape(v) for k, v in out.items()})
