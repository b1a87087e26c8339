#!/usr/bin/env python3
"""
Golden vectors for the DataFrame plumbing of the fit (build container only): the reference's own
``dataframe_to_tuples`` / ``subset_prediction`` (uf3/regression/least_squares.py:666-713, 933-962) on a seeded
feature table.  Writes tests/golden/dataframe_tuples.npz (table + expected tuples for four argument combinations).
"""
import os
import sys
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, "/root/reference")
warnings.simplefilter("ignore")
from uf3.regression import least_squares as rl  # noqa: E402

rng = np.random.default_rng(77)
names = ["frame_a", "frame_b", "frame_c"]
sizes = [3, 2, 4]
index, rows = [], []
n_el, n_feat = 2, 9
for name, n in zip(names, sizes):
    comp = rng.integers(1, 4, n_el).astype(float)
    comp[0] += n - comp.sum() if comp.sum() < n else 0.0
    index.append((name, "energy"))
    rows.append(np.concatenate([[rng.normal(-5 * n, 1.0)], comp, rng.uniform(0, 3, n_feat - n_el)]))
    for c in "xyz":
        for i in range(n):
            index.append((name, f"f{c}_{i}"))
            rows.append(np.concatenate([[rng.normal(0, 0.5)], np.zeros(n_el), rng.normal(0, 1, n_feat - n_el)]))
table = np.array(rows)
columns = ["y"] + [f"c{k}" for k in range(n_feat)]
df = pd.DataFrame(table, index=pd.MultiIndex.from_tuples(index), columns=columns)
out = dict(table=table, index_names=np.array([i[0] for i in index]), index_keys=np.array([i[1] for i in index]))
weights = {"frame_a": 0.5, "frame_c": 2.0}
for tag, kw in [("plain", {}), ("norm", dict(n_elements=n_el)), ("weighted", dict(sample_weights=weights)),
                ("norm_weighted", dict(n_elements=n_el, sample_weights=weights))]:
    x_e, y_e, x_f, y_f = rl.dataframe_to_tuples(df, **kw)
    out.update({f"{tag}_x_e": x_e, f"{tag}_y_e": y_e, f"{tag}_x_f": x_f, f"{tag}_y_f": y_f})


class _Model:                       # subset_prediction only calls predict()
    def __init__(self, c):
        self.c = c

    def predict(self, x):
        return np.dot(x, self.c)


coeff = rng.normal(0, 1, n_feat)
y_e, p_e, y_f, p_f = rl.subset_prediction(df, _Model(coeff), subset_keys=["frame_c", "frame_a", "missing"], n_elements=n_el)
out.update(coeff=coeff, sub_y_e=y_e, sub_p_e=p_e, sub_y_f=y_f, sub_p_f=p_f)
np.savez(os.path.join(HERE, "dataframe_tuples.npz"), **out)
print({k: np.shape(v) for k, v in out.items()})
