#!/usr/bin/env python3
"""
Model-JSON round trip through the REFERENCE's loader (build container only, like make_golden.py).

The builder's ``WeightedLinearModel.to_json`` writes models with fitted-looking coefficients (unary 2-body, binary
2-body, unary 2+3-body with trims, binary 2+3-body); the reference's ``WeightedLinearModel.from_json``
(uf3/regression/least_squares.py:200-216, 528-621; uf3/util/json_io.py:11-83) loads each file, and what IT sees --
coefficients, column names, frozen columns, data coverage, the knots of every interaction -- is stored in
tests/golden/json_roundtrip.npz next to the JSON texts.  tests/test_json_roundtrip.py (CPU) checks that the files
the builder writes today are the committed ones and that its own loader agrees with what the reference saw.
"""
import json
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
warnings.simplefilter("ignore")


def builder_models():
    sys.path.insert(0, ROOT)
    from uf3_amd.data import composition
    from uf3_amd.representation import bspline
    from uf3_amd.regression import least_squares as ls
    cases = {}
    specs = {
        "unary_2b": dict(elements=['W'], degree=2, kw=dict(r_min_map={('W', 'W'): 1.5}, r_max_map={('W', 'W'): 5.5},
                                                        resolution_map={('W', 'W'): 12})),
        "binary_2b": dict(elements=['Ne', 'Xe'], degree=2, kw=dict()),
        "unary_3b": dict(elements=['W'], degree=3, kw=dict(
            r_min_map={('W', 'W'): 0.001, ('W', 'W', 'W'): [1.5, 1.5, 1.5]},
            r_max_map={('W', 'W'): 5.5, ('W', 'W', 'W'): [3.5, 3.5, 7.0]},
            resolution_map={('W', 'W'): 15, ('W', 'W', 'W'): [6, 6, 12]},
            leading_trim={2: 0, 3: 3}, trailing_trim={2: 3, 3: 3})),
        "binary_3b": dict(elements=['Mo', 'W'], degree=3, kw=dict(leading_trim={2: 0, 3: 0}, trailing_trim={2: 3, 3: 3})),
    }
    for idx, (name, sp) in enumerate(specs.items()):
        cs = composition.ChemicalSystem(sp["elements"], sp["degree"])
        basis = bspline.BSplineBasis(cs, **sp["kw"])
        model = ls.WeightedLinearModel(basis)
        rng = np.random.default_rng(100 + idx)
        c = rng.normal(0, 1, basis.n_feats)
        c[basis.col_idx] = 0.0
        model.coefficients = c
        model.data_coverage = rng.random(basis.n_feats) > 0.3
        cases[name] = model
    return cases


def main():
    models = builder_models()
    texts = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, model in models.items():
            path = os.path.join(tmp, name + ".json")
            model.to_json(path)
            texts[name] = open(path).read()
        # ---- the reference reads them -------------------------------------------------------------
        sys.path.insert(0, os.path.join(HERE, "_standins"))
        sys.path.insert(0, REF)
        from uf3.regression import least_squares as ref_ls
        out = {}
        for name in models:
            ref = ref_ls.WeightedLinearModel.from_json(os.path.join(tmp, name + ".json"))
            b = ref.bspline_config
            out[name + "/coefficients"] = np.asarray(ref.coefficients, dtype=float)
            out[name + "/columns"] = np.array(b.get_column_names())
            out[name + "/frozen"] = np.asarray(b.frozen_c if hasattr(b, "frozen_c") else ref.frozen_c, dtype=float)
            out[name + "/col_idx"] = np.asarray(ref.col_idx, dtype=int)
            out[name + "/coverage"] = np.asarray(ref.data_coverage, dtype=bool)
            knots = []
            for key in list(b.interactions_map[2]) + list(b.interactions_map.get(3, [])):
                k = b.knots_map[key]
                knots.append(np.concatenate([np.ravel(x) for x in (k if isinstance(k, (list, tuple)) else [k])]))
            out[name + "/knots"] = np.concatenate(knots)
            # and the reference's own predictions from the loaded coefficients on a fixed random design matrix
            rng = np.random.default_rng(3)
            x = rng.random((5, len(ref.coefficients)))
            out[name + "/predict"] = np.asarray(ref.predict(x), dtype=float)
            print(name, len(ref.coefficients), "columns loaded by the reference")
    np.savez_compressed(os.path.join(HERE, "json_roundtrip.npz"), **out)
    json.dump(texts, open(os.path.join(HERE, "json_roundtrip_texts.json"), "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
