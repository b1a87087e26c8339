"""N1 both directions: model JSON written here == what the reference's loader reads (captured by
tests/golden/make_json_golden.py in the build container), and this loader reads the same files identically."""
import json
import os
import sys

import numpy as np

from _util import GOLDEN

sys.path.insert(0, GOLDEN)
import make_json_golden as gen  # noqa: E402  (the generator's model definitions, without the reference part)

from uf3_amd.regression import least_squares as ls  # noqa: E402


def test_builder_json_is_what_the_reference_loaded(tmp_path):
    seen = np.load(os.path.join(GOLDEN, "json_roundtrip.npz"))
    texts = json.load(open(os.path.join(GOLDEN, "json_roundtrip_texts.json")))
    for name, model in gen.builder_models().items():
        path = tmp_path / (name + ".json")
        model.to_json(str(path))
        # the file written today carries the same content as the one the reference read (JSON-level comparison:
        # key order and float formatting are the serialiser's business)
        assert json.loads(open(path).read()) == json.loads(texts[name]), name
        # the reference saw the builder's coefficients / columns / frozen columns / coverage / knots
        assert np.array_equal(seen[name + "/coefficients"], model.coefficients), name
        assert list(seen[name + "/columns"]) == list(model.bspline_config.get_column_names()), name
        assert np.array_equal(seen[name + "/col_idx"], np.asarray(model.col_idx)), name
        assert np.array_equal(seen[name + "/coverage"], np.asarray(model.data_coverage, dtype=bool)), name
        # this loader on the same file: identical model, identical predictions to the reference's
        back = ls.WeightedLinearModel.from_json(str(path))
        assert np.array_equal(back.coefficients, model.coefficients), name
        x = np.random.default_rng(3).random((5, len(model.coefficients)))
        assert np.allclose(back.predict(x), seen[name + "/predict"], rtol=1e-14, atol=0), name
