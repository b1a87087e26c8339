"""Host-side helpers around a fitted model (repulsive-core post-processing of pair coefficients, the Taylor continuation
behind WeightedLinearModel.fix_repulsion_2b, the row-weight form of the training tuples) against captures of the reference's
functions (tests/golden/make_postprocess_golden.py; reference least_squares.py:623-663, 1075-1144, process.py:508-616)."""
import contextlib
import io
import os
import types

import numpy as np
import pandas as pd
import pytest

from uf3_amd.data import composition
from uf3_amd.regression import least_squares as ls
from uf3_amd.representation import bspline, process
from _util import GOLDEN

G = np.load(os.path.join(GOLDEN, "postprocess.npz"), allow_pickle=False)


def test_well_search_and_core_postprocessing():
    shapes = G["shapes"]
    for rf in (3, 2):
        assert [ls.find_pair_potential_well(c, rf) for c in shapes] == G[f"well_rf{rf}"].tolist()
    settings = [dict(), dict(core_hardness=3.0, min_core=1.0, min_slope=0.05, rounding_factor=2), dict(smooth_cutoff=True),
                dict(min_core=10.0, rounding_factor=4)]
    assert len(settings) == int(G["n_settings"])
    for k, kw in enumerate(settings):
        got = np.array([ls.postprocess_coefficients_2b(c.copy(), **kw) for c in shapes])
        assert np.array_equal(got, G[f"post{k}"]), k
    keep = shapes[0].copy()
    untouched = keep.copy()
    out = ls.postprocess_coefficients_2b(keep)
    assert out is not keep and np.array_equal(keep, untouched)
    assert ls.postprocess_coefficients_2b(keep, in_place=True) is keep and bool(G["in_place_is_same_object"])


def test_taylor_continuation_and_repulsion_fix():
    knots, coeff, r = G["knots"], G["taylor_coeff"], G["taylor_r"]
    for k, rt in enumerate(G["taylor_targets"]):
        for tag, mc in (("c2", 2.0), ("none", None), ("c0", 0.0)):
            got = ls.get_spline_taylor_expansion(float(rt), r, coeff, knots, min_curvature=mc)
            assert np.allclose(got, G[f"taylor_{k}_{tag}"], rtol=1e-11, atol=1e-12), (k, tag)
    cs = composition.ChemicalSystem(["W"], 2)
    basis = bspline.BSplineBasis(cs, r_min_map={("W", "W"): 0.5}, r_max_map={("W", "W"): 5.5}, resolution_map={("W", "W"): 15},
                                 leading_trim=0, trailing_trim=3)
    assert np.array_equal(basis.knots_map[("W", "W")], knots)
    model = ls.WeightedLinearModel(basis)
    for tag, kw in (("default", {}), ("target", dict(r_target=2.2, min_curvature=0.5))):
        model.coefficients = G["fix_input"].copy()
        model.data_coverage = G["fix_coverage"].copy()
        with contextlib.redirect_stdout(io.StringIO()) as said:
            model.fix_repulsion_2b(("W", "W"), **kw)
        assert "adjusted 5 coefficients" in said.getvalue()
        assert np.allclose(model.coefficients, G[f"fix_{tag}"], rtol=1e-11, atol=1e-12), tag
        assert np.array_equal(model.coefficients[6:], G["fix_input"][6:])         # covered coefficients stay


def test_second_derivative_of_the_basis_elements():
    """nu = 2 of BasisFunction (the Taylor continuation needs it) against central differences of nu = 1, clamped ends included"""
    knots = G["knots"]
    x = np.linspace(knots[0] + 1e-3, knots[-1] - 1e-3, 301)
    h = 1e-6
    for i in range(len(knots) - 4):
        f = bspline.BasisFunction(knots[i:i + 5])
        inside = (x - h > knots[i]) & (x + h < knots[i + 4])
        near_knot = np.min(np.abs(x[:, None] - knots[None, :]), axis=1) < 2 * h
        ok = inside & ~near_knot
        num = (f(x + h, nu=1) - f(x - h, nu=1)) / (2 * h)
        assert np.allclose(f(x, nu=2)[ok], num[ok], rtol=1e-5, atol=1e-5), i
    with pytest.raises(NotImplementedError):
        bspline.BasisFunction(knots[:5])(x, nu=3)


def test_row_weight_training_tuples():
    index = pd.MultiIndex.from_tuples(list(zip(G["tt_names"].tolist(), G["tt_keys"].tolist())))
    df = pd.DataFrame(G["tt_table"], index=index)
    for k, kappa in enumerate(G["tt_kappas"]):
        x, y, w = process.dataframe_to_training_tuples(df, kappa=float(kappa), energy_key="energy")
        assert np.array_equal(x, G[f"tt_x{k}"]) and np.array_equal(y, G[f"tt_y{k}"])
        assert np.allclose(w, G[f"tt_w{k}"], rtol=1e-14, atol=0)
    with pytest.raises(ValueError):
        process.dataframe_to_training_tuples(df, kappa=1.5)
    with pytest.raises(ValueError):
        process.dataframe_to_training_tuples(df.iloc[:1])
    fz = process.BasisFeaturizer.__new__(process.BasisFeaturizer)      # (the method needs no device: only the coordinator's key)
    with pytest.warns(DeprecationWarning):
        x2, y2, w2 = fz.get_training_tuples(df, 0.5, types.SimpleNamespace(energy_key="energy"))
    assert np.array_equal(w2, process.dataframe_to_training_tuples(df, 0.5)[2])
