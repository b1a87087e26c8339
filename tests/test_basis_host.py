"""Host-side basis layer against captures of the reference's BSplineBasis (exact equality)."""
import json
import os

import numpy as np
import pytest

from uf3_amd.data import composition
from uf3_amd.representation import bspline
from uf3_amd.regression import regularize
from _util import GOLDEN, basis_from_meta

HOST = json.load(open(os.path.join(GOLDEN, "host_basis.json")))
ARR = np.load(os.path.join(GOLDEN, "host_basis.npz"))


def jkey(t):
    return "-".join(t) if isinstance(t, tuple) else t


@pytest.mark.parametrize("name", sorted(HOST))
def test_basis_matches_reference(name):
    ref = HOST[name]
    b = basis_from_meta(ref)
    assert [jkey(i) for i in b.interactions] == ref["interactions"]
    assert [int(x) for x in b.partition_sizes] == ref["partition_sizes"]
    assert b.r_cut == ref["r_cut"]
    assert b.get_column_names() == ref["columns"]
    assert b.col_idx.tolist() == ref["col_idx"]
    assert {jkey(k): int(v) for k, v in b.symmetry.items()} == ref["symmetry"]
    for d in range(2, b.degree + 1):
        assert [int(h) for h in b.chemical_system.interaction_hashes[d]] == ref["hashes"][str(d)]
    for k, v in b.knots_map.items():
        mine = np.array(v if not isinstance(v, list) else np.concatenate(v))
        assert np.array_equal(mine, ARR[f"{name}|knots|{jkey(k)}"])
    for k in b.symmetry:
        assert np.array_equal(b.template_mask[k], ARR[f"{name}|mask|{jkey(k)}"])
        assert np.array_equal(b.flat_weights[k], ARR[f"{name}|weights|{jkey(k)}"])
    key = f"{name}|regularizer"
    if key in ARR:
        reg = b.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8,
                                          curvature_2b=1e-8, curvature_3b=1e-6)
        assert np.array_equal(reg, ARR[key])


@pytest.mark.parametrize("name", [n for n in sorted(HOST) if HOST[n]["degree"] == 3])
def test_column_sources_reproduce_compress(name):
    b = basis_from_meta(HOST[name])
    rng = np.random.default_rng(0)
    for trio in b.interactions_map[3]:
        grid = rng.random(b.templates[trio].shape)
        lut, w = b.column_sources(trio)
        got = np.zeros(len(b.template_mask[trio]))
        hit = lut >= 0
        np.add.at(got, lut[hit], (grid.ravel() * w)[hit])
        assert np.allclose(got, b.compress_3B(grid, trio), rtol=1e-14, atol=0)
        assert np.allclose(w[hit], 1.0, rtol=0, atol=4e-16)
        vec = rng.random(len(b.template_mask[trio]))
        assert np.allclose(b.compress_3B(b.decompress_3B(vec, trio), trio, fitting=False), vec)


def test_bspline_values_against_scipy_capture():
    d = np.load(os.path.join(GOLDEN, "bspline_values.npz"))
    t, x, ref = d["knots"], d["x"], d["values"]
    for b in range(len(t) - 4):
        el = bspline.BasisFunction(t[b:b + 5])
        for nu in (0, 1):
            assert np.allclose(el(x, nu=nu), ref[nu, b], rtol=1e-12, atol=1e-13)


def test_known_answer_values():
    # tests/test_bsplines.py:529-547 of the reference: clamped knots [0,0,0,0,1,1,1,1]
    t = np.array([0, 0, 0, 0, 1, 1, 1, 1], dtype=float)
    first, v, _ = bspline.basis_values(t, np.array([0.25, 0.5, 0.75]))
    assert np.all(first == 0)
    bern = lambda x: np.array([(1 - x) ** 3, 3 * x * (1 - x) ** 2, 3 * x ** 2 * (1 - x), x ** 3])  # noqa: E731
    for row, x in zip(v, (0.25, 0.5, 0.75)):
        assert np.allclose(row, bern(x), rtol=1e-14)


def test_knot_generators_and_spline_indices():
    k = bspline.generate_uniform_knots(1.0, 6.0, 5)
    assert np.allclose(k, [1, 1, 1, 1, 2, 3, 4, 5, 6, 6, 6, 6])
    k = bspline.generate_lammps_knots(1.0, 6.0, 5)
    assert np.allclose(k[3:-3] ** 2, np.linspace(1, 36, 6))
    pts, idx = bspline.find_spline_indices(np.array([1.5, 5.5]), bspline.generate_uniform_knots(1, 6, 5))
    assert idx.tolist() == [0, 1, 2, 3, 4, 5, 6, 7] and pts.tolist() == [1.5] * 4 + [5.5] * 4


def test_composition_order_and_trios():
    cs = composition.ChemicalSystem(['Xe', 'Ne', 'Ne'], degree=3)
    assert cs.element_list == ('Ne', 'Xe')
    assert cs.interactions_map[2] == [('Ne', 'Ne'), ('Ne', 'Xe'), ('Xe', 'Xe')]
    assert cs.interactions_map[3] == [('Ne', 'Ne', 'Ne'), ('Ne', 'Ne', 'Xe'), ('Ne', 'Xe', 'Xe'),
                                      ('Xe', 'Ne', 'Ne'), ('Xe', 'Ne', 'Xe'), ('Xe', 'Xe', 'Xe')]


def test_regularizer_shapes():
    m = regularize.get_curvature_penalty_matrix_1D(5)
    assert m[0, 0] == -1 and m[4, 4] == -1 and m[2, 2] == -2 and m[2, 1] == 1
    m3 = regularize.get_curvature_penalty_matrix_3D(2, 3, 4)
    assert m3.shape == (24, 24) and np.all(m3.sum(axis=1) == 0)
    full = regularize.combine_regularizer_matrices([np.eye(2), np.ones((3, 4))])
    assert full.shape == (5, 6) and full[2:, 2:].sum() == 12


def test_szudzik_hash_helpers_against_reference_values():
    """literal values produced by the reference's composition.py (:252-376) in the build container"""
    from uf3_amd.data import composition as cp
    assert cp.symbols_to_hash(['W', 'W']) == 5624 and cp.symbols_to_hash(['Mo', 'W']) == 5592
    assert cp.symbols_to_hash(['W', 'Mo', 'W']) == 30448398 and cp.symbols_to_hash(['H', 'O', 'O']) == 5337
    assert cp.hash_to_symbols(5592, 2) == ('Mo', 'W') and cp.hash_to_symbols(30448398, 3) == ('W', 'Mo', 'W')
    triples = np.array([[1, 8, 8], [74, 42, 74], [6, 1, 1]])
    h = cp.get_szudzik_hash(triples)
    assert h.tolist() == [5337, 30448398, 1370]
    assert np.array_equal(cp.unpack_szudzik_hash(h, 3), triples.astype(float))
    assert cp.szudzik_unpair(np.array([0, 1, 2, 3, 4, 5, 8, 9, 5550])).tolist() == [
        [0, 0], [1, 0], [0, 1], [1, 1], [2, 0], [2, 1], [2, 2], [3, 0], [0, 74]]
    rng = np.random.default_rng(0)
    pairs = rng.integers(0, 119, (500, 2))
    assert np.array_equal(cp.szudzik_unpair(cp.szudzik_pair(pairs)), pairs.astype(float))
    gathered = cp.hash_gather(np.array([1.0, 2.0, 3.0, 4.0]), np.array([7, 3, 7, 3]))
    assert list(gathered) == [3, 7] and gathered[7].tolist() == [1.0, 3.0] and gathered[3].tolist() == [2.0, 4.0]
