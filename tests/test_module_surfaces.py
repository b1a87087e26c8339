"""
Module-level surfaces ``uf3_amd.representation.distances`` / ``angles`` against captures of
the reference's own free functions (tests/golden/make_surface_golden.py -> surface_<case>.npz).

Distances and index lists: bit-exact (they are selections of correctly rounded roots in the reference's order).
Direction cosines: IEEE quotients of the same operands -- bit-exact on the explicit-index form, 1e-12 through the
list route (image offsets are added on the device in its own order).  Feature grids: 1e-9 of each entry.
"""
import json
import os

import numpy as np
import pytest

from uf3_amd.data import geometry
from uf3_amd.data.atoms import Atoms
from uf3_amd.representation import angles, distances
from _util import GOLDEN, basis_from_meta, worst_elementwise

CASES = ["case_steel", "case_w16", "case_nexe32"]


def _load(case):
    d = np.load(os.path.join(GOLDEN, f"surface_{case}.npz"))
    meta = json.loads(str(d["meta"]))
    src = np.load(os.path.join(GOLDEN, case + ".npz"))
    atoms = Atoms(numbers=src["numbers"], positions=src["positions"], cell=src["cell"], pbc=src["pbc"])
    basis = basis_from_meta(meta)
    sup = geometry.get_supercell(atoms, r_cut=float(d["r_cut_sup"][0]))
    return d, basis, atoms, sup


# ------------------------------------------------------------------------------ host-only pieces (CPU suite)
def test_host_helpers_match_reference_semantics():
    d, basis, atoms, sup = _load("case_steel")
    # lattice_images: the reference's tiling is recognised, in its order; a copy of the frame has no images
    shifts = distances.lattice_images(atoms, sup)
    assert shifts is not None and shifts is not False
    assert np.array_equal(shifts, geometry.image_shifts(np.array(atoms.get_cell()), atoms.get_pbc(), basis.r_cut))
    assert distances.lattice_images(atoms, None) is None and distances.lattice_images(atoms, atoms.copy()) is None
    odd = sup.copy()
    odd.positions[len(atoms) + 1] += 0.01
    assert distances.lattice_images(atoms, odd) is False
    # masks and Kronecker tables
    gz, sz = atoms.get_atomic_numbers(), sup.get_atomic_numbers()
    m = distances.mask_matrix_by_pair_interaction((6, 26), gz, sz)
    assert m.shape == (len(atoms), len(sup)) and m[0, 8] and not m[0, 1] and m[8, 0]
    k = distances.kronecker_delta(np.arange(3, dtype=np.int32), np.array([0, 5]), np.array([2, 1]))
    assert np.array_equal(k, [[-1, 0], [0, 1], [1, 0]])
    assert np.array_equal(distances.kronecker_vectorized(3, np.array([0, 5]), np.array([2, 1])), k)
    # group_idx_by_center, coefficient counts, symmetrize_3B against the capture
    iv, groups = angles.group_idx_by_center(d["ij_i"], d["ij_j"])
    assert np.array_equal(iv, np.unique(d["ij_i"])) and sum(len(g) for g in groups) == len(d["ij_j"])
    assert np.array_equal(groups[0], d["ij_j"][d["ij_i"] == iv[0]])
    ks = [basis.knots_map[t] for t in basis.interactions_map[3]]
    L, M, N = angles.coefficient_counts_from_knots(ks)
    assert (L[0], M[0], N[0]) == d["energy_grid0"].shape
    assert np.allclose(angles.symmetrize_3B(d["energy_grid0"], int(d["sym0"][0])), d["symmetrized0"], rtol=1e-15, atol=0)
    sq = d["energy_grid0"][:min(L[0], M[0]), :min(L[0], M[0])]
    assert np.allclose(angles.symmetrize_3B(sq, 2), d["symmetrized0_s2"], rtol=1e-15, atol=0)
    assert np.array_equal(angles.symmetrize_3B(sq, 1), sq)


# ------------------------------------------------------------------------------ device
@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_distances_module_against_reference_capture(case):
    d, basis, atoms, sup = _load(case)
    pairs = basis.interactions_map[2]
    got = distances.distances_by_interaction(atoms, pairs, basis.r_min_map, basis.r_max_map, supercell=sup)
    got_at = distances.distances_by_interaction(atoms, pairs, basis.r_min_map, basis.r_max_map, supercell=sup, atomic=True)
    got_cl = distances.distances_by_interaction(atoms, pairs, basis.r_min_map, basis.r_max_map)
    dist, drij = distances.derivatives_by_interaction(atoms, pairs, basis.r_cut, basis.r_min_map, basis.r_max_map,
                                                      supercell=sup)
    for p, pair in enumerate(pairs):
        assert np.array_equal(got[pair], d[f"dist{p}"])                      # values AND order
        assert np.array_equal(got_cl[pair], d[f"dist_cluster{p}"])
        assert np.array_equal([len(v) for v in got_at[pair]], d[f"dist_atomic_len{p}"])
        assert np.array_equal(np.concatenate(got_at[pair]) if got_at[pair] else np.zeros(0), d[f"dist_atomic_cat{p}"])
        assert np.array_equal(dist[pair], d[f"deriv_dist{p}"])
        ref = d[f"deriv_drij{p}"]
        assert drij[pair].shape == ref.shape
        assert np.array_equal(drij[pair] != 0, ref != 0)                      # the same sparsity: same (i, j) per column
        assert np.abs(drij[pair] - ref).max(initial=0.0) < 1e-12
    # the dense route (an arbitrary atom set: the supercell with one ghost removed) agrees with the list route's values
    keep = np.ones(len(sup), dtype=bool)
    keep[len(atoms) + 3] = False
    odd = Atoms(numbers=sup.get_atomic_numbers()[keep], positions=sup.get_positions()[keep])
    dense = distances.distances_by_interaction(atoms, pairs, basis.r_min_map, basis.r_max_map, supercell=odd)
    dm = distances.get_distance_matrix(atoms, odd)
    for p, pair in enumerate(pairs):
        zz = distances.mask_matrix_by_pair_interaction(
            [int(z) for z in distances.composition.symbols2numbers(pair)], atoms.get_atomic_numbers(), odd.get_atomic_numbers())
        want = dm[zz & (dm > max(basis.r_min_map[pair], 0)) & (dm < basis.r_max_map[pair])]
        assert np.array_equal(dense[pair], want)
    # distance matrix: scipy's roots bit for bit
    assert np.array_equal(distances.get_distance_matrix(atoms, sup), d["dm_geom_sup"])
    # compute_direction_cosines on explicit indices: IEEE quotients, bit for bit
    dss = distances.get_distance_matrix(sup, sup)
    dc = distances.compute_direction_cosines(sup.get_positions(), dss, d["dc_i"], d["dc_j"], len(atoms))
    assert np.array_equal(dc, d["dc_out"])
    # legacy unary helpers are consistent with the matrix
    flat = distances.distances_from_geometry(atoms, sup, r_min=0.5, r_max=3.0)
    assert np.array_equal(flat, d["dm_geom_sup"][(d["dm_geom_sup"] > 0.5) & (d["dm_geom_sup"] < 3.0)])
    masked = distances.mask_supercell_with_radius(atoms, sup, 2.5)
    assert len(masked) == int(np.any(d["dm_geom_sup"] <= 2.5, axis=0).sum())
    dd, dr = distances.get_distance_derivatives(atoms, sup, r_min=0.5, r_max=2.5)
    assert dr.shape == (len(atoms), 3, len(dd)) and np.all((dd > 0.5) & (dd <= 2.5))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_angles_module_against_reference_capture(case):
    d, basis, atoms, sup = _load(case)
    trios = basis.interactions_map[3]
    ks = [basis.knots_map[t] for t in trios]
    hashes = basis.chemical_system.interaction_hashes[3]
    assert np.array_equal(np.asarray(hashes, dtype=np.int64), d["hashes3"])
    n_lead, n_trail = (int(x) for x in d["trims3"])
    m, iw, jw = angles.identify_ij(atoms, ks, sup)
    assert m.shape == (len(sup), len(sup))
    assert np.array_equal(iw, d["ij_i"]) and np.array_equal(jw, d["ij_j"])
    pos, m2, iw2, jw2 = angles.identify_ij(atoms, ks, sup, square=True)
    assert np.array_equal(pos, sup.get_positions())
    assert np.array_equal(iw2, d["ij_sq_i"]) and np.array_equal(jw2, d["ij_sq_j"])
    assert np.array_equal(m2[:len(atoms)], d["ij_sq_matrix_rows"])
    eg = angles.featurize_energy_3b(atoms, ks, None, hashes, supercell=sup, n_lead=n_lead, n_trail=n_trail)
    eg0 = angles.featurize_energy_3b(atoms, ks, None, hashes, supercell=sup)
    ec = angles.featurize_energy_3b(atoms, ks, None, hashes, n_lead=n_lead, n_trail=n_trail)
    fg = angles.featurize_force_3b(atoms, ks, None, hashes, supercell=sup, n_lead=n_lead, n_trail=n_trail)
    assert len(eg) == len(trios) == len(fg)
    for t in range(len(trios)):
        for got, key in ((eg[t], f"energy_grid{t}"), (eg0[t], f"energy_grid_notrim{t}"), (ec[t], f"energy_grid_cluster{t}")):
            assert got.shape == d[key].shape
            assert worst_elementwise(got, d[key]) <= 1.0, key
        ref = d[f"force_grid{t}"]
        got = np.array([[np.asarray(c) for c in comps] for comps in fg[t]])
        assert got.shape == ref.shape
        assert worst_elementwise(got, ref) <= 1.0
    # a supercell built with a larger cut-off (get_supercell's default r_cut = 10) gives the same grids
    big = geometry.get_supercell(atoms, r_cut=basis.r_cut + 3.0)
    eb = angles.featurize_energy_3b(atoms, ks, None, hashes, supercell=big, n_lead=n_lead, n_trail=n_trail)
    for t in range(len(trios)):
        assert worst_elementwise(eb[t], d[f"energy_grid{t}"]) <= 1.0
    # an atom set that is not a tiling is refused
    odd = sup.copy()
    odd.positions[len(atoms) + 1] += 0.01
    with pytest.raises(ValueError):
        angles.featurize_energy_3b(atoms, ks, None, hashes, supercell=odd)
