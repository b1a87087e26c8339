"""
GPU parity: the HIP path (through the C ABI, via the product classes) against the golden
vectors captured from the reference and against the oracle on seeded inputs.

Tolerances: neighbour indices bit-exact; feature rows / energies / forces 1e-9 relative
(north_star demands 1e-6; observed ~1e-14).
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from uf3_amd import synthetic, _lib
from uf3_amd.data.atoms import Atoms, read_extxyz
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls
from uf3_amd.representation import process
from _util import GOLDEN, FEATURE_CASES, basis_from_meta, load_case, rel_err, worst_elementwise

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.mark.parametrize("name", FEATURE_CASES)
def test_feature_rows_against_reference_capture(name):
    d, meta, atoms = load_case(name)
    basis = basis_from_meta(meta)
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, _ = fz.featurize_frames([atoms], energy=True, forces="xf" in d)
    assert rel_err(x_e[0], d["xe"]) < TOL
    assert worst_elementwise(x_e[0], d["xe"]) <= 1.0          # entry by entry, 1e-9 of each entry's own magnitude
    if "xf" in d:
        assert rel_err(x_f, d["xf"]) < TOL
        assert worst_elementwise(x_f, d["xf"]) <= 1.0
    pairs, n3 = fz.neighbor_indices(atoms)
    for p, pair in enumerate(basis.interactions_map[2]):
        assert np.array_equal(pairs[pair], d[f"pair{p}_ij"])        # bit-exact neighbour indices
    if basis.degree > 2:
        assert np.array_equal(n3, d["n3_ij"])


def test_reference_api_slices_and_evaluate_configuration():
    d, meta, atoms = load_case("case_steel")
    basis = basis_from_meta(meta)
    fz = process.BasisFeaturizer(basis)
    ref = json.load(open(os.path.join(GOLDEN, "rattled_steel_features.json")))
    emap = fz.evaluate_configuration(atoms, energy=0, forces=np.zeros((3, len(atoms))))
    assert set(emap) == set(ref)
    for key in emap:                                   # tests/test_representation.py:605-648
        assert np.allclose(emap[key], np.array(ref[key]))
    from uf3_amd.data import geometry
    sup = geometry.get_supercell(atoms, r_cut=basis.r_cut)          # what the reference passes (process.py:332-335)
    lo2, hi2 = fz._block(2)
    lo3, hi3 = fz._block(3)
    assert np.allclose(fz.featurize_energy_2B(atoms, sup), d["xe"][lo2:hi2])
    assert np.allclose(fz.featurize_energy_3B(atoms, sup), d["xe"][lo3:hi3])
    assert np.allclose(fz.featurize_force_2B(atoms, sup), d["xf"][:, :, lo2:hi2])
    assert np.allclose(fz.featurize_force_3B(atoms, sup), d["xf"][:, :, lo3:hi3])
    assert fz.featurize_force_2B(atoms, sup).shape == (11, 3, hi2 - lo2)
    # no supercell (or the frame itself): an isolated cluster; anything else is refused, not reinterpreted
    cluster = fz.featurize_frames([atoms], periodic=False)[0][0, lo2:hi2]
    assert np.allclose(fz.featurize_energy_2B(atoms), cluster) and np.allclose(fz.featurize_energy_2B(atoms, atoms), cluster)
    # any tiling by whole lattice images that covers the cut-off is the reference's supercell as far as the features go
    # (get_supercell's default r_cut = 10, sort_indices=True); an arbitrary atom set is refused, not reinterpreted
    for sup2 in (geometry.get_supercell(atoms, r_cut=2 * basis.r_cut), geometry.get_supercell(atoms),
                 geometry.get_supercell(atoms, r_cut=basis.r_cut, sort_indices=True)):
        assert np.allclose(fz.featurize_energy_2B(atoms, sup2), d["xe"][lo2:hi2])
    assert np.allclose(fz.featurize_force_3B(atoms, geometry.get_supercell(atoms)), d["xf"][:, :, lo3:hi3])
    with pytest.raises(ValueError):                                  # (too few images for the cut-off)
        fz.featurize_energy_2B(atoms, Atoms(numbers=sup.get_atomic_numbers()[:3 * len(atoms)], positions=sup.get_positions()[:3 * len(atoms)]))
    with pytest.raises(ValueError):                                  # (one atom short of whole images)
        fz.featurize_energy_2B(atoms, Atoms(numbers=sup.get_atomic_numbers()[:-1], positions=sup.get_positions()[:-1]))


def test_2body_force_feature_invariants():
    """tests/test_bsplines.py:550-571 of the reference: rows sum to zero over atoms."""
    d, meta, atoms = load_case("case_w16")
    x_f = process.BasisFeaturizer(basis_from_meta(meta)).featurize_frames([atoms], energy=False)[1]
    assert np.abs(x_f.sum(axis=0)).max() < 1e-10 * np.abs(x_f).max()


@pytest.mark.parametrize("lead", [3, 0])
def test_w128_frames_batched(lead):
    d = np.load(os.path.join(GOLDEN, f"case_w128_energy_lead{lead}.npz"))
    basis = basis_from_meta(json.loads(str(d["meta"])))
    frames = read_extxyz(os.path.join(GOLDEN, "test.xyz"))
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, off = fz.featurize_frames(frames)                   # one batch of 5 frames
    assert rel_err(x_e, d["xe"]) < TOL and worst_elementwise(x_e, d["xe"]) <= 1.0
    pairs, n3 = fz.neighbor_indices(frames[0])
    assert np.array_equal(pairs[("W", "W")], d["pair0_ij_frame0"])
    assert np.array_equal(n3, d["n3_ij_frame0"])
    # force rows of frame 3 alone == its slice of the batch; oracle agrees
    alone = fz.featurize_frames([frames[3]], energy=False)[1]
    assert rel_err(alone, x_f[off[3]:off[4]]) < 1e-12   # LDS atomics: summation order varies
    ref = O.featurize(O.OracleBasis(basis), frames[3], energy=False)["xf"]
    assert rel_err(alone, ref) < TOL and worst_elementwise(alone, ref.reshape(alone.shape)) <= 1.0
    # all five frames' force rows, entry by entry, against the oracle (the reference itself needs hours for them without numba)
    for k, frame in enumerate(frames):
        ref_k = O.featurize(O.OracleBasis(basis), frame, energy=False)["xf"]
        assert worst_elementwise(x_f[off[k]:off[k + 1]], ref_k.reshape(-1, 3, x_f.shape[-1])) <= 1.0


CALC = json.load(open(os.path.join(GOLDEN, "calculator_cases.json")))


def _model_for(case):
    if case.get("model_file"):
        return ls.WeightedLinearModel.from_json(os.path.join(GOLDEN, case["model_file"]))
    basis = basis_from_meta(case["basis"])
    model = ls.WeightedLinearModel(basis)
    model.coefficients = np.array(case["coefficients"])
    return model


@pytest.mark.parametrize("name", ["unary_dimer_free", "unary_dimer_pbc", "unary_trimer", "unary_pbc",
                                  "binary_dimer", "w16_model23", "w54_model23"])
def test_calculator_energy_forces(name):
    case = CALC[name]
    calc = calculator.UFCalculator(_model_for(case))
    atoms = Atoms(numbers=case["numbers"], positions=case["positions"], cell=case["cell"], pbc=case["pbc"])
    e = calc.get_potential_energy(atoms)
    f = calc.get_forces(atoms)
    assert abs(e - case["energy"]) <= TOL * max(1, abs(case["energy"]))
    assert rel_err(f, case["forces"]) < TOL
    if case.get("literal"):                             # tests/test_calculator.py literals
        assert np.isclose(e, case["literal"]["energy"])
        assert np.allclose(f, case["literal"]["forces"])


def test_energy_and_forces_are_rows_dot_coefficients():
    model = ls.WeightedLinearModel.from_json(os.path.join(GOLDEN, "model_2and3.json"))
    frames = read_extxyz(os.path.join(GOLDEN, "test.xyz"))
    fz = process.BasisFeaturizer(model.bspline_config)
    calc = calculator.UFCalculator(model)
    x_e, x_f, off = fz.featurize_frames(frames[:2])
    e, f, _ = calc.evaluate_frames(frames[:2])
    assert np.allclose(x_e @ model.coefficients, e, rtol=1e-11)
    assert np.allclose(x_f @ model.coefficients, f, rtol=1e-9, atol=1e-10)
    assert abs(e[0] - CALC["w128_model23_energy"]["energy"]) < 1e-8
    stress = calc._get_stress(frames[0])                      # analytic virial / volume
    numeric = calc._get_stress(frames[0], numerical=True)     # the reference's finite-difference route
    assert stress.shape == (6,) and np.all(np.isfinite(stress))
    assert np.abs(stress - numeric).max() < 1e-6 * max(1e-3, np.abs(numeric).max())


def test_gram_and_fit_against_reference_capture():
    d = np.load(os.path.join(GOLDEN, "fit_case.npz"))
    basis = basis_from_meta(json.loads(str(d["meta"])))
    model = ls.WeightedLinearModel(basis, regularizer=d["regularizer"])
    model.fit(d["x_e"], d["y_e"], d["x_f"], d["y_f"], weight=float(d["kappa"][0]))
    assert np.allclose(model.coefficients, d["coefficients"], rtol=1e-6, atol=1e-8)
    assert np.array_equal(model.data_coverage, d["data_coverage"])
    pieces = model.gram_pieces(d["x_e"], d["y_e"], d["x_f"], d["y_f"])
    assert np.allclose(pieces["gram_e"], d["gram_e"], rtol=1e-11, atol=1e-11)
    assert np.allclose(pieces["gram_f"], d["gram_f"], rtol=1e-11, atol=1e-11)
    assert np.allclose(pieces["ord_f"], d["ord_f"], rtol=1e-11, atol=1e-11)
    m2 = ls.WeightedLinearModel(basis, regularizer=d["regularizer"])
    m2.fit(d["x_e"], d["y_e"])
    # 40 energy rows for 70 unknowns: the coefficients hang on the 1e-8 regulariser, so parity is
    # meaningful on the predictions, not on c
    assert np.allclose(m2.predict(d["x_e"]), d["x_e"] @ d["coefficients_energy_only"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("shape", [(1, 5), (37, 70), (1000, 434), (4099, 129)])
def test_gram_kernel_ragged_shapes(shape):
    rng = np.random.default_rng(shape[0])
    x, y = rng.normal(size=shape), rng.normal(size=shape[0])
    g, o = ls.gram_device(x, y)
    assert np.allclose(g, x.T @ x, rtol=1e-12, atol=1e-10)
    assert np.allclose(o, x.T @ y, rtol=1e-12, atol=1e-10)
    assert np.array_equal(g, g.T)


@pytest.mark.parametrize("shape", [(16384, 73), (20011, 80), (17003, 5), (40001, 33), (16385, 16)])
def test_slab_gram_kernel_for_narrow_matrices(shape):
    """k_gram_small (F <= 80, at least 16 k rows): every tile-pair count from 1 to 15, column counts inside a tile, row counts
    that are no multiple of the 32-row slab or of the rows per workgroup; X^T y from the staging threads; accumulate mode."""
    rng = np.random.default_rng(shape[1])
    x, y = rng.normal(size=shape), rng.normal(size=shape[0])
    g, o = ls.gram_device(x, y)
    ref = x.T @ x
    assert np.abs(g - ref).max() < 1e-12 * np.abs(ref).max()
    assert np.allclose(o, x.T @ y, rtol=1e-11, atol=1e-9)
    assert np.array_equal(g, g.T)


@pytest.mark.parametrize("shape", [(65536, 129), (70013, 257), (66001, 434), (65551, 333)])
def test_tiled_gram_kernel_ragged_shapes(shape):
    """The LDS-tiled kernel (more than 128 columns, at least 64 k rows): column counts that end inside a 64-column range /
    inside a 16-column tile, an odd number of ranges, row counts that are no multiple of the 16-row slab; X^T y is
    accumulated by the workgroups of the diagonal patches."""
    rng = np.random.default_rng(shape[1])
    x, y = rng.normal(size=shape), rng.normal(size=shape[0])
    g, o = ls.gram_device(x, y)
    ref = x.T @ x
    assert np.allclose(g, ref, rtol=1e-11, atol=1e-9 * np.abs(ref).max())
    assert np.allclose(o, x.T @ y, rtol=1e-11, atol=1e-9 * np.abs(ref).max())
    assert np.array_equal(g, g.T)


def test_oracle_parity_mid_size_configs():
    """configs[1] (1024-atom W) fully, configs[2] (4096-atom Ne-Xe) on the energy row + sampled atoms."""
    atoms, basis = synthetic.config_c2()
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, _ = fz.featurize_frames([atoms])
    ref = O.featurize(O.OracleBasis(basis), atoms)
    assert rel_err(x_e[0], ref["xe"]) < TOL and rel_err(x_f, ref["xf"]) < TOL
    assert worst_elementwise(x_e[0], ref["xe"]) <= 1.0 and worst_elementwise(x_f, ref["xf"].reshape(x_f.shape)) <= 1.0
    atoms, basis = synthetic.config_c3()
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, _ = fz.featurize_frames([atoms])
    ref = O.featurize(O.OracleBasis(basis), atoms)
    assert rel_err(x_e[0], ref["xe"]) < TOL and rel_err(x_f, ref["xf"]) < TOL
    assert worst_elementwise(x_e[0], ref["xe"]) <= 1.0 and worst_elementwise(x_f, ref["xf"].reshape(x_f.shape)) <= 1.0


def test_full_size_properties_10k_atoms():
    """configs[3]-sized frame (the bench workload): size-independent properties and the oracle on every row."""
    atoms, basis = synthetic.config_c4(binary=True)
    assert len(atoms) == 10000
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, _ = fz.featurize_frames([atoms])
    F = x_e.shape[1]
    assert F == 434
    scale = np.abs(x_f).max()
    assert np.abs(x_f.sum(axis=0)).max() < 1e-9 * scale               # translation invariance per column
    assert np.all(x_f[:, :, :2] == 0) and x_e[0, :2].sum() == 10000   # 1-body columns
    # a rigid translation (atoms leave the cell by a fraction of it; the reference's image range
    # still covers every neighbour) leaves every row unchanged.  NB: moving single atoms by whole
    # lattice vectors is NOT neutral in the reference (images are limited to +-ceil(r_cut/height),
    # geometry.py:131-138) and the kernels reproduce that.
    cell = atoms.get_cell()
    pos = atoms.get_positions() + np.array([0.37, -1.2, 2.9])
    moved = Atoms(numbers=atoms.get_atomic_numbers(), positions=pos, cell=cell, pbc=True)
    y_e, y_f, _ = fz.featurize_frames([moved])
    assert rel_err(y_e, x_e) < 1e-9 and rel_err(y_f, x_f) < 1e-8
    # finite difference of the energy row along one atom's x equals minus its force row
    h = 1e-5
    rows = []
    for sgn in (+1, -1):
        p = atoms.get_positions()
        p[1234, 0] += sgn * h
        rows.append(fz.featurize_frames([Atoms(numbers=atoms.get_atomic_numbers(), positions=p, cell=cell, pbc=True)],
                                        forces=False)[0][0])
    fd = -(rows[0] - rows[1]) / (2 * h)
    assert np.abs(fd - x_f[1234, 0]).max() < 1e-5 * max(1.0, np.abs(x_f[1234, 0]).max())
    # the whole frame against the oracle: energy row and all 30 000 force rows (2.4 s of CPU)
    ref = O.featurize(O.OracleBasis(basis), atoms)
    assert rel_err(x_e[0], ref["xe"]) < TOL
    assert rel_err(x_f, ref["xf"].reshape(x_f.shape)) < TOL
    # ... and entry by entry: each of the 13 M entries within 1e-9 of ITS OWN magnitude (small outer-shell columns included)
    assert worst_elementwise(x_e[0], ref["xe"]) <= 1.0
    assert worst_elementwise(x_f, ref["xf"].reshape(x_f.shape)) <= 1.0


def test_unknown_species_raises():
    d, meta, atoms = load_case("case_h2o")
    fz = process.BasisFeaturizer(basis_from_meta(meta))
    with pytest.raises(_lib.SpeciesError):
        fz.featurize_frames([Atoms("Ar2", positions=[[0, 0, 0], [3, 0, 0]])])


def test_device_resident_fit_pipeline_matches_host_rows():
    """frames -> rows -> Gram pieces entirely in HBM == oracle fit on the downloaded rows."""
    import torch
    from uf3_amd import pipeline
    basis = synthetic.notebook_basis(['W'])
    frames = [synthetic.lattice_frame("bcc", (3, 3, 3), 3.165, [74], seed=50 + k) for k in range(6)]
    fz = process.BasisFeaturizer(basis)
    reg = basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0)
    rng = np.random.default_rng(7)
    c_true = rng.normal(0, 1, basis.n_feats)
    c_true[basis.col_idx] = 0
    x_e, x_f, off = fz.featurize_frames(frames)
    energies = x_e @ c_true + rng.normal(0, 1e-3, len(frames))
    forces_flat = x_f.reshape(-1, basis.n_feats) @ c_true + rng.normal(0, 1e-3, 3 * off[-1])
    forces = [forces_flat[3 * off[k]:3 * off[k + 1]].reshape(-1, 3) for k in range(len(frames))]
    model = ls.WeightedLinearModel(basis, regularizer=reg)
    pieces = pipeline.fit_frames(model, fz, frames, energies, forces, weight=0.4, reduce=False)
    n = x_e[:, :1].sum(axis=1)
    ref = O.fit(basis, reg, x_e / n[:, None], energies / n, x_f.reshape(-1, basis.n_feats), forces_flat, weight=0.4)
    assert np.allclose(pieces["gram_f"], ref["gram_f"], rtol=1e-10, atol=1e-9)
    assert np.allclose(pieces["gram_e"], ref["gram_e"], rtol=1e-10, atol=1e-12)
    assert np.allclose(model.predict(x_f.reshape(-1, basis.n_feats)), x_f.reshape(-1, basis.n_feats) @ ref["coefficients"],
                       rtol=1e-6, atol=1e-7)
    # the planted model is recovered wherever the data reach (short-range pair columns see no data)
    pred, true = x_f.reshape(-1, basis.n_feats) @ model.coefficients, x_f.reshape(-1, basis.n_feats) @ c_true
    assert np.abs(pred - true).max() < 2e-2 * np.abs(true).max()


def test_config4_fit_in_chunks_10k_atom_tungsten_frames():
    """BASELINE config 4 on one GPU's share: 10 000-atom W frames (F = 73) x 66 through the device-resident
    accumulator in five chunks == the oracle's fit on the downloaded rows (least_squares.py:274-321)."""
    from uf3_amd import pipeline
    basis = synthetic.notebook_basis(['W'])
    n_frames = 66
    frames = [synthetic.config_c4(frame=k, binary=False)[0] for k in range(n_frames)]
    assert len(frames[0]) == 10000 and basis.n_feats == 73
    fz = process.BasisFeaturizer(basis)
    reg = basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0)
    x_e, x_f, off = fz.featurize_frames(frames)
    x_f = x_f.reshape(-1, basis.n_feats)
    rng = np.random.default_rng(7)
    c_true = rng.normal(0, 1, basis.n_feats)
    c_true[basis.col_idx] = 0
    energies = x_e @ c_true + rng.normal(0, 1e-3, n_frames)
    forces_flat = x_f @ c_true + rng.normal(0, 1e-3, len(x_f))
    forces = [forces_flat[3 * off[k]:3 * off[k + 1]].reshape(-1, 3) for k in range(n_frames)]
    model = ls.WeightedLinearModel(basis, regularizer=reg)
    acc = pipeline.DeviceFitAccumulator(model, fz, max_atoms_per_chunk=250000)
    acc.add_frames(frames, energies, forces)
    assert acc.n_chunks == 5                 # (chunks grow from an eighth of the limit, the rest is spread evenly: 3 + 6 + 12 + 23 + 22 frames)
    pieces = acc.pieces()
    model.fit_from_pieces(pieces, weight=0.3)
    n = x_e[:, :1].sum(axis=1)
    ref = O.fit(basis, reg, x_e / n[:, None], energies / n, x_f, forces_flat, weight=0.3)
    for key in ("gram_e", "gram_f", "ord_e", "ord_f"):
        assert rel_err(pieces[key], ref[key]) < 1e-9, key
    pred = model.predict(x_f)
    assert rel_err(pred, x_f @ ref["coefficients"]) < 1e-6
    assert np.abs(pred - x_f @ c_true).max() < 2e-2 * np.abs(x_f @ c_true).max()


def test_native_fit_accumulator_matches_the_torch_backed_one_and_the_oracle():
    """uf3_fit_create / uf3_fit_add / uf3_fit_pack (the fit's accumulation inside the library: one pointer per frame in, pinned
    staging and copy stream of its own, no PyTorch) == pipeline.DeviceFitAccumulator == the oracle's fit on the downloaded rows;
    several chunks, frames of different sizes, with and without forces, frozen columns, a second call after reset."""
    from uf3_amd import pipeline
    basis = synthetic.notebook_basis(['Mo', 'W'])
    frames = [synthetic.lattice_frame("bcc", (4 + k % 3, 4, 3 + k % 2), 3.165, [42, 74], seed=300 + k) for k in range(11)]
    fz = process.BasisFeaturizer(basis)
    reg = basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0)
    x_e, x_f, off = fz.featurize_frames(frames)
    x_f = x_f.reshape(-1, basis.n_feats)
    rng = np.random.default_rng(27)
    c_true = rng.normal(0, 1, basis.n_feats)
    c_true[basis.col_idx] = 0
    energies = x_e @ c_true + rng.normal(0, 1e-3, len(frames))
    forces_flat = x_f @ c_true + rng.normal(0, 1e-3, len(x_f))
    forces = [forces_flat[3 * off[k]:3 * off[k + 1]].reshape(-1, 3) for k in range(len(frames))]
    model = ls.WeightedLinearModel(basis, regularizer=reg)
    native = pipeline.NativeFitAccumulator(model, fz, max_atoms_per_chunk=400)
    native.add_frames(frames, energies, forces)
    assert native.n_chunks >= 4
    pieces = native.pieces()
    n = x_e[:, :2].sum(axis=1)
    ref = O.fit(basis, reg, x_e / n[:, None], energies / n, x_f, forces_flat, weight=0.3)
    for key in ("gram_e", "gram_f", "ord_e", "ord_f"):
        assert rel_err(pieces[key], ref[key]) < 1e-9, key
    torch_backed = pipeline.DeviceFitAccumulator(model, fz, max_atoms_per_chunk=400)
    torch_backed.add_frames(frames, energies, forces)
    other = torch_backed.pieces()
    for key in pieces:
        assert rel_err(pieces[key], other[key]) < 1e-11, key
    model.fit_from_pieces(pieces, weight=0.3)
    assert rel_err(model.predict(x_f), x_f @ ref["coefficients"]) < 1e-6
    # again after a reset, through the convenience entry, in one chunk; then without forces
    native.reset()
    native.add_frames(frames[:3], energies[:3], forces[:3])
    part = native.pieces()
    native.reset()
    native.add_frames(frames, energies, forces)
    again = native.pieces()
    for key in pieces:
        assert rel_err(again[key], pieces[key]) < 1e-11 and (key.startswith("m_") or rel_err(part[key], pieces[key]) > 1e-3), key
    m2 = ls.WeightedLinearModel(basis, regularizer=reg)
    pipeline.fit_frames_native(m2, fz, frames, energies, forces, weight=0.3)
    assert np.allclose(m2.coefficients, model.coefficients, rtol=1e-5, atol=1e-6)
    m3, m4 = ls.WeightedLinearModel(basis, regularizer=reg), ls.WeightedLinearModel(basis, regularizer=reg)
    p3 = pipeline.fit_frames_native(m3, fz, frames, energies, None)          # (energies only: 11 rows for 425 unknowns -- compare the
    p4 = pipeline.fit_frames(m4, fz, frames, energies, None, reduce=False)   #  pieces, the solution is the regulariser's)
    assert set(p3) == set(p4) == {"gram_e", "ord_e", "m_e"}
    for key in p3:
        assert rel_err(p3[key], p4[key]) < 1e-11, key
    with pytest.raises(ValueError):
        native.add_frames(frames, energies, None)


def test_two_element_fit_in_chunks_through_the_tiled_gram_kernel():
    """The metric variant of config 4 (W/Mo, F = 434) through the device-resident accumulator: a chunk of five 10 000-atom
    frames (150 000 force rows: listed by species, each list through the LDS-tiled X^T X kernel on its species' columns,
    X^T y riding along) and one of three (90 000 rows: the plain tiled product, accumulating on top); pieces == the
    oracle's on the downloaded rows."""
    from uf3_amd import pipeline
    basis = synthetic.notebook_basis(['Mo', 'W'])
    n_frames = 8
    frames = [synthetic.config_c4(frame=k)[0] for k in range(n_frames)]
    assert len(frames[0]) == 10000 and basis.n_feats == 434
    fz = process.BasisFeaturizer(basis)
    reg = basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0)
    x_e, x_f, off = fz.featurize_frames(frames)
    x_f = x_f.reshape(-1, basis.n_feats)
    rng = np.random.default_rng(17)
    c_true = rng.normal(0, 1, basis.n_feats)
    c_true[basis.col_idx] = 0
    energies = x_e @ c_true + rng.normal(0, 1e-3, n_frames)
    forces_flat = x_f @ c_true + rng.normal(0, 1e-3, len(x_f))
    forces = [forces_flat[3 * off[k]:3 * off[k + 1]].reshape(-1, 3) for k in range(n_frames)]
    model = ls.WeightedLinearModel(basis, regularizer=reg)
    acc = pipeline.DeviceFitAccumulator(model, fz, max_atoms_per_chunk=50000, first_chunk_fraction=1.0)
    acc.add_frames(frames, energies, forces)
    assert acc.n_chunks == 2
    pieces = acc.pieces()
    n = x_e[:, :2].sum(axis=1)
    ref = O.fit(basis, reg, x_e / n[:, None], energies / n, x_f, forces_flat, weight=0.3)
    for key in ("gram_e", "gram_f", "ord_e", "ord_f"):
        assert rel_err(pieces[key], ref[key]) < 1e-9, key
    model.fit_from_pieces(pieces, weight=0.3)
    pred = model.predict(x_f)
    assert rel_err(pred, x_f @ ref["coefficients"]) < 1e-6


@pytest.mark.parametrize("elements,numbers,reps", [(['Mo', 'W'], [42, 74], (10, 20, 25)),
                                                    (['Mo', 'W'], [42, 74, 74, 74, 74], (10, 20, 25)),
                                                    (['Mo', 'Nb', 'W'], [41, 42, 74], (12, 22, 25))])
def test_force_row_gram_by_species_equals_the_dense_product(elements, numbers, reps):
    """uf3_gram_force_rows_dev (rows listed by species on the device, each list multiplied on the columns of its species'
    blocks) == uf3_gram_dev on the same rows in HBM, X^T y included; overwrite, then accumulate.  Even and 20 / 80
    compositions, three species; and the rows really are zero outside the columns the entry multiplies."""
    import torch
    from uf3_amd.data import composition
    dev = torch.device("cuda", 0)
    basis = synthetic.notebook_basis(elements)
    fz = process.BasisFeaturizer(basis)
    ctx, db = fz._dev()
    F = basis.n_feats
    frames = [synthetic.lattice_frame("bcc", reps, 3.165, numbers, seed=900 + k) for k in range(5)]
    batch = _lib.FrameBatch(frames)
    n_atoms = batch.n_atoms
    assert 3 * n_atoms // len(elements) >= 65536
    d_pos, d_z = torch.from_numpy(batch.pos).to(dev), torch.from_numpy(batch.z).to(dev)
    x_e = torch.empty((len(frames), F), dtype=torch.float64, device=dev)
    x_f = torch.empty((3 * n_atoms, F), dtype=torch.float64, device=dev)
    y_f = torch.from_numpy(np.random.default_rng(3).normal(size=3 * n_atoms)).to(dev)
    prev = ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    try:
        fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), x_e.data_ptr(), x_f.data_ptr())
        g_ref, o_ref = torch.empty((F, F), dtype=torch.float64, device=dev), torch.empty(F, dtype=torch.float64, device=dev)
        g, o = torch.full((F, F), 7.0, dtype=torch.float64, device=dev), torch.full((F,), 7.0, dtype=torch.float64, device=dev)
        ctx.check(ctx.lib.uf3_gram_dev(ctx.handle, x_f.data_ptr(), y_f.data_ptr(), 3 * n_atoms, F, F, 0, g_ref.data_ptr(),
                                       o_ref.data_ptr()))
        ctx.check(ctx.lib.uf3_gram_force_rows_dev(db.handle, x_f.data_ptr(), y_f.data_ptr(), d_z.data_ptr(), n_atoms, F, 0,
                                                  g.data_ptr(), o.data_ptr()))
        ctx.synchronize()
        g1, o1 = g.cpu().numpy(), o.cpu().numpy()
        ctx.check(ctx.lib.uf3_gram_force_rows_dev(db.handle, x_f.data_ptr(), y_f.data_ptr(), d_z.data_ptr(), n_atoms, F, 1,
                                                  g.data_ptr(), o.data_ptr()))
        ctx.synchronize()
    finally:
        ctx.restore_stream(prev)
    g_ref, o_ref = g_ref.cpu().numpy(), o_ref.cpu().numpy()
    scale = np.abs(g_ref).max()
    assert np.abs(g1 - g_ref).max() < 1e-12 * scale and np.array_equal(g1, g1.T)
    assert np.abs(o1 - o_ref).max() < 1e-11 * np.abs(o_ref).max()
    assert np.abs(g.cpu().numpy() - 2 * g_ref).max() < 2e-12 * scale
    assert np.abs(o.cpu().numpy() - 2 * o_ref).max() < 2e-11 * np.abs(o_ref).max()
    # the premise, on the rows themselves: an atom's rows vanish in every block its species takes no part in
    sizes, offsets = basis.get_interaction_partitions()
    rows = x_f.view(n_atoms, 3, F)
    for el in basis.element_list:
        inside = np.zeros(F, dtype=bool)
        for inter in sizes:
            if not isinstance(inter, str) and el in inter:
                inside[offsets[inter]:offsets[inter] + sizes[inter]] = True
        assert 0 < inside.sum() < F
        idx = torch.from_numpy(np.flatnonzero(batch.z == composition.atomic_numbers[el])[:4000]).to(dev)
        assert len(idx) > 100
        outside = torch.from_numpy(np.flatnonzero(~inside)).to(dev)
        assert float(rows[idx][:, :, outside].abs().max()) == 0.0


def test_fit_bookkeeping_entries_against_numpy():
    """uf3_fit_rows_dev / uf3_fit_pack_dev (per-atom normalisation, moments of the frozen energies and of the force targets,
    frozen columns folded out of the packed pieces) against the reference's host arithmetic restated in NumPy
    (least_squares.py:697-700, :296-304, :817-890)."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    ctx = _lib.get_context(0)
    prev = ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    try:
        rng = np.random.default_rng(5)
        nf, F, n_yf = 37, 61, 10007
        x_e = rng.normal(size=(nf, F)); counts = rng.integers(3, 90, nf).astype(float)
        y_e = rng.normal(size=nf); y_f = rng.normal(size=n_yf)
        frozen = np.array([0, 7, 8, 40], dtype=np.int64); c_fro = rng.normal(size=4)
        keep = np.array([q for q in range(F) if q not in set(frozen.tolist())], dtype=np.int64)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
        d_x, d_c, d_ye, d_yf, d_fro, d_cf, d_keep = t(x_e), t(counts), t(y_e), t(y_f), t(frozen), t(c_fro), t(keep)
        flat = rng.normal(size=2 * F * F + 2 * F + 6)
        flat[-6:] = 0.0
        d_flat = t(flat)
        ctx.check(ctx.lib.uf3_fit_rows_dev(ctx.handle, nf, F, d_x.data_ptr(), d_c.data_ptr(), d_ye.data_ptr(), d_yf.data_ptr(), n_yf,
                                           d_fro.data_ptr(), d_cf.data_ptr(), 4, d_flat[-6:].data_ptr()))
        x_n = x_e / counts[:, None]
        assert np.array_equal(d_x.cpu().numpy(), x_n)                           # one division per entry: the same bits
        y_fro = y_e - x_n[:, frozen] @ c_fro
        m = d_flat[-6:].cpu().numpy()
        assert np.allclose(m[[1, 2, 4, 5]], [y_fro.sum(), (y_fro ** 2).sum(), y_f.sum(), (y_f ** 2).sum()], rtol=1e-12, atol=1e-12)
        K = len(keep)
        out = torch.empty(2 * K * K + 2 * K + 6, dtype=torch.float64, device=dev)
        ctx.check(ctx.lib.uf3_fit_pack_dev(ctx.handle, F, d_flat.data_ptr(), d_keep.data_ptr(), K, d_fro.data_ptr(), d_cf.data_ptr(), 4,
                                           float(nf), float(n_yf), out.data_ptr()))
        got = out.cpu().numpy()
        full = d_flat.cpu().numpy()
        for which in range(2):
            G = full[which * F * F:(which + 1) * F * F].reshape(F, F)
            o = full[2 * F * F + which * F:2 * F * F + (which + 1) * F]
            assert np.array_equal(got[which * K * K:(which + 1) * K * K].reshape(K, K), G[np.ix_(keep, keep)])
            assert np.allclose(got[2 * K * K + which * K:2 * K * K + (which + 1) * K], o[keep] - G[np.ix_(keep, frozen)] @ c_fro, rtol=1e-13, atol=1e-13)
        assert np.array_equal(got[-6:], [nf, m[1], m[2], n_yf, m[4], m[5]])
        # no frozen columns, no force targets
        d_flat2 = t(np.zeros(2 * F * F + 2 * F + 6))
        ctx.check(ctx.lib.uf3_fit_rows_dev(ctx.handle, nf, F, t(x_e).data_ptr(), d_c.data_ptr(), d_ye.data_ptr(), None, 0, None, None, 0,
                                           d_flat2[-6:].data_ptr()))
        assert np.allclose(d_flat2[-6:].cpu().numpy(), [0, y_e.sum(), (y_e ** 2).sum(), 0, 0, 0], rtol=1e-12)
    finally:
        ctx.restore_stream(prev)


def test_ragged_batch_of_large_frames_equals_per_frame_calls():
    """32 frames of 1 000 - 10 000 atoms (mixed sizes, W/Mo) in one batch == one call per frame."""
    basis = synthetic.notebook_basis(['Mo', 'W'])
    shapes = [(5, 10, 10), (8, 8, 8), (10, 20, 25), (6, 9, 14), (10, 10, 13), (7, 7, 11), (9, 12, 17), (10, 15, 20)]
    frames = [synthetic.lattice_frame("bcc", shapes[k % len(shapes)], 3.165, [42, 74], 800 + k) for k in range(32)]
    assert min(len(f) for f in frames) >= 1000 and max(len(f) for f in frames) == 10000
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, off = fz.featurize_frames(frames, max_bytes=12 << 30)
    for k in (0, 2, 5, 13, 31):
        y_e, y_f, _ = fz.featurize_frames([frames[k]])
        assert rel_err(x_e[k], y_e[0]) < 1e-12 and rel_err(x_f[off[k]:off[k + 1]], y_f) < 1e-12
    ref = O.featurize(O.OracleBasis(basis), frames[3])
    assert rel_err(x_e[3], ref["xe"]) < TOL and rel_err(x_f[off[3]:off[4]], ref["xf"].reshape(-1, 3, basis.n_feats)) < TOL


def test_bases_and_featurizers_stay_picklable_and_follow_basis_updates():
    """Device tables live outside the objects: pickling / deep copies work after GPU use, and a change of
    the knots retires the tables built for the old ones."""
    import copy
    import pickle
    atoms, basis = synthetic.config_c2()
    fz = process.BasisFeaturizer(basis)
    x_e, _, _ = fz.featurize_frames([atoms], forces=False)
    fz2 = pickle.loads(pickle.dumps(fz))
    fz3 = copy.deepcopy(fz)
    assert rel_err(fz2.featurize_frames([atoms], forces=False)[0], x_e) < 1e-13
    assert rel_err(fz3.featurize_frames([atoms], forces=False)[0], x_e) < 1e-13
    # same object, new knots: the rows must be those of a basis built with the new knots from scratch
    pair = basis.interactions_map[2][0]
    basis.update_knots(r_max_map={pair: 4.8})
    basis.update_basis_functions()
    fresh = synthetic.notebook_basis(['W'])
    fresh.update_knots(r_max_map={pair: 4.8})
    fresh.update_basis_functions()
    y_e = fz.featurize_frames([atoms], forces=False)[0]
    z_e = process.BasisFeaturizer(fresh).featurize_frames([atoms], forces=False)[0]
    assert rel_err(y_e, z_e) < 1e-13 and rel_err(y_e, x_e) > 1e-3


def test_asynchronous_featurize_reports_a_capacity_overflow_afterwards():
    """``uf3_featurize_dev`` does not wait once a context knows its capacities; a batch that needs longer lists than
    any before is flagged by the next synchronisation (UF3_ERETRY) and succeeds when repeated."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)                                   # a context of its own: capacities start from scratch
    basis = synthetic.notebook_basis(['W'])
    db = _lib.DeviceBasis(basis, ctx)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def run(frames):
        batch = _lib.FrameBatch(frames)
        d_pos, d_z = torch.from_numpy(batch.pos).to(dev), torch.from_numpy(batch.z).to(dev)
        x_e = torch.empty((batch.n_frames, db.n_feat), dtype=torch.float64, device=dev)
        x_f = torch.empty((batch.n_atoms, 3, db.n_feat), dtype=torch.float64, device=dev)
        ctx.check(ctx.lib.uf3_featurize_dev(db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()),
                                            C.c_void_p(d_z.data_ptr()), C.c_void_p(x_e.data_ptr()), C.c_void_p(x_f.data_ptr())))
        return x_e, x_f

    sparse = [synthetic.lattice_frame("bcc", (4, 4, 4), 3.165, [74], 90 + k) for k in range(2)]
    dense = [synthetic.lattice_frame("bcc", (5, 5, 5), 2.4, [74], 95, rattle=0.05)]         # third shell inside the 3-body range
    run(sparse); ctx.synchronize()                          # learns the capacities (synchronous calls)
    run(sparse); ctx.synchronize()
    run(dense)                                              # asynchronous, lists too short
    with pytest.raises(_lib.RetryError):
        ctx.synchronize()
    x_e, x_f = run(dense)                                   # capacities were raised: now it fits
    ctx.synchronize()
    ref = O.featurize(O.OracleBasis(basis), dense[0])
    assert rel_err(x_e[0].cpu().numpy(), ref["xe"]) < TOL
    assert rel_err(x_f.cpu().numpy().reshape(ref["xf"].shape), ref["xf"]) < TOL


def test_asynchronous_verdict_waits_for_the_synchronize_of_its_owner():
    """A capacity overflow of an asynchronous call is reported by the next ``uf3_ctx_synchronize`` even when other entries
    of the same context (a host-buffer featurize, an evaluator call) ran in between and met the verdict first."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    basis = synthetic.notebook_basis(['W'])
    db = _lib.DeviceBasis(basis, ctx)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    keep = []

    def run_async(frames):
        batch = _lib.FrameBatch(frames)
        d_pos, d_z = torch.from_numpy(batch.pos).to(dev), torch.from_numpy(batch.z).to(dev)
        x_f = torch.empty((batch.n_atoms, 3, db.n_feat), dtype=torch.float64, device=dev)
        keep.append((batch, d_pos, d_z, x_f))
        ctx.check(ctx.lib.uf3_featurize_dev(db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()),
                                            C.c_void_p(d_z.data_ptr()), None, C.c_void_p(x_f.data_ptr())))

    def run_host(frames):
        batch = _lib.FrameBatch(frames)
        x_e = np.empty((batch.n_frames, db.n_feat))
        ctx.check(ctx.lib.uf3_featurize(db.handle, C.byref(batch.struct), _lib._p(batch.pos), _lib._p(batch.z), _lib._p(x_e), None))
        return x_e

    sparse = [synthetic.lattice_frame("bcc", (4, 4, 4), 3.165, [74], 90 + k) for k in range(2)]
    dense = [synthetic.lattice_frame("bcc", (5, 5, 5), 2.4, [74], 95, rattle=0.05)]
    run_async(sparse); ctx.synchronize()
    run_async(sparse); ctx.synchronize()
    run_async(dense)                                        # asynchronous, lists too short: its rows are invalid
    torch.cuda.synchronize(dev)
    x_e = run_host(sparse)                                  # an unrelated synchronous call meets the verdict first ...
    ref = O.featurize(O.OracleBasis(basis), sparse[0], forces=False)
    assert rel_err(x_e[0], ref["xe"]) < TOL                 # ... and is not disturbed by it
    with pytest.raises(_lib.RetryError):                    # the owner still learns about it
        ctx.synchronize()
    ctx.synchronize()                                       # reported once
    run_async(dense); ctx.synchronize()                     # capacities were raised: now it fits
    ref = O.featurize(O.OracleBasis(basis), dense[0])
    assert rel_err(keep[-1][3].cpu().numpy().reshape(ref["xf"].shape), ref["xf"]) < TOL


def test_device_entries_follow_the_callers_stream():
    """_dev entries on torch's (null) stream see data produced just before on that stream."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    ctx = _lib.get_context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    for rows in (50000, 200000):
        x = torch.randn((rows, 160), dtype=torch.float64, device=dev)      # produced on the same stream, no sync
        y = torch.randn((rows,), dtype=torch.float64, device=dev)
        g = torch.empty((160, 160), dtype=torch.float64, device=dev)
        o = torch.empty((160,), dtype=torch.float64, device=dev)
        ctx.check(ctx.lib.uf3_gram_dev(ctx.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), rows, 160, 160, 0,
                                       C.c_void_p(g.data_ptr()), C.c_void_p(o.data_ptr())))
        ref = x.T @ x
        assert ((g - ref).abs().max() / ref.abs().max()).item() < 1e-12


# ------------------------------------------------------------------------------------------------
# edge cases (the oracle is the checker)
# ------------------------------------------------------------------------------------------------
def _check_against_oracle(basis, frames, tol=TOL):
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, off = fz.featurize_frames(frames)
    ob = O.OracleBasis(basis)
    for k, atoms in enumerate(frames):
        ref = O.featurize(ob, atoms)
        assert rel_err(x_e[k], ref["xe"]) < tol, k
        assert rel_err(x_f[off[k]:off[k + 1]], ref["xf"]) < tol, k
    return x_e, x_f


def test_ragged_batch_mixed_boundary_conditions():
    """one call: isolated atom, dimer out of range, cluster, slab, tiny triclinic cell, 300-atom bulk."""
    rng = np.random.default_rng(3)
    basis = synthetic.notebook_basis(['Mo', 'W'], lead3=0)
    cell = np.array([[4.1, 0.3, 0.0], [-0.5, 3.9, 0.2], [0.4, -0.3, 4.4]])
    frames = [
        Atoms(numbers=[74], positions=[[0.0, 0.0, 0.0]]),                                    # no neighbours at all
        Atoms(numbers=[74, 42], positions=[[0, 0, 0], [9.0, 0, 0]]),                        # out of every range
        Atoms(numbers=rng.choice([42, 74], 9), positions=rng.uniform(0, 5.5, (9, 3))),      # cluster
        Atoms(numbers=rng.choice([42, 74], 12), positions=rng.uniform(-1, 7, (12, 3)),
              cell=np.diag([6.3, 6.9, 20.0]), pbc=[True, True, False]),                     # slab, atoms outside cell
        Atoms(numbers=rng.choice([42, 74], 3), positions=rng.uniform(0, 4, (3, 3)), cell=cell, pbc=True),   # cell < r_cut
        synthetic.lattice_frame("bcc", (5, 5, 6), 3.165, [42, 74], seed=9),
    ]
    x_e, x_f = _check_against_oracle(basis, frames)
    assert np.count_nonzero(x_f[0]) == 0 and x_e[0, 1] == 1.0 and np.count_nonzero(x_e[0, 2:]) == 0


def test_dense_neighbourhoods_and_capacity_regrowth():
    """> 64 three-body neighbours per atom and > 128 pair candidates: list loops and capacity retries."""
    rng = np.random.default_rng(5)
    cs = synthetic.composition.ChemicalSystem(['W'], 3)
    basis = synthetic.bspline.BSplineBasis(
        cs, r_min_map={('W', 'W'): 0.3, ('W', 'W', 'W'): [0.5, 0.5, 0.5]},
        r_max_map={('W', 'W'): 6.5, ('W', 'W', 'W'): [5.0, 5.0, 10.0]},
        resolution_map={('W', 'W'): 12, ('W', 'W', 'W'): [5, 5, 10]}, leading_trim=0, trailing_trim=3)
    # 4 x 2.6 A = 10.4 A >= 2 r_max3: the reference's supercell holds every third atom of a ghost-centred
    # triplet (below that it silently drops force terms -- DESIGN.md section 7 -- and parity is undefined)
    atoms = synthetic.lattice_frame("bcc", (4, 4, 4), 2.6, [74], seed=2, rattle=0.1)      # rho = 0.114 / A^3
    _check_against_oracle(basis, [atoms], tol=1e-8)
    _, n3 = process.BasisFeaturizer(basis).neighbor_indices(atoms)
    assert len(n3) / len(atoms) > 50


def test_lists_too_long_for_the_bond_factorised_launch_fall_back():
    """k_featurize3's LDS layout grows ~170-260 B per list entry and wave; lists of > ~100 (6 x 12 windows)
    or > ~200 entries (default trims) do not fit 160 KB.  Such calls must run on the matrix-core / generic launches that
    served them before, not fail with UF3_EOVERFLOW."""
    # 6 x 12 windows, ~115 three-body neighbours per atom (cell edge 7.1 A >= 2 r_max3): against the oracle
    basis = synthetic.notebook_basis(['Mo', 'W'], lead3=0)
    atoms = synthetic.lattice_frame("bcc", (5, 5, 5), 1.42, [42, 74], 5, rattle=0.05, strain=0.0)
    fz = process.BasisFeaturizer(basis)
    assert fz._dev()[1].featurizer_modes & 0x1000                      # the basis qualifies; the lists do not
    _, n3 = fz.neighbor_indices(atoms)
    assert np.bincount(n3[:, 0]).max() > 100
    _check_against_oracle(basis, [atoms])
    # default trims, > 200 three-body neighbours per atom (a short pair range keeps the pair launch's candidate stage small):
    # the default route against the generic kernels
    cs = synthetic.composition.ChemicalSystem(['W'], 3)
    basis3 = synthetic.bspline.BSplineBasis(
        cs, r_min_map={('W', 'W'): 0.3, ('W', 'W', 'W'): [1.0, 1.0, 1.0]},
        r_max_map={('W', 'W'): 2.5, ('W', 'W', 'W'): [4.0, 4.0, 8.0]},
        resolution_map={('W', 'W'): 10, ('W', 'W', 'W'): [6, 6, 12]}, leading_trim={2: 0, 3: 3}, trailing_trim={2: 3, 3: 3})
    dense = synthetic.lattice_frame("bcc", (6, 6, 6), 1.357, [74], 6, rattle=0.03, strain=0.0)
    _, n3 = process.BasisFeaturizer(basis3).neighbor_indices(dense)
    assert np.bincount(n3[:, 0]).max() > 200
    xe_d, xf_d, modes_d = _fresh_rows(basis3, [dense])
    xe_g, xf_g, modes_g = _fresh_rows(basis3, [dense], UF3_NO_FEAT3="1", UF3_NO_MFMA_FEAT="1")
    assert (modes_d & 0x1000) and not (modes_g & 0x13c0)
    assert rel_err(xe_d, xe_g) < 1e-11 and rel_err(xf_d, xf_g) < 1e-11


@pytest.mark.parametrize("lattice,a,r3,want", [("fcc", 3.9, 4.0, 18), ("bcc", 3.165, 4.6, 26)])
def test_bond_factorised_launch_at_list_capacities_24_and_32(lattice, a, r3, want, monkeypatch):
    """k_featurize3 has the list capacity as a compile-time constant at 16, 24 and 32 (16 alone was tuned to the
    bench cells): an fcc cell with 18 and a bcc cell with 26 three-body neighbours per atom settle on 24 and 32 -- the tuned
    calls against the oracle and against the first call, which ran the generic instance at the estimated capacity."""
    cs = synthetic.composition.ChemicalSystem(['Mo', 'W'], 3)
    pairs, trios = cs.interactions_map[2], cs.interactions_map[3]
    basis = synthetic.bspline.BSplineBasis(
        cs, r_min_map={**{p: 0.001 for p in pairs}, **{t: [1.5, 1.5, 1.5] for t in trios}},
        r_max_map={**{p: 5.5 for p in pairs}, **{t: [r3, r3, 2 * r3] for t in trios}},
        resolution_map={**{p: 15 for p in pairs}, **{t: [6, 6, 12] for t in trios}}, leading_trim={2: 0, 3: 3}, trailing_trim={2: 3, 3: 3})
    frames = [synthetic.lattice_frame(lattice, (4, 4, 4), a, [42, 74], seed=70 + k, rattle=0.03, strain=0.0) for k in range(2)]
    monkeypatch.setattr(_lib, "_contexts", {})                    # (a context of its own: capacities are a context's grow-only memory)
    fz = process.BasisFeaturizer(basis)
    assert fz._dev()[1].featurizer_modes & 0x1000
    _, n3 = fz.neighbor_indices(frames[0])
    assert np.bincount(n3[:, 0]).max() == want
    ob = O.OracleBasis(basis)
    first = fz.featurize_frames(frames)                           # (at the estimated capacity: the generic instance)
    x_e, x_f, off = fz.featurize_frames(frames)                   # (at the tuned capacity: the instance with the constant)
    for k, atoms in enumerate(frames):
        ref = O.featurize(ob, atoms)
        assert rel_err(x_e[k], ref["xe"]) < TOL and rel_err(x_f[off[k]:off[k + 1]], ref["xf"]) < TOL
    assert rel_err(first[1], x_f) < 1e-13 and rel_err(first[0], x_e) < 1e-13
    _lib.drop_device_basis(basis)


def test_instance_for_short_lists_is_picked_on_the_device_when_the_context_remembers_a_dense_batch(monkeypatch, capfd):
    """A context's list capacity only grows (a dense batch leaves it at 32 or more), and a context's first call runs at an
    estimate: k_featurize3's instance laid out for 16 entries is then chosen on the device from the batch's longest list
    (two launches, one leaves at once) -- rows against the oracle, and the same rows as a context that never saw the dense
    batch (angles.py:142-286)."""
    basis = synthetic.notebook_basis(['Mo', 'W'])
    dense = synthetic.lattice_frame("bcc", (5, 5, 5), 2.4, [42, 74], seed=5, rattle=0.03, strain=0.0)
    frames = [synthetic.lattice_frame("bcc", (4, 4, 5), 3.165, [42, 74], seed=90 + k) for k in range(2)]
    monkeypatch.setenv("UF3_DEBUG_LDS", "1")                      # (read when the context is made: the launches say what they are)
    monkeypatch.setattr(_lib, "_contexts", {})
    fz = process.BasisFeaturizer(basis)
    assert fz._dev()[1].featurizer_modes & 0x1000
    _, n3 = fz.neighbor_indices(dense)
    assert np.bincount(n3[:, 0]).max() > 16
    _, n3 = fz.neighbor_indices(frames[0])
    assert np.bincount(n3[:, 0]).max() <= 16
    ob = O.OracleBasis(basis)
    plain = fz.featurize_frames(frames)
    plain = fz.featurize_frames(frames)                            # (tuned: capacity 16, one launch)
    xd = fz.featurize_frames([dense])                              # the capacity grows
    ref = O.featurize(ob, dense)
    assert rel_err(xd[0][0], ref["xe"]) < TOL and rel_err(xd[1], ref["xf"]) < TOL
    capfd.readouterr()
    x_e, x_f, off = fz.featurize_frames(frames)
    said = capfd.readouterr().err
    assert "cap 16 (lists" in said and "selection 1" in said and "selection 2" in said, said
    for k, atoms in enumerate(frames):
        ref = O.featurize(ob, atoms)
        assert rel_err(x_e[k], ref["xe"]) < TOL and rel_err(x_f[off[k]:off[k + 1]], ref["xf"]) < TOL
    assert np.array_equal(x_f, plain[1])                           # the very same instance ran
    _lib.drop_device_basis(basis)


def test_three_species_wide_blocks():
    """ternary, lead 0: 18 trio blocks of 139/233 columns (several 64-column chunks, nsrc 1 and 2)."""
    d, meta, atoms = load_case("case_ternary24_slab")
    basis = basis_from_meta(meta)
    wide = synthetic.bspline.BSplineBasis(basis.chemical_system, leading_trim=0, trailing_trim=3,
                                          r_min_map=basis.r_min_map, r_max_map=basis.r_max_map,
                                          resolution_map=basis.resolution_map)
    assert wide.n_feats > 2000
    _check_against_oracle(wide, [atoms])


def test_evaluate_dataframe_surface():
    """tests/test_representation.py:539-603 of the reference: DataFrame in, MultiIndex frame out."""
    import pandas as pd
    cs = synthetic.composition.ChemicalSystem(['H', 'O'])
    basis = synthetic.bspline.BSplineBasis(cs)
    fz = process.BasisFeaturizer(basis)
    water = Atoms('H2O', positions=[[0, 0, 0], [3, 0.0, 0.0], [0, 4.0, 0]])
    df = pd.DataFrame({'geometry': [water, water], 'energy': [1.5, 1.5],
                       'fx': [[4, 3, 0], [4.1, 3.1, 0]], 'fy': [[0, 1, 2], [0, 1.1, 2.1]], 'fz': [[2, 1, 0], [2, 1, 0]]})
    out = fz.evaluate(df, 'geometry', 'energy', progress=False)
    assert len(out) == 2 * (1 + 3 * 3) and len(out.columns) == 1 + 2 + 18 * 3
    assert list(out.index[:4]) == [(0, 'energy'), (0, 'fx_0'), (0, 'fx_1'), (0, 'fx_2')]
    assert np.allclose(out['y'].to_numpy()[:10], [1.5, 4, 3, 0, 0, 1, 2, 2, 1, 0])
    assert out.loc[(0, 'energy'), 'n_H'] == 2 and out.loc[(0, 'energy'), 'n_O'] == 1
    ref = O.featurize(O.OracleBasis(basis), water)
    assert rel_err(out.loc[(1, 'energy')].to_numpy()[1:], ref["xe"]) < TOL
    assert rel_err(out.loc[(1, 'fy_2')].to_numpy()[1:], ref["xf"][2, 1]) < TOL
    # the reference's in-memory workflow: table -> tuples (per-atom energies) -> fit -> predictions
    x_e, y_e, x_f, y_f = ls.dataframe_to_tuples(out, n_elements=2)
    assert x_e.shape == (2, 56) and np.allclose(y_e, 0.5) and np.allclose(x_e[0, :2], [2 / 3, 1 / 3])
    assert rel_err(x_f[:9].reshape(3, 3, 56), ref["xf"].transpose(1, 0, 2)) < TOL         # rows fx_0..2, fy_0..2, fz_0..2
    model = ls.WeightedLinearModel(basis)
    model.fit(x_e, y_e, x_f, y_f, weight=0.5)
    y_e2, p_e, y_f2, p_f = ls.subset_prediction(out, model, subset_keys=[1], n_elements=2)
    assert len(p_e) == 1 and len(p_f) == 9 and np.all(np.isfinite(p_e)) and np.all(np.isfinite(p_f))


def test_batched_to_hdf_chunks_and_resume(monkeypatch, tmp_path):
    """``BasisFeaturizer.batched_to_hdf`` (process.py:256-291): chunk boundaries, table names, skipping the chunks a file already
    holds.  No PyTables in the image: the writer and the table listing are replaced by an in-memory store."""
    import pandas as pd
    basis = synthetic.notebook_basis(['W'])
    fz = process.BasisFeaturizer(basis)
    frames = [synthetic.lattice_frame("bcc", (2, 2, 2), 3.165, [74], 40 + k) for k in range(7)]
    rng = np.random.default_rng(2)
    df = pd.DataFrame({"geometry": frames, "energy": rng.normal(size=7), "fx": [rng.normal(size=16) for _ in frames],
                       "fy": [rng.normal(size=16) for _ in frames], "fz": [rng.normal(size=16) for _ in frames]},
                      index=[f"s{k}" for k in range(7)])
    store = {}
    monkeypatch.setattr(process, "save_feature_db", lambda d, filename, table_name="features": store.__setitem__(table_name, d))
    monkeypatch.setattr(process, "existing_feature_tables", lambda filename: sorted(store))
    path = tmp_path / "features.h5"
    fz.batched_to_hdf(str(path), df, batch_size=3)
    assert sorted(store) == ["features_000", "features_001", "features_002"]
    whole = fz.evaluate(df, progress=False)

    def same(a, b):      # (energy rows are summed with atomics: the last bits depend on the batch)
        return a.index.equals(b.index) and list(a.columns) == list(b.columns) and np.allclose(a.to_numpy(), b.to_numpy(), rtol=1e-12, atol=1e-12)

    assert same(pd.concat([store[k] for k in sorted(store)]), whole)
    assert [len(store[k].index.unique(level=0)) for k in sorted(store)] == [3, 3, 1]
    path.write_bytes(b"")                                       # the file exists now: chunks it holds are skipped
    del store["features_001"]
    kept = store["features_000"]
    with pytest.warns(RuntimeWarning):
        fz.batched_to_hdf(str(path), df, batch_size=3)
    assert sorted(store) == ["features_000", "features_001", "features_002"] and store["features_000"] is kept
    assert same(pd.concat([store[k] for k in sorted(store)]), whole)


def test_batched_evaluate_equals_frame_by_frame_evaluation():
    """``evaluate`` batches the frames of a table; rows, order and skip rules must be those of the reference's
    frame-by-frame loop over ``evaluate_configuration`` (process.py:121-194, 293-367)."""
    import pandas as pd
    basis = synthetic.notebook_basis(['Mo', 'W'])
    fz = process.BasisFeaturizer(basis)
    rng = np.random.default_rng(8)
    geoms = [synthetic.lattice_frame("bcc", (2, 2, 2), 3.165, [42, 74], seed=1),
             synthetic.lattice_frame("bcc", (2, 3, 2), 3.2, [42, 74], seed=2),
             Atoms('W2Mo', positions=[[0, 0, 0], [2.6, 0, 0], [0, 2.8, 0.3]]),                  # cluster
             synthetic.lattice_frame("bcc", (2, 2, 2), 3.165, [29], seed=3),                     # Cu: not in the basis
             synthetic.lattice_frame("bcc", (3, 2, 2), 3.1, [74], seed=4)]
    fcols = {c: [] for c in ('fx', 'fy', 'fz')}
    for k, g in enumerate(geoms):
        for c in fcols:
            f = rng.normal(0, 1, len(g))
            if k == 1:
                f[0] = np.nan                                                                    # forces unusable
            fcols[c].append(f.tolist())
    df = pd.DataFrame(dict(geometry=geoms, energy=[-10.0, -20.0, -1.0, -5.0, -30.0], **fcols),
                      index=["a", "b", "c", "d", "e"])
    with pytest.warns(RuntimeWarning, match="Invalid elements"):
        got = fz.evaluate(df, progress=False)
    expect = {}
    with pytest.warns(RuntimeWarning):
        for name, row in df.iterrows():
            forces = [row[c] for c in ('fx', 'fy', 'fz')]
            forces = None if np.any(np.isnan(np.asarray(forces, dtype=float))) else forces
            expect.update(fz.evaluate_configuration(row["geometry"], name, row["energy"], forces, "energy"))
    expect = fz.arrange_features_dataframe(expect)
    assert list(got.index) == list(expect.index) and list(got.columns) == list(expect.columns)
    assert ("d", "energy") not in got.index and ("b", "fx_0") not in got.index and ("b", "energy") in got.index
    assert np.array_equal(got["y"].to_numpy(), expect["y"].to_numpy())
    assert rel_err(got.to_numpy(), expect.to_numpy()) < 1e-13
    with pytest.warns(RuntimeWarning):
        assert len(fz.evaluate(df.iloc[[3]], progress=False)) == 0


def test_results_are_bitwise_repeatable():
    """No run-to-run drift: force rows, energies, forces and virials come out bit-identical on repeated calls, also
    when calls on frames of other sizes come in between (one wave owns an atom's sums; the evaluator's LDS adds keep
    their order).  Energy rows go through global atomics across workgroups: compared to 1e-13."""
    basis = synthetic.notebook_basis(['Mo', 'W'])
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(1).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    calc, fz = calculator.UFCalculator(model), process.BasisFeaturizer(basis)
    frames = [synthetic.lattice_frame("bcc", r, a, [42, 74], seed=k) for k, (r, a) in
              enumerate((((3, 3, 3), 3.165), ((6, 5, 4), 3.0), ((2, 2, 2), 3.3)))]
    for f in frames * 2:        # (capacities settled first: a context's very first calls run at the ESTIMATED list capacity, through
        calc.evaluate_frames([f], virial=True); fz.featurize_frames([f])     # other kernel instances -- same sums, other last bits)
    first = [(calc.evaluate_frames([f], virial=True), fz.featurize_frames([f])) for f in frames]
    for it in range(12):
        k = it % 3
        (e, f, _, v), (x_e, x_f, _) = calc.evaluate_frames([frames[k]], virial=True), fz.featurize_frames([frames[k]])
        (e0, f0, _, v0), (x_e0, x_f0, _) = first[k]
        assert np.array_equal(e, e0) and np.array_equal(f, f0) and np.array_equal(v, v0)
        assert np.array_equal(x_f, x_f0) and rel_err(x_e, x_e0) < 1e-13


def test_md_step_call_routes_agree_bit_for_bit(monkeypatch):
    """An MD-step call (a small cell through `uf3_eval*`) has shortcuts -- the one-workgroup cell list, kernels reading and
    writing the caller's pinned blocks, the host polling the result block for the call's sequence number instead of waiting
    for the stream -- each with a switch that restores the plain route: same bits every way, one frame or several, with
    and without the strain derivative, and equal to the oracle."""
    basis = synthetic.notebook_basis(['Mo', 'W'])
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(4).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    calc = calculator.UFCalculator(model)
    frames = [synthetic.lattice_frame("bcc", r, 3.165, [42, 74], seed=10 + k) for k, r in enumerate(((4, 4, 4), (3, 3, 2), (5, 4, 3)))]
    for batch in ([frames[0]], frames):
        for virial in (False, True):
            for _ in range(2):                               # (capacities tuned: the deferred-verdict route)
                ref = calc.evaluate_frames(batch, virial=virial)
            for switch in ("UF3_NO_TAIL_SPIN", "UF3_NO_ZERO_COPY", "UF3_NO_SMALL_PREPARE"):
                monkeypatch.setenv(switch, "1")
                got = calc.evaluate_frames(batch, virial=virial)
                monkeypatch.delenv(switch)
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), switch
                if virial:
                    assert np.array_equal(got[3], ref[3]), switch
            again = calc.evaluate_frames(batch, virial=virial)
            assert np.array_equal(again[0], ref[0]) and np.array_equal(again[1], ref[1])
    e, f, off = calc.evaluate_frames([frames[0]])
    e_o, f_o = O.evaluate(O.OracleBasis(basis), frames[0], coeff)
    assert abs(e[0] - e_o) <= 1e-11 * max(1.0, abs(e_o)) and worst_elementwise(f, f_o, 1e-9) <= 1.0


def _random_model(basis, seed):
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(seed).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    return model, coeff


@pytest.mark.parametrize("reps,steps", [((4, 4, 4), 50), ((6, 6, 6), 50), ((12, 12, 12), 24), ((25, 25, 20), 12)])
def test_md_route_on_a_displaced_trajectory_does_not_depend_on_rebuilds(reps, steps):
    """the evaluator's MD route (persistent superset lists with a skin, uf3_ctx_md_skin) on a seeded
    random walk that crosses several rebuilds -- every step bit-equal to the same route with the lists built from scratch at
    that step's positions (a step filters the lists by the true distances and takes the survivors in (species, supercell
    index) order, so when the lists were built cannot matter), equal to the plain rebuild-everything route to rounding, and
    to the oracle to 1e-9.  Sizes: the one-cell fast path (128 atoms), the one-workgroup cell list (432), the pinned staging
    block (3456), plain copies (25 000).  Reference: uf3/forcefield/calculator.py:124-153, 183-343 (every call from scratch)."""
    elements, numbers = (['Mo', 'W'], [42, 74]) if reps[0] <= 12 else (['V', 'Mo', 'W'], [23, 42, 74])
    basis = synthetic.notebook_basis(elements)
    model, coeff = _random_model(basis, 21)
    calc_md = calculator.UFCalculator(model, md_skin=0.4)
    calc_plain = calculator.UFCalculator(model, md_skin=0.0)
    start = synthetic.lattice_frame("bcc", reps, 3.165, numbers, seed=31)
    n = len(start)
    ctx = _lib.get_context(None)
    for _ in range(2):
        calc_plain.evaluate_frames([start])                      # (list capacity tuned: the MD route starts from a tuned context)
    walk = 0.03 if n < 10000 else 0.02
    rng = np.random.default_rng(5)
    path, pos = [], start.get_positions()
    for step in range(steps):
        pos = pos + rng.uniform(-walk, walk, (n, 3))
        if step in (steps // 2, steps // 2 + 1):
            pos[7] += [0.31, -0.2, 0.12]                          # one atom outruns skin / 2 in a single step: the call repeats itself
        path.append(pos.copy())                                   # (twice: were the first absorbed by a rebuild that was due anyway)

    def frame(step):
        return Atoms(numbers=start.get_atomic_numbers(), positions=path[step], cell=start.get_cell(), pbc=True)

    # the walk on lists that live across steps
    ctx.md_skin(0.0)
    before = ctx.md_stats()
    trace = [calc_md.evaluate_frames([frame(step)], virial=step % 5 == 4) for step in range(steps)]
    after = ctx.md_stats()
    builds, served, redone = (after[k] - before[k] for k in ("builds", "steps", "redone"))
    assert 3 <= builds < steps and redone >= 1 and served >= steps, (builds, served, redone)
    # every step against lists built from scratch at that step's positions, the plain route, the oracle
    ob = O.OracleBasis(basis)
    for step in range(steps):
        atoms, virial, got = frame(step), step % 5 == 4, trace[step]
        hot = step in (steps // 2, steps // 2 + 1)
        if step % 2 == 0 or hot:
            ctx.md_skin(0.0)                                      # (changing the skin drops the lists: the next call builds them)
            fresh = calc_md.evaluate_frames([atoms], virial=virial)
            assert np.array_equal(got[0], fresh[0]) and np.array_equal(got[1], fresh[1]), step
            if virial:
                assert np.array_equal(got[3], fresh[3]), step
        if step % 6 == 0 or hot:
            plain = calc_plain.evaluate_frames([atoms], virial=virial)
            assert abs(got[0][0] - plain[0][0]) <= 1e-12 * max(1.0, abs(plain[0][0])) and rel_err(got[1], plain[1]) < 1e-12
            if virial:
                assert rel_err(got[3], plain[3]) < 1e-11
        if (step % 12 == 0 or hot) and n <= 4000:
            e_o, f_o = O.evaluate(ob, atoms, coeff)
            assert abs(got[0][0] - e_o) <= 1e-11 * max(1.0, abs(e_o)) and worst_elementwise(got[1], f_o, 1e-9) <= 1.0
    ctx.md_skin(0.0)


def test_blocks_of_centres_on_the_md_route_add_up_to_the_frame():
    """uf3_eval_centres with a skin (round 5): every rank keeps whole-frame lists across steps; the shares of disjoint blocks of
    centres add up to the whole-frame result and to the oracle on every step of a walk that outruns the lists more than once,
    rows of atoms far from a block stay zero, and an atom OUTSIDE the block that moves too far is noticed by the block's rank."""
    from uf3_amd import parallel
    basis = synthetic.notebook_basis(['Mo', 'W'])
    model, coeff = _random_model(basis, 23)
    calc_md = calculator.UFCalculator(model, md_skin=0.4)
    calc_plain = calculator.UFCalculator(model, md_skin=0.0)
    start = synthetic.lattice_frame("bcc", (7, 6, 5), 3.165, [42, 74], seed=2)
    n = len(start)
    ctx = _lib.get_context(None)
    for _ in range(2):
        calc_plain.evaluate_frames([start], virial=True)
    rng = np.random.default_rng(9)
    pos = start.get_positions()
    ob = O.OracleBasis(basis)
    ctx.md_skin(0.0)
    before = ctx.md_stats()
    for step in range(10):
        pos = pos + rng.uniform(-0.04, 0.04, (n, 3))
        if step == 6:
            pos[n - 1] += [0.3, 0.1, -0.2]                        # an atom of the LAST block: the first block's rank must notice
        atoms = Atoms(numbers=start.get_atomic_numbers(), positions=pos, cell=start.get_cell(), pbc=True)
        for w in ((3, 8) if step % 3 == 0 else (3,)):
            cs = [calc_md.evaluate_centre_range(atoms, *parallel.shard_range(n, r, w), virial=True) for r in range(w)]
            e_sum, f_sum, v_sum = sum(s[0] for s in cs), sum(s[1] for s in cs), sum(s[2] for s in cs)
            if step % 3 == 0 or step == 6:
                e_o, f_o = O.evaluate(ob, atoms, coeff)
                assert abs(e_sum - e_o) <= 1e-10 * abs(e_o) and worst_elementwise(f_sum, f_o) <= 1.0
            for r, (_, fs, _) in enumerate(cs):
                lo, hi = parallel.shard_range(n, r, w)
                touched = np.flatnonzero(np.abs(fs).sum(axis=1) > 0)
                assert set(range(lo, hi)) <= set(touched.tolist()) and (len(touched) < n or w == 3)
        whole = calc_md.evaluate_frames([atoms], virial=True)
        assert abs(e_sum - whole[0][0]) <= 1e-12 * abs(whole[0][0]) and rel_err(f_sum, whole[1]) < 1e-12 and rel_err(v_sum, whole[3][0]) < 1e-11
    after = ctx.md_stats()
    assert 2 <= after["builds"] - before["builds"] < 10 and after["redone"] - before["redone"] >= 1
    ctx.md_skin(0.0)


@pytest.mark.parametrize("case", ["tiny_cell", "slab", "cluster", "ragged_batch", "triclinic"])
def test_md_route_on_small_cells_slabs_clusters_and_batches(case):
    """The MD route where a list holds several images of one neighbour (a 16-atom cell, 6.3 A across, against r_cut + skin = 5.9 A),
    with open boundaries (slab, cluster), on a skewed cell and on a batch of frames of different sizes: a short walk, every step
    against the oracle and the plain route, lists reused across steps."""
    rng = np.random.default_rng(12)
    els, nums = ['Mo', 'W'], [42, 74]
    basis = synthetic.notebook_basis(els)
    model, coeff = _random_model(basis, 24)
    calc_md = calculator.UFCalculator(model, md_skin=0.4)
    calc_plain = calculator.UFCalculator(model, md_skin=0.0)
    if case == "tiny_cell":
        frames = [synthetic.lattice_frame("bcc", (2, 2, 2), 3.165, nums, seed=81)]
    elif case == "slab":
        a = synthetic.lattice_frame("bcc", (4, 4, 3), 3.165, nums, seed=82)
        frames = [Atoms(numbers=a.get_atomic_numbers(), positions=a.get_positions(), cell=a.get_cell(), pbc=[True, True, False])]
    elif case == "cluster":
        a = synthetic.lattice_frame("bcc", (3, 3, 3), 3.165, nums, seed=83)
        frames = [Atoms(numbers=a.get_atomic_numbers(), positions=a.get_positions(), cell=a.get_cell(), pbc=False)]
    elif case == "triclinic":
        a = synthetic.lattice_frame("bcc", (3, 3, 3), 3.165, nums, seed=84)
        shear = np.eye(3) + np.array([[0, 0.18, 0.07], [0, 0, -0.12], [0, 0, 0]])
        frames = [Atoms(numbers=a.get_atomic_numbers(), positions=a.get_positions() @ shear, cell=np.asarray(a.get_cell()) @ shear, pbc=True)]
    else:
        frames = [synthetic.lattice_frame("bcc", r, 3.165, nums, seed=85 + k) for k, r in enumerate(((3, 3, 3), (2, 2, 2), (4, 3, 3)))]
    ctx = _lib.get_context(None)
    for _ in range(2):
        calc_plain.evaluate_frames(frames)
    ob = O.OracleBasis(basis)
    ctx.md_skin(0.0)
    s0 = ctx.md_stats()
    trace, path = [], []
    for step in range(8):
        frames = [Atoms(numbers=f.get_atomic_numbers(), positions=f.get_positions() + rng.uniform(-0.03, 0.03, (len(f), 3)),
                        cell=f.get_cell(), pbc=f.get_pbc()) for f in frames]
        path.append(frames)
        trace.append(calc_md.evaluate_frames(frames, virial=step % 2 == 1))
    s1 = ctx.md_stats()
    assert s1["steps"] - s0["steps"] >= 8 and 1 <= s1["builds"] - s0["builds"] < 8
    for step, frames in enumerate(path):
        got = trace[step]
        plain = calc_plain.evaluate_frames(frames, virial=step % 2 == 1)
        assert np.allclose(got[0], plain[0], rtol=1e-12, atol=1e-12) and rel_err(got[1], plain[1]) < 1e-12
        if step % 2 == 1:
            assert rel_err(got[3], plain[3]) < 1e-11
        off = got[2]
        for k, f in enumerate(frames):
            e_o, f_o = O.evaluate(ob, f, coeff)
            assert abs(got[0][k] - e_o) <= 1e-11 * max(1.0, abs(e_o)) and worst_elementwise(got[1][off[k]:off[k + 1]], f_o, 1e-9) <= 1.0
    ctx.md_skin(0.0)


def test_md_route_rebuilds_on_layout_species_and_cell_changes():
    """The lists are tied to (basis, offsets, cells, pbc, species): any change rebuilds them instead of serving stale neighbours;
    a batch of several frames runs on lists as well.  Each result against the plain route."""
    basis = synthetic.notebook_basis(['Mo', 'W'])
    model, coeff = _random_model(basis, 22)
    calc_md = calculator.UFCalculator(model, md_skin=0.5)
    calc_plain = calculator.UFCalculator(model, md_skin=0.0)
    frames = [synthetic.lattice_frame("bcc", r, 3.165, [42, 74], seed=40 + k) for k, r in enumerate(((4, 4, 4), (3, 3, 2), (5, 4, 3)))]
    ctx = _lib.get_context(None)
    for _ in range(2):
        calc_plain.evaluate_frames(frames)

    cases = []                                                      # (the plain route afterwards: its calls would drop the lists)

    def md(batch, virial=False):
        cases.append((batch, virial, calc_md.evaluate_frames(batch, virial=virial)))

    ctx.md_skin(0.0)
    s0 = ctx.md_stats()
    md(frames); md(frames, True); md(frames)
    s1 = ctx.md_stats()
    assert s1["builds"] - s0["builds"] == 1 and s1["steps"] - s0["steps"] >= 3
    md([frames[0]])                                                 # another layout
    assert ctx.md_stats()["builds"] == s1["builds"] + 1
    z = frames[0].get_atomic_numbers()
    z[5] = 42 if z[5] == 74 else 74
    swapped = Atoms(numbers=z, positions=frames[0].get_positions(), cell=frames[0].get_cell(), pbc=True)
    md([swapped])                                                   # a species changed: found on the device, the call repeats on new lists
    s2 = ctx.md_stats()
    assert s2["builds"] == s1["builds"] + 2 and s2["redone"] >= s1["redone"] + 1
    strained = Atoms(numbers=z, positions=swapped.get_positions() * 1.01, cell=np.asarray(swapped.get_cell()) * 1.01, pbc=True)
    md([strained], True)                                            # the cell changed
    assert ctx.md_stats()["builds"] == s2["builds"] + 1
    wrapped = strained.get_positions()
    wrapped[3] += np.asarray(strained.get_cell())[0]                # an atom re-wrapped by a whole lattice vector: far beyond the skin
    md([Atoms(numbers=z, positions=wrapped, cell=strained.get_cell(), pbc=True)])
    assert ctx.md_stats()["redone"] >= s2["redone"] + 1
    with pytest.raises(_lib.SpeciesError):
        z2 = z.copy(); z2[0] = 29
        calc_md.evaluate_frames([Atoms(numbers=z2, positions=wrapped, cell=strained.get_cell(), pbc=True)])
    md([strained])
    for batch, virial, a in cases:
        b = calc_plain.evaluate_frames(batch, virial=virial)
        assert np.allclose(a[0], b[0], rtol=1e-12, atol=1e-12) and rel_err(a[1], b[1]) < 1e-12
        if virial:
            assert rel_err(a[3], b[3]) < 1e-11
    ctx.md_skin(0.0)


def _table_fit_case():
    import pandas as pd
    t = np.load(os.path.join(GOLDEN, "table_fit.npz"), allow_pickle=False)
    basis = basis_from_meta(json.loads(str(t["meta"])))
    columns = basis.get_column_names()
    tables = [pd.DataFrame(t[f"table{k}"], columns=columns, index=pd.MultiIndex.from_tuples(
        list(zip(t[f"table{k}_names"].tolist(), t[f"table{k}_keys"].tolist())))) for k in range(3)]
    weights = dict(zip(t["weight_names"].tolist(), t["weight_values"].tolist()))
    return t, basis, tables, weights


def test_fit_from_feature_tables_against_reference_capture():
    """The from-file workflow (gram_from_df per table, streamed statistics, one solve; batched prediction) against
    the reference's functions run table by table (tests/golden/make_table_fit_golden.py)."""
    t, basis, tables, weights = _table_fit_case()
    subset = t["subset"].tolist()
    model = ls.WeightedLinearModel(basis, regularizer=t["regularizer"])
    e_var, f_var = ls.VarianceRecorder(), ls.VarianceRecorder()
    for k, df in enumerate(tables):
        keys = df.index.unique(level=0).intersection(subset)
        pieces = model.gram_from_df(df, keys, e_variance=e_var, f_variance=f_var, sample_weights=weights)
        for got, name in zip(pieces, ("gram_e", "gram_f", "ord_e", "ord_f")):
            assert rel_err(got, t[f"{name}{k}"]) < 1e-12, (name, k)
    assert np.allclose([e_var.mean, e_var.std, e_var.n], t["e_stats"], rtol=1e-13)
    assert np.allclose([f_var.mean, f_var.std, f_var.n], t["f_stats"], rtol=1e-13)
    model.fit_from_tables(tables, subset, weight=float(t["kappa"][0]), sample_weights=weights)
    assert np.allclose(model.coefficients, t["coefficients"], rtol=1e-7, atol=1e-9)
    assert np.array_equal(model.data_coverage, t["data_coverage"])
    model.coefficients = t["coefficients"]
    y_e, p_e, y_f, p_f = model.batched_predict(keys=["a1", "b0", "c2", "c3"], score=False, tables=tables)
    assert np.array_equal(y_e, t["pred_y_e"]) and np.allclose(p_e, t["pred_p_e"], rtol=1e-13)
    assert np.array_equal(y_f, t["pred_y_f"]) and np.allclose(p_f, t["pred_p_f"], rtol=1e-13, atol=1e-14)
    assert len(model.batched_predict(keys=["a1"], score=True, tables=tables)) == 6
    # the stand-alone solvers: same normal equations as numpy's least squares
    rng = np.random.default_rng(3)
    x, y, w = rng.normal(size=(200, 12)), rng.normal(size=200), rng.uniform(0.1, 2.0, 200)
    assert np.allclose(ls.linear_least_squares(x, y), np.linalg.lstsq(x, y, rcond=None)[0], rtol=1e-9)
    reg = 0.1 * np.eye(12)
    sw = np.sqrt(w)
    expect = np.linalg.lstsq(np.vstack([x * sw[:, None], reg]), np.concatenate([y * sw, np.zeros(12)]), rcond=None)[0]
    assert np.allclose(ls.weighted_least_squares(x, y, weights=w, regularizer=reg), expect, rtol=1e-9)


def test_file_wrappers_of_the_table_fit(monkeypatch, tmp_path):
    """``fit_from_file(filename)`` / ``batched_predict(filename)``: the image has neither PyTables nor h5py, so the two
    functions that touch the file (``hdf_table_names``, ``pd_read_hdf``) are replaced by an in-memory store here; everything
    the wrappers do themselves -- existence check, table order, hand-over to the table loop -- runs."""
    t, basis, tables, weights = _table_fit_case()
    store = {f"features_{k:03d}": df for k, df in enumerate(tables)}
    path = tmp_path / "features.h5"
    path.write_bytes(b"")
    reads = []
    monkeypatch.setattr(ls, "hdf_table_names", lambda filename: sorted(store) if str(filename) == str(path) else [])
    monkeypatch.setattr(ls, "pd_read_hdf", lambda filename, name: (reads.append(name), store[name])[1])
    subset = t["subset"].tolist()
    model = ls.WeightedLinearModel(basis, regularizer=t["regularizer"])
    model.fit_from_file(str(path), subset, weight=float(t["kappa"][0]), sample_weights=weights)
    assert reads == sorted(store)
    assert np.allclose(model.coefficients, t["coefficients"], rtol=1e-7, atol=1e-9)
    assert np.array_equal(model.data_coverage, t["data_coverage"])
    model.coefficients = t["coefficients"]
    y_e, p_e, y_f, p_f = model.batched_predict(str(path), keys=["a1", "b0", "c2", "c3"], score=False)
    assert np.array_equal(y_e, t["pred_y_e"]) and np.allclose(p_e, t["pred_p_e"], rtol=1e-13)
    assert np.array_equal(y_f, t["pred_y_f"]) and np.allclose(p_f, t["pred_p_f"], rtol=1e-13, atol=1e-14)
    with pytest.raises(FileNotFoundError):
        model.fit_from_file(str(tmp_path / "missing.h5"), subset)


def test_energy_only_and_forces_only_modes_agree():
    atoms, basis = synthetic.config_c2()
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, _ = fz.featurize_frames([atoms])
    e_only = fz.featurize_frames([atoms], forces=False)[0]
    f_only = fz.featurize_frames([atoms], energy=False)[1]
    assert rel_err(e_only, x_e) < 1e-12 and rel_err(f_only, x_f) < 1e-12


def test_evaluator_50k_atom_ternary():
    """configs[4]: 3-element 50k-atom cell, energy/forces of a random smooth model vs the oracle."""
    atoms = synthetic.lattice_frame("bcc", (25, 25, 40), 3.165, [23, 42, 74], seed=4000)
    assert len(atoms) == 50000
    basis = synthetic.notebook_basis(['V', 'Mo', 'W'])
    model = ls.WeightedLinearModel(basis)
    rng = np.random.default_rng(11)
    coeff = rng.normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    calc = calculator.UFCalculator(model)
    e, f, _ = calc.evaluate_frames([atoms])
    e_ref, f_ref = O.evaluate(O.OracleBasis(basis), atoms, coeff)
    assert abs(e[0] - e_ref) <= 1e-9 * abs(e_ref)
    assert rel_err(f, f_ref) < 1e-9                       # north_star: forces within 1e-6 of the CPU path
    assert worst_elementwise(f, f_ref) <= 1.0             # ... literally: each of the 150 000 components to 1e-9 of itself
    x_e = process.BasisFeaturizer(basis).featurize_frames([atoms], forces=False)[0]
    assert abs(x_e[0] @ coeff - e[0]) <= 1e-10 * abs(e[0])


def test_evaluator_on_the_ragged_binary_fcc_frame():
    """configs[2]'s workload through the EVALUATOR: Ne-Xe fcc, r_max 4.5 / 4.5 / 9.0, ragged neighbour counts (the evaluator
    is otherwise checked on bcc W / W-Mo / ternary cells only); energy, forces and the strain derivative vs the oracle."""
    atoms, basis = synthetic.config_c3()
    coeff = np.random.default_rng(23).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model = ls.WeightedLinearModel(basis)
    model.coefficients = coeff
    e, f, _, v = calculator.UFCalculator(model).evaluate_frames([atoms], virial=True)
    e_ref, f_ref = O.evaluate(O.OracleBasis(basis), atoms, coeff)
    assert abs(e[0] - e_ref) <= TOL * abs(e_ref)
    assert rel_err(f, f_ref) < TOL and worst_elementwise(f, f_ref) <= 1.0
    assert np.abs(f.sum(axis=0)).max() < 1e-9 * np.abs(f).max() * len(atoms) ** 0.5
    x_e, x_f, _ = process.BasisFeaturizer(basis).featurize_frames([atoms])
    assert abs(x_e[0] @ coeff - e[0]) <= 1e-10 * abs(e[0]) and rel_err(x_f @ coeff, f) < TOL     # E = x_e c, F = X_f c (the force rows carry the sign)


def test_gram_of_a_single_column():
    """n_feat = 1 (advisor, round 2): every Gram kernel's tile logic at its smallest, X^T X and X^T y as plain dot products"""
    rng = np.random.default_rng(9)
    for rows in (1, 7, 4097, 70001):
        x, y = rng.normal(size=(rows, 1)), rng.normal(size=rows)
        g, o = ls.gram_device(x, y)
        assert g.shape == (1, 1) and o.shape == (1,)
        assert np.isclose(g[0, 0], float(x[:, 0] @ x[:, 0]), rtol=1e-12) and np.isclose(o[0], float(x[:, 0] @ y), rtol=1e-10, atol=1e-10)


def test_atom_range_shares_add_up_to_the_frame():
    """uf3_eval_atoms: shares of disjoint atom blocks (what each rank of a decomposed frame computes) sum to
    the whole-frame energy / virial, and their force rows are the whole-frame rows (the whole frame takes the
    two-pass route -- every triplet once, at its centre -- and a block the gather route: two independent traversals)."""
    from uf3_amd import parallel
    atoms = synthetic.lattice_frame("bcc", (5, 6, 7), 3.165, [42, 74], seed=77)
    basis = synthetic.notebook_basis(['Mo', 'W'])
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(5).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    calc = calculator.UFCalculator(model, md_skin=float(os.environ.get("UF3_MD_SKIN", "0") or 0))     # (the plain routes against each other: no lists unless forced)
    for _ in range(2):                     # (capacities settled: a context's first call runs other kernel instances, same sums, other last bits)
        calc.evaluate_frames([atoms], virial=True)
    e, f, _, v = calc.evaluate_frames([atoms], virial=True)
    n, world = len(atoms), 3
    assert n == 420
    shares = [calc.evaluate_atom_range(atoms, *parallel.shard_range(n, r, world), virial=True) for r in range(world)]
    # the shares against the ORACLE's evaluator directly (not only against the product's other route)
    e_ref, f_ref = O.evaluate(O.OracleBasis(basis), atoms, coeff)
    assert abs(sum(s[0] for s in shares) - e_ref) <= 1e-10 * abs(e_ref)
    assert rel_err(sum(s[1] for s in shares), f_ref) < 1e-10 and worst_elementwise(sum(s[1] for s in shares), f_ref) <= 1.0
    for r, (_, fs, _) in enumerate(shares):
        lo, hi = parallel.shard_range(n, r, world)
        assert rel_err(fs[lo:hi], f_ref[lo:hi]) < 1e-10
    assert abs(sum(s[0] for s in shares) - e[0]) <= 1e-12 * abs(e[0])
    assert rel_err(sum(s[2] for s in shares), v[0]) < 1e-12
    for r, (_, fs, _) in enumerate(shares):
        lo, hi = parallel.shard_range(n, r, world)
        assert rel_err(fs[lo:hi], f[lo:hi]) < 1e-12
        assert not fs[:lo].any() and not fs[hi:].any()
    # blocks of CENTRES (uf3_eval_centres: every triplet once, at its centre; what it puts on atoms of other blocks comes back
    # in the same array): the shares add up to the oracle's and to the whole-frame results, rows far from the block are zero
    for w in (3, 8):
        cs = [calc.evaluate_centre_range(atoms, *parallel.shard_range(n, r, w), virial=True) for r in range(w)]
        assert abs(sum(s[0] for s in cs) - e_ref) <= 1e-10 * abs(e_ref) and abs(sum(s[0] for s in cs) - e[0]) <= 1e-12 * abs(e[0])
        assert rel_err(sum(s[1] for s in cs), f_ref) < 1e-10 and worst_elementwise(sum(s[1] for s in cs), f_ref) <= 1.0
        assert rel_err(sum(s[1] for s in cs), f) < 1e-12 and rel_err(sum(s[2] for s in cs), v[0]) < 1e-12
        for r, (es, fs, _) in enumerate(cs):
            lo, hi = parallel.shard_range(n, r, w)
            assert abs(es - shares[r][0]) <= 1e-12 * abs(es) if w == 3 else True       # the same centres' energies
            touched = np.flatnonzero(np.abs(fs).sum(axis=1) > 0)
            assert set(range(lo, hi)) <= set(touched.tolist())
            assert len(touched) < n or w == 3                                            # (a halo, not the whole frame)
    cs2 = [calc.evaluate_centre_range(atoms, *parallel.shard_range(n, r, 3), virial=True) for r in range(3)]   # repeatable
    assert all(np.array_equal(a[1], b2[1]) and a[0] == b2[0] for a, b2 in zip(cs2, [calc.evaluate_centre_range(atoms, *parallel.shard_range(n, r, 3), virial=True) for r in range(3)]))
    e0c, f0c, _ = calc.evaluate_centre_range(atoms, 10, 10)
    assert e0c == 0.0 and not f0c.any()
    os.environ["UF3_EVAL_GATHER"] = "1"                                       # whole frame through the gather route
    try:
        e_g, f_g, _, v_g = calc.evaluate_frames([atoms], virial=True)
    finally:
        del os.environ["UF3_EVAL_GATHER"]
    # (to rounding, not to the bit: the two routes are different kernels since round 5 -- the centre pass takes its centre legs
    # from per-bond tables and leg n's knot records from LDS, and the compiler contracts the two spline evaluations differently)
    assert abs(e_g[0] - e[0]) <= 1e-13 * abs(e[0]) and rel_err(v_g, v) < 1e-13 and rel_err(f_g, f) < 1e-12
    os.environ["UF3_SEPARATE_N3"] = "1"                                       # lists from k_build_n3 instead of the
    try:                                                                      # centre pass's own walk: same lists
        e_s, f_s, _, v_s = calc.evaluate_frames([atoms], virial=True)
        big = [atoms] * 40                                                    # results > the pinned-staging limit:
        e_b, f_b, off_b = calc.evaluate_frames(big)                           # the non-deferred flag check
    finally:
        del os.environ["UF3_SEPARATE_N3"]
    if float(os.environ.get("UF3_MD_SKIN", "0") or 0) > 0:     # (the whole suite on the MD route: `e, f, v` came from the lists, whose
        assert rel_err(f_s, f) < 1e-12 and rel_err(v_s, v) < 1e-12 and abs(e_s[0] - e[0]) <= 1e-12 * abs(e[0])    # sums run in list order)
    else:
        assert np.array_equal(e_s, e) and np.array_equal(f_s, f) and np.array_equal(v_s, v)
    e_b2, f_b2, _ = calc.evaluate_frames(big)
    if float(os.environ.get("UF3_MD_SKIN", "0") or 0) > 0:     # (forced lists: e_b2 ran on them, e_b -- separate list build -- could not)
        assert np.allclose(e_b, e_b2, rtol=1e-12) and rel_err(f_b, f_b2) < 1e-12 and rel_err(f_b[off_b[7]:off_b[8]], f) < 1e-12
    else:
        assert np.array_equal(e_b, e_b2) and np.array_equal(f_b, f_b2) and np.array_equal(f_b[off_b[7]:off_b[8]], f)
    for r, (_, fs, _) in enumerate(shares):
        lo, hi = parallel.shard_range(n, r, world)
        assert np.array_equal(fs[lo:hi], f_g[lo:hi])
    e1, f1, v1 = parallel.sharded_evaluate(calc, atoms, virial=True)          # no process group: whole frame
    assert e1 == e[0] and np.array_equal(f1, f) and np.array_equal(v1, v[0])
    e0, f0, v0 = calc.evaluate_atom_range(atoms, 10, 10)                      # empty block
    assert e0 == 0.0 and not f0.any() and v0 is None
    with pytest.raises(RuntimeError):
        calc.evaluate_atom_range(atoms, 5, n + 1)


def test_evaluator_list_capacity_regrows_between_calls():
    """Small batches read the neighbour stage's flags together with the results: a frame denser than any seen
    before overflows the remembered list capacity, and the call must notice and repeat itself."""
    basis = synthetic.notebook_basis(['W'])
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(9).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    calc = calculator.UFCalculator(model)
    ob = O.OracleBasis(basis)
    for a0 in (3.6, 3.165, 2.7, 3.3, 2.5):                 # neighbours inside the 3-body range: few -> many
        atoms = synthetic.lattice_frame("bcc", (3, 3, 4), a0, [74], seed=int(a0 * 1000))
        e, f, _, v = calc.evaluate_frames([atoms], virial=True)
        e_ref, f_ref = O.evaluate(ob, atoms, coeff)
        assert abs(e[0] - e_ref) <= 1e-10 * max(1.0, abs(e_ref)), a0
        assert rel_err(f, f_ref) < 1e-10, a0
    with pytest.raises(RuntimeError):                      # the species error takes the same deferred route
        bad = synthetic.lattice_frame("bcc", (3, 3, 4), 3.165, [29], seed=1)
        calc.evaluate_frames([bad])
    atoms = synthetic.lattice_frame("bcc", (3, 3, 4), 3.165, [74], seed=2)
    e, f, _ = calc.evaluate_frames([atoms])                # ... and leaves the context usable
    e_ref, f_ref = O.evaluate(ob, atoms, coeff)
    assert abs(e[0] - e_ref) <= 1e-10 * abs(e_ref) and rel_err(f, f_ref) < 1e-10


def test_evaluator_list_capacity_regrows_in_big_batches():
    """Same as above for batches whose results do not go through the pinned staging block (the flags are then read
    at the end of the launch sequence and the whole sequence repeated), on a fresh context so that the remembered
    capacity is the small one of the first call."""
    saved = dict(_lib._contexts)
    _lib._contexts.clear()
    try:
        basis = synthetic.notebook_basis(['W'])
        model = ls.WeightedLinearModel(basis)
        coeff = np.random.default_rng(9).normal(0, 0.05, basis.n_feats)
        coeff[basis.col_idx] = 0.0
        model.coefficients = coeff
        calc = calculator.UFCalculator(model)
        calc.evaluate_frames([synthetic.lattice_frame("bcc", (3, 3, 4), 3.6, [74], seed=1)])     # tunes a small capacity
        dense = [synthetic.lattice_frame("bcc", (3, 3, 4), 2.6, [74], seed=100 + k) for k in range(320)]
        e, f, off = calc.evaluate_frames(dense)                     # 23 040 atoms: 553 KB of forces
        ob = O.OracleBasis(basis)
        for k in (0, 319):
            e_ref, f_ref = O.evaluate(ob, dense[k], coeff)
            assert abs(e[k] - e_ref) <= 1e-10 * abs(e_ref) and rel_err(f[off[k]:off[k + 1]], f_ref) < 1e-10
    finally:
        _lib._contexts.clear()
        _lib._contexts.update(saved)


def test_degenerate_frames_in_a_batch():
    """Frames with nothing to do next to a normal one: a lone atom, two atoms beyond every cut-off, an empty frame.
    Their energy rows hold the composition only, their force rows and forces are zero, and the normal frame is not
    disturbed (its rows equal those of a batch of its own)."""
    basis = synthetic.notebook_basis(['Mo', 'W'])
    fz = process.BasisFeaturizer(basis)
    big = 40.0 * np.eye(3)
    lone = Atoms('W', positions=[[1.0, 2.0, 3.0]], cell=big, pbc=True)
    apart = Atoms('MoW', positions=[[0, 0, 0], [15.0, 0, 0]], cell=big, pbc=True)
    empty = Atoms(numbers=[], positions=np.zeros((0, 3)), cell=big, pbc=True)
    normal = synthetic.lattice_frame("bcc", (2, 2, 3), 3.165, [42, 74], seed=6)
    frames = [lone, apart, empty, normal, lone]
    x_e, x_f, off = fz.featurize_frames(frames)
    assert list(np.diff(off)) == [1, 2, 0, len(normal), 1]
    expect = np.zeros((5, x_e.shape[1]))
    expect[0, 1] = expect[4, 1] = 1.0                    # columns 0, 1: n_Mo, n_W
    expect[1, 0] = expect[1, 1] = 1.0
    x_e1, x_f1, _ = fz.featurize_frames([normal])
    expect[3] = x_e1[0]
    assert rel_err(x_e, expect) < 1e-13
    assert not x_f[:3].any() and not x_f[-1:].any() and rel_err(x_f[off[3]:off[4]], x_f1) < 1e-13
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(2).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    calc = calculator.UFCalculator(model)
    e, f, off_e, v = calc.evaluate_frames(frames, virial=True)
    e1, f1, _, v1 = calc.evaluate_frames([normal], virial=True)
    assert np.allclose(e, x_e @ coeff, rtol=1e-12, atol=1e-12) and e[2] == 0.0
    assert not f[:3].any() and not f[-1:].any() and not v[:3].any()
    assert rel_err(f[off_e[3]:off_e[4]], f1) < 1e-13 and rel_err(v[3], v1[0]) < 1e-13


def test_atoms_outside_the_cell(monkeypatch):
    """Positions are not wrapped by the reference (geometry.py:108-149 tiles them as given): an atom a few lattice
    vectors away keeps only the neighbours the finite image range reaches, and the 3-body FORCE rows additionally lose
    the terms of ghost-centred triplets whose third atom that supercell does not hold (so they stop being the gradient
    of the energy row).  The kernels reproduce all of it: every row equals the oracle's.  With UF3_KEEP_GHOST_TERMS the
    dropped terms are kept instead and the rows are the numerical gradient of the energy row again.  An atom hundreds
    of cells away is refused."""
    basis = synthetic.notebook_basis(['W'])
    fz = process.BasisFeaturizer(basis)
    atoms = synthetic.lattice_frame("bcc", (2, 2, 2), 3.165, [74], seed=12)
    pos = atoms.get_positions().copy()
    cell = np.array(atoms.get_cell()).reshape(3, 3)
    pos[3] += 1.0 * cell[0] - 2.0 * cell[2]
    pos[7] -= 1.0 * cell[1]

    def rows(p):
        return fz.featurize_frames([Atoms(numbers=atoms.get_atomic_numbers(), positions=p, cell=cell, pbc=True)])

    x_e, x_f, _ = rows(pos)
    ref = O.featurize(O.OracleBasis(basis), Atoms(numbers=atoms.get_atomic_numbers(), positions=pos, cell=cell, pbc=True))
    assert rel_err(x_e[0], ref["xe"]) < TOL and rel_err(x_f, ref["xf"]) < TOL
    # the same through the batched path (several frames per call, one of them with every atom inside its cell)
    inside = Atoms(numbers=atoms.get_atomic_numbers(), positions=atoms.get_positions(), cell=cell, pbc=True)
    y_e, y_f, off = fz.featurize_frames([inside, Atoms(numbers=atoms.get_atomic_numbers(), positions=pos, cell=cell, pbc=True)])
    ref_in = O.featurize(O.OracleBasis(basis), inside)
    assert rel_err(y_f[off[1]:off[2]], ref["xf"]) < TOL and rel_err(y_f[off[0]:off[1]], ref_in["xf"]) < TOL
    monkeypatch.setenv("UF3_KEEP_GHOST_TERMS", "1")
    k_e, k_f, _ = rows(pos)
    n2 = basis.n_feats - basis.partition_sizes[-1]                       # one-body + pair columns
    assert rel_err(k_e[0], ref["xe"]) < TOL and rel_err(k_f[:, :, :n2], ref["xf"][:, :, :n2]) < TOL
    assert rel_err(k_f, ref["xf"]) > 1e-6                                # (this frame does lose terms in the reference)
    h = 1e-5
    for a in (0, 3, 7, 9):
        for c in range(3):
            up, dn = pos.copy(), pos.copy()
            up[a, c] += h
            dn[a, c] -= h
            grad = -(rows(up)[0][0] - rows(dn)[0][0]) / (2 * h)
            assert np.abs(grad - k_f[a, c]).max() < 1e-6 * np.abs(k_f).max(), (a, c)
    monkeypatch.delenv("UF3_KEEP_GHOST_TERMS")
    # asynchronous entry: the ordinary 3-body launches leave such a batch to the image-range ones -- the verdict arrives
    # with the next synchronisation (RetryError), the repeated call is right, and a batch of wrapped atoms afterwards
    # brings the context back to the ordinary launches
    import torch
    ctx, db = fz._dev()
    dev = torch.device("cuda", ctx.device)
    wrapped = Atoms(numbers=atoms.get_atomic_numbers(), positions=atoms.get_positions(), cell=cell, pbc=True)
    rows(atoms.get_positions())                                           # (ordinary launches again)
    batch = _lib.FrameBatch([Atoms(numbers=atoms.get_atomic_numbers(), positions=pos, cell=cell, pbc=True)])
    d_pos, d_z = torch.from_numpy(batch.pos).to(dev), torch.from_numpy(batch.z).to(dev)
    d_xe = torch.zeros((1, db.n_feat), dtype=torch.float64, device=dev)
    d_xf = torch.zeros((batch.n_atoms, 3, db.n_feat), dtype=torch.float64, device=dev)
    prev = ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    try:
        fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), d_xe.data_ptr(), d_xf.data_ptr())
        with pytest.raises(_lib.RetryError):
            ctx.synchronize()
        fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), d_xe.data_ptr(), d_xf.data_ptr())
        ctx.synchronize()
    finally:
        ctx.restore_stream(prev)
    assert rel_err(d_xf.cpu().numpy(), ref["xf"]) < TOL
    assert rel_err(fz.featurize_frames([wrapped])[1], ref_in["xf"]) < TOL
    far = pos.copy()
    far[5] += 300.0 * cell[0]
    with pytest.raises(RuntimeError, match="outside the periodic cell"):
        rows(far)
    assert rel_err(rows(pos)[1], x_f) < 1e-13                             # the context is usable afterwards


def test_featurize_frames_into_caller_buffers():
    atoms, basis = synthetic.config_c2()
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, _ = fz.featurize_frames([atoms])
    buf_e, buf_f = np.full_like(x_e, np.nan), np.full_like(x_f, np.nan)
    y_e, y_f, _ = fz.featurize_frames([atoms], out=(buf_e, buf_f))
    assert y_e is buf_e and y_f is buf_f and rel_err(buf_e, x_e) < 1e-12 and rel_err(buf_f, x_f) < 1e-12
    with pytest.raises(ValueError):
        fz.featurize_frames([atoms], out=(buf_e, buf_f[:, :2]))


@pytest.mark.parametrize("lead3,env", [(3, {}), (0, {}), (3, {"UF3_NO_FEAT3": "1"}), (3, {"UF3_NO_FEAT3": "1", "UF3_NO_MFMA_FEAT": "1"})])
def test_force_rows_with_a_row_stride(lead3, env, monkeypatch):
    """uf3_featurize_ld_dev: force rows `ld` doubles apart (every row on 128-byte lines of its own) hold the dense rows' values
    in their first F columns and leave the padding alone -- through the bond-factorised launch, its wide-window instance and the
    matrix-core / generic launches."""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    basis = synthetic.notebook_basis(['Mo', 'W'], lead3=lead3)
    frames = [synthetic.lattice_frame("bcc", (4, 4, 3), 3.165, [42, 74], seed=90 + k) for k in range(3)]
    fz = process.BasisFeaturizer(basis)
    ctx, db = fz._dev()
    dev = torch.device("cuda", ctx.device)
    batch = _lib.FrameBatch(frames)
    F, n = db.n_feat, batch.n_atoms
    ld = fz.aligned_ld(F)
    assert ld % 16 == 0 and F <= ld < F + 16
    d_pos, d_z = torch.from_numpy(batch.pos).to(dev), torch.from_numpy(batch.z).to(dev)
    prev = ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    try:
        dense_e = torch.empty((len(frames), F), dtype=torch.float64, device=dev)
        dense = torch.empty((n, 3, F), dtype=torch.float64, device=dev)
        wide_e = torch.empty_like(dense_e)
        wide = torch.full((n, 3, ld), -7.25, dtype=torch.float64, device=dev)
        for _ in range(2):                                                    # (capacities settled: the launches of the steady state)
            fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), dense_e.data_ptr(), dense.data_ptr())
            ctx.synchronize()
            fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), wide_e.data_ptr(), wide.data_ptr(), ld=ld)
            ctx.synchronize()
        with pytest.raises(_lib.UF3Error):
            fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), wide_e.data_ptr(), wide.data_ptr(), ld=F - 1)
    finally:
        ctx.restore_stream(prev)
    assert torch.equal(wide[:, :, :F], dense) and bool((wide[:, :, F:] == -7.25).all())
    assert rel_err(wide_e.cpu().numpy(), dense_e.cpu().numpy()) < 1e-13
    x_e, x_f, _ = fz.featurize_frames(frames)
    assert rel_err(dense.cpu().numpy(), x_f) < 1e-13


def test_featurize_frames_chunking_is_transparent():
    basis = synthetic.notebook_basis(['W'])
    frames = [synthetic.lattice_frame("bcc", (2, 2, 2 + (k % 2)), 3.165, [74], seed=70 + k) for k in range(7)]
    fz = process.BasisFeaturizer(basis)
    a_e, a_f, off = fz.featurize_frames(frames)
    b_e, b_f, off2 = fz.featurize_frames(frames, max_bytes=40 * 24 * basis.n_feats)      # ~2 frames per chunk
    assert np.array_equal(off, off2) and rel_err(b_e, a_e) < 1e-12 and rel_err(b_f, a_f) < 1e-12


def _fresh_rows(basis, frames, **env):
    """Rows from a featurizer whose device tables are built under the given environment."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        _lib.drop_device_basis(basis)
        fz = process.BasisFeaturizer(basis)
        _, db = fz._dev()
        modes = db.featurizer_modes
        x_e, x_f, _ = fz.featurize_frames(frames)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _lib.drop_device_basis(basis)
    return x_e, x_f, modes


def _asymmetric_window_basis():
    """Mo/W basis whose mixed trios have 2 x 5 x 16 kept bins (record stride 60: fewer than 21 records per pass)."""
    from uf3_amd.data import composition
    from uf3_amd.representation import bspline
    cs = composition.ChemicalSystem(['Mo', 'W'], 3)
    pairs, trios = cs.interactions_map[2], cs.interactions_map[3]
    res = {p: 12 for p in pairs}
    for t in trios:
        res[t] = [5, 8, 19] if t[1] != t[2] else [6, 6, 12]
    return bspline.BSplineBasis(
        cs, r_min_map={**{p: 0.5 for p in pairs}, **{t: [1.5, 1.5, 1.5] for t in trios}},
        r_max_map={**{p: 5.0 for p in pairs}, **{t: [3.6, 3.6, 7.2] for t in trios}},
        resolution_map=res, leading_trim={2: 0, 3: 3}, trailing_trim={2: 3, 3: 3})


def _resolution_basis(res3, lead3=3, elements=('Mo', 'W')):
    from uf3_amd.data import composition
    from uf3_amd.representation import bspline
    cs = composition.ChemicalSystem(list(elements), 3)
    pairs, trios = cs.interactions_map[2], cs.interactions_map[3]
    return bspline.BSplineBasis(
        cs, r_min_map={**{p: 0.001 for p in pairs}, **{t: [1.5, 1.5, 1.5] for t in trios}},
        r_max_map={**{p: 5.5 for p in pairs}, **{t: [3.5, 3.5, 7.0] for t in trios}},
        resolution_map={**{p: 15 for p in pairs}, **{t: list(res3) for t in trios}},
        leading_trim={2: 0, 3: lead3}, trailing_trim={2: 3, 3: 3})


@pytest.mark.parametrize("which,bit", [("notebook_binary", 7), ("asymmetric_window", 9), ("h2o_golden", None), ("w16_sym1", None),
                                       ("w16_sym3", None), ("default_resolution", 6), ("four_by_four", 8), ("six_by_six", 9),
                                       ("five_by_five_unary", 9)])
def test_matrix_core_and_generic_trio_kernels_agree(which, bit):
    """The fp64 MFMA specialisations (mode bits 6 / 7 / 8 / 9: windows of (row tiles, column tiles) = (1,1), (1,2),
    (1,<=4), (<=2,<=6) with rows (component, l) and columns (m, n)) against the output-stationary kernels on the same
    inputs, and both against the oracle."""
    if which == "notebook_binary":
        frames = [synthetic.lattice_frame("bcc", (3, 3, 4), 3.165, [42, 74], 11 + k) for k in range(2)]
        basis = synthetic.notebook_basis(['Mo', 'W'])
    elif which == "asymmetric_window":
        frames = [synthetic.lattice_frame("bcc", (3, 4, 3), 3.2, [42, 74], 5, rattle=0.12)]
        basis = _asymmetric_window_basis()
    elif which == "default_resolution":        # 2 x 2 x 7 kept bins (the reference's default 3-body resolution [5, 5, 10])
        frames = [synthetic.lattice_frame("bcc", (3, 3, 4), 3.165, [42, 74], 30, rattle=0.1)]
        basis = _resolution_basis([5, 5, 10])
    elif which == "four_by_four":              # 4 x 4 x 11 kept bins: 16 rows, 44 columns
        frames = [synthetic.lattice_frame("bcc", (3, 3, 4), 3.165, [42, 74], 31, rattle=0.1)]
        basis = _resolution_basis([7, 7, 14])
    elif which == "six_by_six":                # 6 x 6 x 12 kept bins (no leading trim): 24 rows, 72 columns
        frames = [synthetic.lattice_frame("bcc", (3, 4, 3), 3.165, [42, 74], 32, rattle=0.1)]
        basis = synthetic.notebook_basis(['Mo', 'W'], lead3=0)
    elif which == "five_by_five_unary":        # 5 x 5 x 13 kept bins, symmetry 2 fold: 20 rows, 65 columns
        frames = [synthetic.lattice_frame("bcc", (3, 3, 3), 3.165, [74], 33, rattle=0.1)]
        basis = _resolution_basis([8, 8, 16], elements=('W',))
    else:
        d, meta, atoms = load_case({"h2o_golden": "case_h2o", "w16_sym1": "case_w16_sym1", "w16_sym3": "case_w16_sym3"}[which])
        if which != "h2o_golden":        # the captured bases keep too many bins: trim the leading ones as well
            meta = json.loads(json.dumps(meta))
            meta["basis_kwargs"]["leading_trim"] = {"2": 0, "3": 3}
            meta["basis_kwargs"]["trailing_trim"] = {"2": 3, "3": 3}
        frames, basis = [atoms], basis_from_meta(meta)
    xe_m, xf_m, modes_m = _fresh_rows(basis, frames, UF3_NO_FEAT3="1")
    xe_g, xf_g, modes_g = _fresh_rows(basis, frames, UF3_NO_FEAT3="1", UF3_NO_MFMA_FEAT="1")
    assert modes_m & (0x3c0 if bit is None else 1 << bit), f"expected featurizer mode bit {bit} for this basis, got {modes_m:#x}"
    assert not (modes_g & 0x3c0) and (modes_g & 0x3e) and not (modes_m & 0x1000) and not (modes_g & 0x1000)
    assert rel_err(xe_m, xe_g) < 1e-12 and rel_err(xf_m, xf_g) < 1e-12
    # round 4: the bond-factorised launch (k_featurize3, mode bit 12) where the basis qualifies -- a third, independent traversal
    xe_b, xf_b, modes_b = _fresh_rows(basis, frames)
    assert bool(modes_b & 0x1000) == (which in ("notebook_binary", "default_resolution", "h2o_golden", "four_by_four", "six_by_six", "five_by_five_unary")), hex(modes_b)
    assert rel_err(xe_b, xe_m) < 1e-12 and rel_err(xf_b, xf_m) < 1e-12
    assert worst_elementwise(xf_b, xf_m, rtol=1e-9, floor=1e-12) <= 1.0
    ob = O.OracleBasis(basis)
    off = 0
    for k, fr in enumerate(frames):
        ref = O.featurize(ob, fr)
        assert rel_err(xe_m[k], ref["xe"]) < TOL
        assert rel_err(xf_m[off:off + len(fr)].reshape(ref["xf"].shape), ref["xf"]) < TOL
        off += len(fr)
    # energy-only and forces-only launches of the same specialisation
    fz = process.BasisFeaturizer(basis)
    assert rel_err(fz.featurize_frames(frames, forces=False)[0], xe_m) < 1e-12
    assert rel_err(fz.featurize_frames(frames, energy=False)[1], xf_m) < 1e-12


@pytest.mark.parametrize("lead3", [3, 0])
def test_trios_with_different_settings_in_one_basis(lead3):
    """Every trio its own cut-offs / resolution: several window LAYOUTS in one basis (the per-lane layout constants are
    recomputed when a block of another layout comes along), grouped and two-tile windows side by side in the mode-7 launch
    (lead3 = 3), banded and dense wide windows side by side in the mode-9 launch (lead3 = 0)."""
    from uf3_amd.data import composition
    from uf3_amd.representation import bspline
    cs = composition.ChemicalSystem(['Mo', 'W'], 3)
    pairs, trios = cs.interactions_map[2], cs.interactions_map[3]
    res = [[6, 6, 12], [6, 6, 13], [6, 6, 10], [6, 6, 12], [5, 5, 12], [6, 6, 11]]
    rmax = [3.5, 3.5, 3.7, 3.9, 3.5, 3.6]
    basis = bspline.BSplineBasis(
        cs,
        r_min_map={**{p: 0.001 for p in pairs}, **{t: [1.5, 1.5, 1.5] for t in trios}},
        r_max_map={**{p: 5.5 for p in pairs}, **{t: [r, r, 2 * r] for t, r in zip(trios, rmax)}},
        resolution_map={**{p: 15 for p in pairs}, **{t: r for t, r in zip(trios, res)}},
        leading_trim={2: 0, 3: lead3}, trailing_trim={2: 3, 3: 3})
    frames = [synthetic.lattice_frame("bcc", (4, 4, 5), 3.165, [42, 74], 31), synthetic.lattice_frame("bcc", (3, 5, 4), 3.0, [42, 74], 32)]
    fz = process.BasisFeaturizer(basis)
    modes = fz._dev()[1].featurizer_modes
    assert modes & ((1 << 7) if lead3 else (1 << 9)), hex(modes)
    x_e, x_f, off = fz.featurize_frames(frames)
    ob = O.OracleBasis(basis)
    for k, atoms in enumerate(frames):
        ref = O.featurize(ob, atoms)
        assert rel_err(x_e[k], ref["xe"]) < TOL and worst_elementwise(x_e[k], ref["xe"]) <= 1.0
        assert rel_err(x_f[off[k]:off[k + 1]], ref["xf"]) < TOL and worst_elementwise(x_f[off[k]:off[k + 1]], ref["xf"]) <= 1.0
    xe2 = fz.featurize_frames(frames, forces=False)[0]                 # (energy-only launches: the dense loops)
    assert rel_err(xe2, x_e) < 1e-12


def test_windows_too_wide_for_the_tiles_stay_on_generic_kernels():
    """More than 96 (m, n) columns (or more than 32 (component, l) rows): output-stationary kernels; default trims:
    matrix cores."""
    fz = process.BasisFeaturizer(_resolution_basis([6, 6, 20], lead3=0, elements=('W',)))     # 6 x 6 x 20 kept bins
    modes = fz._dev()[1].featurizer_modes
    assert not (modes & 0x3c0) and (modes & 0x3e)
    atoms = synthetic.lattice_frame("bcc", (3, 3, 3), 3.165, [74], 35, rattle=0.1)
    _check_against_oracle(fz.bspline_config, [atoms])
    # the reference's default trims (3 leading, 3 trailing) give 3 x 3 x 9: matrix cores, one row tile x two column tiles
    assert process.BasisFeaturizer(synthetic.config_c3()[1])._dev()[1].featurizer_modes & (1 << 7)


def test_grouped_windows_with_non_uniform_knot_sequences():
    """3 x 3 x 9 windows (mode bit 7: window rows, polynomial pieces per knot interval) on knot sequences the user supplies,
    spaced unevenly: the interval guess is wrong for most distances and the row is fetched again; pieces next to clamped
    ends and next to short intervals; (l = m, n) and three separate sequences."""
    from uf3_amd.data import composition
    from uf3_amd.representation import bspline

    def clamped(lo, hi, n_int, power):
        inner = lo + (hi - lo) * np.linspace(0.0, 1.0, n_int + 1) ** power
        return np.concatenate([[lo] * 3, inner, [hi] * 3])

    cs = composition.ChemicalSystem(['Mo', 'W'], 3)
    pairs, trios = cs.interactions_map[2], cs.interactions_map[3]
    knots = {p: clamped(0.001, 5.5, 15, 1.3) for p in pairs}
    for q, t in enumerate(trios):
        if q % 2:
            knots[t] = [clamped(1.5, 3.5, 6, 1.6), clamped(1.5, 3.5, 6, 0.7), clamped(1.5, 7.0, 12, 1.4)]
        else:
            knots[t] = [clamped(1.5, 3.5, 6, 1.5), clamped(1.5, 7.0, 12, 0.75)]
    basis = bspline.BSplineBasis(cs, knots_map=knots, leading_trim={2: 0, 3: 3}, trailing_trim={2: 3, 3: 3})
    fz = process.BasisFeaturizer(basis)
    assert fz._dev()[1].featurizer_modes & (1 << 7)
    frames = [synthetic.lattice_frame("bcc", (3, 3, 4), 3.165, [42, 74], 77, rattle=0.15),
              synthetic.lattice_frame("fcc", (2, 3, 2), 4.0, [42, 74], 78, rattle=0.2)]
    x_e, x_f = _check_against_oracle(basis, frames)
    assert np.abs(x_f[:, :, basis.n_feats - 40:]).max() > 0


def test_bond_factorised_launch_with_non_uniform_knot_sequences():
    """k_featurize3 (mode bit 12) on one unevenly spaced set of knot sequences shared by all trios -- the interval guess of its
    window rows is wrong for most distances -- at the default trims and without the leading trim (6 x 6 x 12 windows, three
    rounds), against the oracle entry by entry and against the matrix-core launches."""
    from uf3_amd.data import composition
    from uf3_amd.representation import bspline

    def clamped(lo, hi, n_int, power):
        inner = lo + (hi - lo) * np.linspace(0.0, 1.0, n_int + 1) ** power
        return np.concatenate([[lo] * 3, inner, [hi] * 3])

    cs = composition.ChemicalSystem(['Mo', 'W'], 3)
    pairs, trios = cs.interactions_map[2], cs.interactions_map[3]
    knots = {p: clamped(0.001, 5.5, 15, 1.3) for p in pairs}
    for t in trios:
        knots[t] = [clamped(1.5, 3.5, 6, 1.5), clamped(1.5, 7.0, 12, 0.75)]
    frames = [synthetic.lattice_frame("bcc", (3, 3, 4), 3.165, [42, 74], 87, rattle=0.15),
              synthetic.lattice_frame("fcc", (2, 3, 2), 4.0, [42, 74], 88, rattle=0.2)]
    for lead in (3, 0):
        basis = bspline.BSplineBasis(cs, knots_map=knots, leading_trim={2: 0, 3: lead}, trailing_trim={2: 3, 3: 3})
        assert process.BasisFeaturizer(basis)._dev()[1].featurizer_modes & (1 << 12)
        x_e, x_f = _check_against_oracle(basis, frames)
        xe_m, xf_m, modes_m = _fresh_rows(basis, frames, UF3_NO_FEAT3="1")
        assert not (modes_m & (1 << 12)) and worst_elementwise(x_f, xf_m) <= 1.0 and rel_err(x_e, xe_m) < 1e-12


def _lj_like_model():
    """2-body W model whose coefficients are a least-squares B-spline fit of a Lennard-Jones curve."""
    from uf3_amd.data import composition
    from uf3_amd.representation import bspline
    cs = composition.ChemicalSystem(['W'], 2)
    basis = bspline.BSplineBasis(cs, r_min_map={('W', 'W'): 1.8}, r_max_map={('W', 'W'): 5.5},
                                 resolution_map={('W', 'W'): 24}, leading_trim=0, trailing_trim=3)
    knots = basis.knots_map[('W', 'W')]
    r = np.linspace(1.85, 5.49, 2000)
    first, vals, _ = bspline.basis_values(knots, r)
    nb = len(knots) - 4
    design = np.zeros((len(r), nb))
    for q in range(4):
        design[np.arange(len(r)), first + q] += vals[:, q]
    design = design[:, :nb - 3]                                      # trailing-trimmed functions stay 0
    sigma, eps = 2.45, 0.4
    phi = 4 * eps * ((sigma / r) ** 12 - (sigma / r) ** 6) * (1 - (r / 5.5) ** 2) ** 2   # smooth cut-off
    c = np.zeros(nb)
    c[:nb - 3] = np.linalg.lstsq(design, phi, rcond=None)[0]
    model = ls.WeightedLinearModel(basis)
    model.coefficients = np.concatenate([[0.0], 0.5 * c])           # each bond is visited from both ends
    return model


def test_relax_fmax_positions_and_cell():
    """relax_fmax (calculator.py:406-436): forces and stress go to zero, the energy goes down."""
    calc = calculator.UFCalculator(_lj_like_model())
    atoms = synthetic.lattice_frame("fcc", (2, 2, 2), 3.9, [74], 3, rattle=0.08, strain=0.02)
    e0 = calc.get_potential_energy(atoms)
    f0 = np.abs(calc.get_forces(atoms)).max()
    relaxed = calc.relax_fmax(atoms, fmax=0.01, relax_cell=False, timeout=120.0)
    assert np.abs(calc.get_forces(relaxed)).max() < 0.01 < f0
    assert calc.get_potential_energy(relaxed) < e0
    assert np.allclose(np.array(relaxed.get_cell()), np.array(atoms.get_cell()))
    both = calc.relax_fmax(atoms, fmax=0.01, relax_cell=True, timeout=120.0)
    assert np.abs(calc.get_forces(both)).max() < 0.01
    vol = abs(np.linalg.det(np.array(both.get_cell(), dtype=float).reshape(3, 3)))
    assert np.abs(calc._get_stress(both)).max() * vol / len(both) < 0.01
    assert calc.get_potential_energy(both) <= calc.get_potential_energy(relaxed) + 1e-9


def test_fused_and_separate_neighbour_list_builds_agree():
    """The 3-body lists built inside the pair launch (from its LDS candidates) against k_build_n3's."""
    cases = [(synthetic.notebook_basis(['Mo', 'W']), [synthetic.lattice_frame("bcc", (3, 4, 3), 3.165, [42, 74], 21)])]
    cs = synthetic.composition.ChemicalSystem(['W'], 3)
    dense = synthetic.bspline.BSplineBasis(
        cs, r_min_map={('W', 'W'): 0.3, ('W', 'W', 'W'): [0.5, 0.5, 0.5]},
        r_max_map={('W', 'W'): 4.0, ('W', 'W', 'W'): [5.0, 5.0, 10.0]},          # 3-body range beyond the pair range
        resolution_map={('W', 'W'): 8, ('W', 'W', 'W'): [5, 5, 10]})
    cases.append((dense, [synthetic.lattice_frame("bcc", (4, 4, 4), 2.6, [74], seed=3, rattle=0.1)]))
    for basis, frames in cases:
        fz = process.BasisFeaturizer(basis)
        x_e, x_f, _ = fz.featurize_frames(frames)
        os.environ["UF3_SEPARATE_N3"] = "1"
        try:
            y_e, y_f, _ = fz.featurize_frames(frames)
        finally:
            del os.environ["UF3_SEPARATE_N3"]
        assert rel_err(x_e, y_e) < 1e-13 and rel_err(x_f, y_f) < 1e-13
        ref = O.featurize(O.OracleBasis(basis), frames[0])
        assert rel_err(x_e[0], ref["xe"]) < 1e-8 and rel_err(x_f, ref["xf"]) < 1e-8


def test_energy_row_longer_than_lds():
    """F = 15925 columns: the block-shared energy row (8 F bytes) no longer fits LDS; contributions go to HBM."""
    basis = _resolution_basis([9, 9, 17], lead3=0, elements=('Al', 'Cu', 'Zr'))
    assert basis.n_feats * 8 > 48 * 1024
    rng = np.random.default_rng(12)
    atoms = synthetic.lattice_frame("bcc", (3, 3, 3), 3.1, [13, 29, 40], 41, rattle=0.1)
    _check_against_oracle(basis, [atoms])
    e_only = process.BasisFeaturizer(basis).featurize_frames([atoms], forces=False)[0]
    assert rel_err(e_only[0], O.featurize(O.OracleBasis(basis), atoms, forces=False)["xe"]) < TOL


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_bases_and_cells_against_oracle(seed):
    """Random species sets, resolutions, trims, cut-offs, strained cells and periodicity patterns (a short version
    of tools/experiments/random_parity.py): feature rows and evaluator against the oracle."""
    from uf3_amd.data import composition
    from uf3_amd.representation import bspline
    rng = np.random.default_rng(100 + seed)
    zmap = {'Al': 13, 'Cu': 29, 'Mo': 42, 'W': 74, 'Zr': 40}
    for _ in range(3):
        els = sorted(rng.choice(list(zmap), int(rng.integers(1, 4)), replace=False).tolist())
        cs = composition.ChemicalSystem(els, 3)
        pairs, trios = cs.interactions_map[2], cs.interactions_map[3]
        r3 = float(rng.uniform(3.0, 4.2))
        res_l, res_n = int(rng.integers(4, 10)), int(rng.integers(8, 20))
        basis = bspline.BSplineBasis(
            cs, r_min_map={**{p: float(rng.uniform(0.2, 1.0)) for p in pairs}, **{t: [float(rng.uniform(0.8, 1.6))] * 3 for t in trios}},
            r_max_map={**{p: float(rng.uniform(4.0, 6.0)) for p in pairs}, **{t: [r3, r3, 2 * r3] for t in trios}},
            resolution_map={**{p: int(rng.integers(6, 18)) for p in pairs}, **{t: [res_l, res_l, res_n] for t in trios}},
            leading_trim={2: 0, 3: int(rng.choice([0, 3]))}, trailing_trim={2: 3, 3: int(rng.choice([3, 2]))})
        reps = tuple(int(x) for x in rng.integers(3, 5, 3))
        a = float(rng.uniform(2.9, 3.4))
        grid = np.array([[i, j, k] for i in range(reps[0]) for j in range(reps[1]) for k in range(reps[2])], float)
        frac = (grid[:, None, :] + np.array([[0, 0, 0], [.5, .5, .5]])[None]).reshape(-1, 3)
        cell = np.diag(np.array(reps, float) * a) @ (np.eye(3) + rng.normal(0, 0.02, (3, 3)))
        pos = (frac / np.array(reps)) @ cell + rng.normal(0, 0.1, (len(frac), 3))
        pbc = [True, True, True] if rng.random() < 0.6 else [bool(b) for b in rng.integers(0, 2, 3)]
        atoms = Atoms(numbers=rng.choice([zmap[e] for e in els], len(pos)), positions=pos, cell=cell, pbc=pbc)
        _check_against_oracle(basis, [atoms])
        coeff = rng.normal(0, 0.05, basis.n_feats)
        coeff[basis.col_idx] = 0.0
        model = ls.WeightedLinearModel(basis)
        model.coefficients = coeff
        e, f, _ = calculator.UFCalculator(model).evaluate_frames([atoms])
        e_ref, f_ref = O.evaluate(O.OracleBasis(basis), atoms, coeff)
        assert abs(e[0] - e_ref) <= TOL * max(1.0, abs(e_ref)) and rel_err(f, f_ref) < TOL


def _nccl_fit_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from uf3_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["UF3_DEVICE"] = str(rank)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    basis = synthetic.notebook_basis(['W'])
    frames = [synthetic.lattice_frame("bcc", (3, 3, 3 + (k % 3)), 3.165, [74], seed=150 + k) for k in range(9)]
    rng = np.random.default_rng(11)
    energies = rng.normal(size=len(frames))
    forces = [rng.normal(size=(len(f), 3)) for f in frames]
    fz = process.BasisFeaturizer(basis, device=rank)
    reg = basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0)
    model = ls.WeightedLinearModel(basis, regularizer=reg)
    parallel.sharded_fit(model, fz, frames, energies, forces, weight=0.4)        # RCCL all_reduce of the device buffer
    np.save(os.path.join(out_dir, f"nccl_{rank}.npy"), model.coefficients)
    dist.destroy_process_group()


def test_two_gpu_fit_over_rccl_matches_one_gpu(tmp_path):
    """The N > 1 path on hardware: frames sharded over 2 GPUs, packed pieces summed by RCCL on the device buffers ==
    the same fit on one GPU.  Skips where the box has a single GPU (the driver's 8-GPU node runs it)."""
    import socket
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    from uf3_amd import pipeline
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_nccl_fit_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    basis = synthetic.notebook_basis(['W'])
    frames = [synthetic.lattice_frame("bcc", (3, 3, 3 + (k % 3)), 3.165, [74], seed=150 + k) for k in range(9)]
    rng = np.random.default_rng(11)
    energies = rng.normal(size=len(frames))
    forces = [rng.normal(size=(len(f), 3)) for f in frames]
    reg = basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0)
    model = ls.WeightedLinearModel(basis, regularizer=reg)
    pipeline.fit_frames(model, process.BasisFeaturizer(basis), frames, energies, forces, weight=0.4, reduce=False)
    c0, c1 = np.load(tmp_path / "nccl_0.npy"), np.load(tmp_path / "nccl_1.npy")
    assert np.array_equal(c0, c1)
    assert np.allclose(c0, model.coefficients, rtol=1e-8, atol=1e-10)


def _nccl_eval_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from uf3_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["UF3_DEVICE"] = str(rank)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    basis = synthetic.notebook_basis(['Mo', 'W'])
    atoms = synthetic.lattice_frame("bcc", (6, 7, 8), 3.165, [42, 74], seed=77)
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(5).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    calc = calculator.UFCalculator(model, device=rank)
    e, f, v = parallel.sharded_evaluate(calc, atoms, forces=True, virial=True, device=rank)   # RCCL all_reduce of [E | dE/deps | F]
    # the device-resident MD-loop form of the same decomposition (what bench.py --mode eval times): two steps on persistent lists
    ev = parallel.ShardedEvaluator(calc, atoms, device=rank, md_skin=0.5)
    e2, f2, v2 = ev.step().result()
    ev.positions.add_(0.01)                                  # (a rigid shift: same energy and forces, through the MD route's reuse)
    e3, f3, v3 = ev.step().result()
    ev.close()
    assert abs(e2 - e) <= 1e-12 * abs(e) and np.abs(f2 - f).max() <= 1e-12 * np.abs(f).max() and np.allclose(v2, v, rtol=1e-10, atol=1e-10)
    assert abs(e3 - e) <= 1e-10 * abs(e) and np.abs(f3 - f).max() <= 1e-9 * np.abs(f).max()
    np.savez(os.path.join(out_dir, f"nccl_eval_{rank}.npz"), e=e, f=f, v=v)
    dist.destroy_process_group()


def test_two_gpu_decomposed_evaluation_over_rccl_matches_the_oracle(tmp_path):
    """N4 on hardware: one frame, a block of centres per GPU (uf3_eval_centres), one RCCL all_reduce == the oracle's
    evaluator and the one-GPU whole-frame route.  Skips where the box has a single GPU."""
    import socket
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_nccl_eval_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    basis = synthetic.notebook_basis(['Mo', 'W'])
    atoms = synthetic.lattice_frame("bcc", (6, 7, 8), 3.165, [42, 74], seed=77)
    coeff = np.random.default_rng(5).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    e_ref, f_ref = O.evaluate(O.OracleBasis(basis), atoms, coeff)
    model = ls.WeightedLinearModel(basis)
    model.coefficients = coeff
    e1, f1, _, v1 = calculator.UFCalculator(model).evaluate_frames([atoms], virial=True)
    for rank in range(2):
        d = np.load(tmp_path / f"nccl_eval_{rank}.npz")
        assert abs(float(d["e"]) - e_ref) <= TOL * max(1.0, abs(e_ref))
        assert worst_elementwise(d["f"], f_ref) <= 1.0
        assert np.allclose(d["v"], v1[0], rtol=1e-10, atol=1e-10) and rel_err(d["f"], f1) < 1e-12


def _native_fit_worker(rank, world, out_dir):
    """no torch.distributed: the communicator's id travels through a file, the sum through uf3_allreduce_sum_f64"""
    import torch
    from uf3_amd import parallel, pipeline
    os.environ["UF3_DEVICE"] = str(rank)
    torch.cuda.set_device(rank)
    basis = synthetic.notebook_basis(['W'])
    frames = [synthetic.lattice_frame("bcc", (3, 3, 3 + (k % 3)), 3.165, [74], seed=150 + k) for k in range(9)]
    rng = np.random.default_rng(11)
    energies = rng.normal(size=len(frames))
    forces = [rng.normal(size=(len(f), 3)) for f in frames]
    reg = basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0)
    model = ls.WeightedLinearModel(basis, regularizer=reg)
    fz = process.BasisFeaturizer(basis, device=rank)
    ctx, _ = fz._dev()
    parallel.native_comm(ctx, rank, world, id_path=os.path.join(out_dir, "comm_id"))
    lo, hi = parallel.shard_range(len(frames), rank, world)
    pipeline.fit_frames(model, fz, frames[lo:hi], energies[lo:hi], forces[lo:hi], weight=0.4, with_forces=True)
    np.save(os.path.join(out_dir, f"native_{rank}.npy"), model.coefficients)
    ctx.comm_destroy()


def test_two_gpu_fit_through_the_librarys_own_communicator(tmp_path):
    """The same two-GPU fit with RCCL bound behind the C ABI (uf3_comm_init / uf3_allreduce_sum_f64) and NO torch.distributed:
    the id goes through a file.  Skips where the box has a single GPU."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    from uf3_amd import pipeline
    mp.spawn(_native_fit_worker, args=(2, str(tmp_path)), nprocs=2, join=True)
    basis = synthetic.notebook_basis(['W'])
    frames = [synthetic.lattice_frame("bcc", (3, 3, 3 + (k % 3)), 3.165, [74], seed=150 + k) for k in range(9)]
    rng = np.random.default_rng(11)
    energies = rng.normal(size=len(frames))
    forces = [rng.normal(size=(len(f), 3)) for f in frames]
    reg = basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0)
    model = ls.WeightedLinearModel(basis, regularizer=reg)
    pipeline.fit_frames(model, process.BasisFeaturizer(basis), frames, energies, forces, weight=0.4, reduce=False)
    c0, c1 = np.load(tmp_path / "native_0.npy"), np.load(tmp_path / "native_1.npy")
    assert np.array_equal(c0, c1)
    assert np.allclose(c0, model.coefficients, rtol=1e-8, atol=1e-10)


def test_rccl_behind_the_c_abi_single_rank():
    """uf3_comm_unique_id / uf3_comm_init / uf3_allreduce_sum_f64 (librccl opened by the library itself): a communicator of one
    rank comes up on this GPU and the in-place sum of a device buffer runs on the context's stream; the fit pipeline takes that
    route when the context has a communicator."""
    import torch
    from uf3_amd import parallel, pipeline
    ctx = _lib.get_context(None)
    assert ctx.comm_info() == (0, -1)
    with pytest.raises(_lib.UF3Error):
        ctx.allreduce_sum(0, 0)                                   # no communicator yet
    parallel.native_comm(ctx, rank=0, world_size=1)
    try:
        assert ctx.comm_info() == (1, 0)
        dev = torch.device("cuda", ctx.device)
        buf = torch.arange(100003, dtype=torch.float64, device=dev) * 0.25
        ref = buf.clone()
        prev = ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        try:
            ctx.allreduce_sum(buf.data_ptr(), buf.numel())
        finally:
            ctx.restore_stream(prev)
        torch.cuda.synchronize(dev)
        assert torch.equal(buf, ref)
        out = parallel.allreduce_packed(buf, force=True, ctx=ctx)
        torch.cuda.synchronize(dev)
        assert out is buf and torch.equal(buf, ref)
        # the fit pipeline end to end with the context's communicator in the data path
        basis = synthetic.notebook_basis(['W'])
        frames = [synthetic.lattice_frame("bcc", (3, 3, 3), 3.165, [74], seed=60 + k) for k in range(6)]
        rng = np.random.default_rng(2)
        e = rng.normal(-480, 1.0, len(frames))
        f = [rng.normal(0, 0.3, (len(a), 3)) for a in frames]
        reg = basis.get_regularization_matrix(ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0)
        m1, m2 = ls.WeightedLinearModel(basis, regularizer=reg), ls.WeightedLinearModel(basis, regularizer=reg)
        fz = process.BasisFeaturizer(basis)
        os.environ["UF3_FORCE_COLLECTIVE"] = "1"
        try:
            pipeline.fit_frames(m1, fz, frames, e, f, weight=0.4)
        finally:
            del os.environ["UF3_FORCE_COLLECTIVE"]
        ctx.comm_destroy()
        pipeline.fit_frames(m2, fz, frames, e, f, weight=0.4)
        assert np.allclose(m1.coefficients, m2.coefficients, rtol=1e-5, atol=1e-6)      # (the Gram kernels add with atomics: not to the bit)
    finally:
        ctx.comm_destroy()
    assert ctx.comm_info() == (0, -1)


@pytest.mark.parametrize("mode", ["featurize", "fit", "eval"])
def test_bench_under_the_launcher_runs_rccl_on_device_buffers(mode):
    """every GPU test run initialises RCCL.  bench.py exactly as the driver starts it for N > 1
    (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) with ONE rank and UF3_FORCE_COLLECTIVE=1: the
    communicator comes up, the barrier / MAX-reduce run, and the data-path all_reduce (packed fit pieces; the decomposed
    evaluator's [forces | energy | strain derivative]) is issued on the device buffer itself."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--mode", mode, "--no-cpu-baseline", "--no-extra", "--no-traffic"]
    argv += ["--atoms", "2000"] + (["--frames-per-step", "4"] if mode != "eval" else [])
    env = dict(os.environ, UF3_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    values = {}
    for native in ((False, True) if mode != "featurize" else (False,)):       # (the data-path sum through the library's own communicator)
        if native:
            env["UF3_NATIVE_RCCL"] = "1"
        res = subprocess.run(bench.rank_command(1, argv, port + (1 if native else 0)), env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, res.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 1 and out["config"]["rccl_world_size"] == 1
        assert out["config"].get("forced_collective") is True
        assert out["config"]["collective"].startswith("uf3_allreduce_sum_f64" if native else "torch.distributed")
        assert np.isfinite(out["value"]) and out["value"] > 0
        values[native] = out["value"]


def test_random_call_sequences_on_one_context_do_not_depend_on_its_history():
    """A short run of tools/experiments/state_fuzz.py (its header: one process, the shared context, three bases x five kinds of
    frames, a seeded random sequence of featurize / evaluate / fit calls, every result against the oracle): capacities that only
    grow, persistent neighbour lists, staging blocks and cached coefficients must not leak from one call into the next."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "experiments", "state_fuzz.py"), "120", "9"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "mismatches 0" in res.stdout


def test_small_md_steps_staged_through_the_bar_equal_the_fetched_ones(monkeypatch):
    """Small MD steps put positions | species straight into device memory through the BAR when the status words are known to be
    clean (no k_md_fetch launch): the same walk on a context with the route switched off (UF3_NO_BAR_STAGE, read when the context
    is made) gives the same bits, a step that outruns the lists in between included; against the oracle at the end."""
    basis = synthetic.notebook_basis(['Mo', 'W'])
    model, coeff = _random_model(basis, 4)
    start = synthetic.lattice_frame("bcc", (4, 4, 4), 3.165, [42, 74], seed=8)
    rng = np.random.default_rng(2)
    path, pos = [], start.get_positions()
    for step in range(40):
        pos = pos + rng.uniform(-0.02, 0.02, pos.shape)
        if step == 17:
            pos[5] += [0.3, 0.1, -0.2]
        path.append(pos.copy())
    runs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("UF3_NO_BAR_STAGE", "1")
        monkeypatch.setattr(_lib, "_contexts", {})
        calc = calculator.UFCalculator(model, md_skin=0.4)
        out = []
        for p in path:
            atoms = Atoms(numbers=start.get_atomic_numbers(), positions=p, cell=start.get_cell(), pbc=True)
            e, f, _ = calc.evaluate_frames([atoms])
            out.append((e.copy(), f.copy()))
        st = _lib.get_context(None).md_stats()
        assert st["steps"] >= len(path) and st["redone"] >= 1
        runs.append(out)
        _lib.drop_device_basis(basis)
    for (e0, f0), (e1, f1) in zip(*runs):
        assert np.array_equal(e0, e1) and np.array_equal(f0, f1)
    atoms = Atoms(numbers=start.get_atomic_numbers(), positions=path[-1], cell=start.get_cell(), pbc=True)
    e_ref, f_ref = O.evaluate(O.OracleBasis(basis), atoms, coeff)
    assert rel_err(runs[0][-1][0][0], e_ref) < TOL and rel_err(runs[0][-1][1], f_ref) < TOL


def test_md_rebuild_that_nobody_waits_for_overflows_into_a_repeat():
    """A rebuild of the MD lists at a capacity that has held before does not wait for its own report (md_build, may_defer): when
    the same atoms come back in a cell compressed by 25 % the superset lists no longer fit, k_sup_reverse raises the step's
    "lists outrun" word, the step is discarded and the repeat builds with the host looking -- results against the oracle, and
    the walk goes on (calculator.py:124-153: every call from scratch)."""
    basis = synthetic.notebook_basis(['Mo', 'W'])
    model, coeff = _random_model(basis, 6)
    calc = calculator.UFCalculator(model, md_skin=0.5)
    start = synthetic.lattice_frame("bcc", (4, 4, 4), 3.3, [42, 74], seed=12)
    ob = O.OracleBasis(basis)
    ctx = _lib.get_context(None)
    rng = np.random.default_rng(3)
    pos, cell = start.get_positions(), np.array(start.get_cell(), float)
    for step in range(12):
        pos = pos + rng.uniform(-0.02, 0.02, pos.shape)
        calc.evaluate_frames([Atoms(numbers=start.get_atomic_numbers(), positions=pos, cell=cell, pbc=True)])
    before = ctx.md_stats()
    pos, cell = pos * 0.75, cell * 0.75
    for step in range(6):
        pos = pos + rng.uniform(-0.02, 0.02, pos.shape)
        atoms = Atoms(numbers=start.get_atomic_numbers(), positions=pos, cell=cell, pbc=True)
        e, f, _ = calc.evaluate_frames([atoms])
        e_ref, f_ref = O.evaluate(ob, atoms, coeff)
        assert rel_err(e[0], e_ref) < TOL and rel_err(f, f_ref) < TOL, step
    after = ctx.md_stats()
    assert after["redone"] > before["redone"] and after["builds"] >= before["builds"] + 2


# ---------------------------------------------------------------------------------------------------------------------------
# Neighbour indices at scale.  north_star: "bit-exact neighbor indices".  The captures above are
# <= 128 atoms; here the BASELINE-size cells -- multi-bin walks, capacity regrowth, every size class of the cell-list stage --
# against the oracle's explicit-supercell search (reference: distances.py:48-69, angles.py:289-346), bit for bit: the pair
# lists per interaction and the 3-body pairs from the query entry (uf3_neighbors_debug), AND the 3-body lists the featurizer's
# own launches build and consume (uf3_n3_lists_debug).  Row parity cannot see a pair mis-classified AT a cut-off (it contributes
# ~0 to every B-spline column); the last test puts atoms within a few ulp of each cut-off, on both sides.
def _assert_indices_equal_oracle(basis, atoms):
    ref = O.featurize(O.OracleBasis(basis), atoms, energy=False, forces=False, indices=True)
    fz = process.BasisFeaturizer(basis)
    pairs, n3 = fz.neighbor_indices(atoms)
    assert set(pairs) == set(ref["pairs"])
    for pair in basis.interactions_map[2]:
        assert pairs[pair].shape == ref["pairs"][pair].shape, pair
        assert np.array_equal(pairs[pair], ref["pairs"][pair]), pair
    assert n3.shape == ref["n3"].shape and np.array_equal(n3, ref["n3"])
    own = fz.product_n3_indices(atoms)                      # the lists MODE 0 built for its own 3-body launches
    assert own.shape == ref["n3"].shape and np.array_equal(own, ref["n3"])
    return ref


@pytest.mark.parametrize("config", ["c2", "c3", "c4", "c5"])
def test_neighbour_indices_bit_exact_at_baseline_sizes(config):
    atoms, basis = {"c2": synthetic.config_c2, "c3": synthetic.config_c3, "c4": synthetic.config_c4, "c5": synthetic.config_c5}[config]()
    ref = _assert_indices_equal_oracle(basis, atoms)
    assert len(ref["n3"]) > 10 * len(atoms)                 # (the lists are not trivially empty)


def test_neighbour_indices_bit_exact_within_an_ulp_of_every_cutoff():
    basis = synthetic.notebook_basis(['Mo', 'W'])
    r_max2 = 5.5
    r_min3, r_max3 = 1.5, 3.5
    atoms = synthetic.cutoff_probe_frame([r_max2, r_min3, r_max3], elements=(42, 74))
    ref = _assert_indices_equal_oracle(basis, atoms)
    # the probe does straddle the cut-offs: of the pairs placed around each radius some are in and some are out
    pos, cell = atoms.get_positions(), np.asarray(atoms.cell)[0, 0]
    d = pos[1::2] - pos[0::2]
    d -= cell * np.round(d / cell)
    r = np.sqrt((d * d).sum(axis=1))
    third = len(r) // 3
    n_pair = sum(len(v) for v in ref["pairs"].values())
    inside2 = int((r[:third] < r_max2).sum())
    # (this NumPy distance is not cdist's: which side a pair within an ulp falls on may differ from the oracle's verdict for a few)
    assert 0 < inside2 < third and abs(n_pair - 2 * (inside2 + 2 * third)) <= third // 4
    in3 = int(((r[third:] > r_min3) & (r[third:] <= r_max3)).sum())
    assert 0 < in3 < 2 * third and abs(len(ref["n3"]) - 2 * in3) <= third // 4
    # and the rows of that frame agree as well
    fz = process.BasisFeaturizer(basis)
    x_e, x_f, _ = fz.featurize_frames([atoms])
    o = O.featurize(O.OracleBasis(basis), atoms)
    assert rel_err(x_e[0], o["xe"]) < TOL and rel_err(x_f, o["xf"]) < TOL


def test_hand_off_launch_pair_matches_the_one_kernel_launch_and_the_oracle():
    """UF3_F3_HANDOFF=1 (the measured experiment of round 6, DESIGN 3.6: k_feat3_w computes the neighbour role's stage-1 sums
    once, at the centre, and k_featurize3<HO> reads them back): same rows as the default launch and as the oracle -- one, two and
    three species, list capacities 16 and 24, several frames per slice and one frame per slice."""
    cases = [(synthetic.notebook_basis(['W']), [synthetic.config_c2()[0]]),
             (synthetic.notebook_basis(['Mo', 'W']), [synthetic.lattice_frame("bcc", (6, 6, 6), 3.165, [42, 74], s) for s in (1, 2, 3)]),
             (synthetic.notebook_basis(['Mo', 'Nb', 'W']), [synthetic.lattice_frame("bcc", (6, 6, 6), 3.2, [41, 42, 74], 5)]),
             (synthetic.notebook_basis(['W']), [synthetic.lattice_frame("fcc", (6, 6, 6), 3.9, [74], 7)])]
    for basis, frames in cases:
        fz = process.BasisFeaturizer(basis)
        e0, f0, off = fz.featurize_frames(frames)
        for slice_atoms in ("", "1"):
            os.environ["UF3_F3_HANDOFF"] = "1"
            if slice_atoms:
                os.environ["UF3_F3_SLICE"] = slice_atoms
            try:
                e1, f1, _ = fz.featurize_frames(frames)
            finally:
                os.environ.pop("UF3_F3_HANDOFF", None)
                os.environ.pop("UF3_F3_SLICE", None)
            assert rel_err(e1, e0) < 1e-12 and rel_err(f1, f0) < 1e-12
        ref = O.featurize(O.OracleBasis(basis), frames[-1])
        assert rel_err(f1[off[-2]:off[-1]], ref["xf"]) < TOL and rel_err(e1[-1], ref["xe"]) < TOL


def test_sharded_evaluator_and_feature_batch_on_one_gpu():
    """The package's device-resident N > 1 drivers at world size 1: parallel.ShardedEvaluator (whole-frame
    evaluator into the flat buffer, MD route, positions moved on the device) against UFCalculator and the oracle, and
    parallel.featurize_sharded's DeviceFeatureBatch against featurize_frames."""
    import torch
    from uf3_amd import parallel
    basis = synthetic.notebook_basis(['Mo', 'W'])
    atoms = synthetic.lattice_frame("bcc", (6, 7, 8), 3.165, [42, 74], seed=77)
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(5).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    calc = calculator.UFCalculator(model)
    e1, f1, _, v1 = calc.evaluate_frames([atoms], virial=True)
    ev = parallel.ShardedEvaluator(calc, atoms, md_skin=0.5)
    assert ev.device_route and not ev.decomposed and (ev.lo, ev.hi) == (0, len(atoms))
    e, f, v = ev.step().result()
    assert abs(e - e1[0]) <= 1e-12 * abs(e1[0]) and rel_err(f, f1) < 1e-12 and np.allclose(v, v1[0], rtol=1e-10, atol=1e-10)
    rng = np.random.default_rng(9)
    for _ in range(5):                                       # an MD loop on the device: the lists persist, the atoms move
        ev.positions.add_(torch.from_numpy(rng.uniform(-0.02, 0.02, (len(atoms), 3))).to(ev.positions.device))
        ev.step()
    e, f, v = ev.result()
    moved = Atoms(numbers=atoms.get_atomic_numbers(), positions=ev.host_positions(), cell=atoms.get_cell(), pbc=True)
    ev.close()
    e_ref, f_ref = O.evaluate(O.OracleBasis(basis), moved, coeff)
    assert abs(e - e_ref) <= TOL * abs(e_ref) and worst_elementwise(f, f_ref) <= 1.0
    # the featurize-only fan-out: this rank's block (all frames at world size 1), rows resident on the device
    frames = [synthetic.lattice_frame("bcc", (4, 4, 4 + k), 3.165, [42, 74], seed=20 + k) for k in range(3)]
    fz = process.BasisFeaturizer(basis)
    fb, (lo, hi) = parallel.featurize_sharded(fz, frames)
    assert (lo, hi) == (0, 3)
    fb.run()
    x_e, x_f, off = fz.featurize_frames(frames)
    assert rel_err(fb.x_e.cpu().numpy(), x_e) < 1e-12 and rel_err(fb.x_f.cpu().numpy(), x_f) < 1e-12
    fb2, (lo2, hi2) = parallel.featurize_sharded(fz, lambda i: frames[i], n_frames=3, rank=1, world_size=2, ld=-1)
    assert (lo2, hi2) == (2, 3) and fb2.ld % 16 == 0
    fb2.run()
    assert rel_err(fb2.x_f.cpu().numpy(), x_f[off[2]:off[3]]) < 1e-12


_BAR_ALTERNATION = r'''
import numpy as np, sys
from uf3_amd import synthetic
from uf3_amd.forcefield import calculator
from uf3_amd.regression import least_squares as ls
basis = synthetic.notebook_basis(['W'])
model = ls.WeightedLinearModel(basis)
coeff = np.random.default_rng(5).normal(0, 0.05, basis.n_feats)
coeff[basis.col_idx] = 0.0
model.coefficients = coeff
a = synthetic.lattice_frame("bcc", (4, 4, 4), 3.165, [74], seed=3)
b = a.copy()
b.positions = a.positions + np.random.default_rng(1).uniform(-0.05, 0.05, a.positions.shape)
plain = calculator.UFCalculator(model, md_skin=0.0)
want = [plain.evaluate_frames([x]) for x in (a, b)]
calc = calculator.UFCalculator(model, md_skin=0.5)          # the MD route: small batches staged by the host, steps waited for on a polled word
bad = 0
for step in range(400):
    k = step & 1
    e, f, _ = calc.evaluate_frames([(a, b)[k]])
    bad += not (abs(e[0] - want[k][0][0]) <= 1e-12 * abs(want[k][0][0]) and np.abs(f - want[k][1]).max() <= 1e-12 * np.abs(want[k][1]).max())
print("BAD", bad)
sys.exit(1 if bad else 0)
'''


@pytest.mark.parametrize("env", [{}, {"UF3_BAR_FORCE_FAIL": "1"}, {"UF3_NO_BAR_STAGE": "1"}])
def test_alternating_positions_through_the_staged_block_for_many_steps(env):
    """the host stores an MD step's positions straight into a fine-grained device block (BAR staging, gated on
    bar_self_test) and waits for the step on a polled status word, no stream synchronisation in between -- a stale cache line would
    make a step silently use the previous step's positions.  400 consecutive steps alternate between two position sets; every
    result must be the one of ITS positions.  Also with the self-test forced to fail / the path switched off (the pinned-block
    route), each in a fresh process and context."""
    import subprocess
    import sys
    full = dict(os.environ)
    full.update(env)
    full["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + full.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", _BAR_ALTERNATION], env=full, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BAD 0" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_lds_table_instances_of_the_md_evaluator_agree_with_the_others():
    """k_eval<..., MD, TAB, CW> (round 6: window coefficients, knot records and pair coefficients in LDS, contraction over the kept-bin
    rows, neighbour forces gathered instead of added atomically, strain derivative from the gathered forces, DPP sums) against the
    instances without the tables (UF3_EVAL_NO_CW=1), the rebuild-everything route and the oracle: one, two and three species, with
    and without the strain derivative, over several moved steps."""
    for elements, numbers, reps in ((['W'], [74], (5, 5, 5)), (['Mo', 'W'], [42, 74], (6, 5, 4)), (['V', 'Mo', 'W'], [23, 42, 74], (5, 6, 7))):
        basis = synthetic.notebook_basis(elements)
        atoms = synthetic.lattice_frame("bcc", reps, 3.165, numbers, seed=31)
        model = ls.WeightedLinearModel(basis)
        coeff = np.random.default_rng(2).normal(0, 0.05, basis.n_feats)
        coeff[basis.col_idx] = 0.0
        model.coefficients = coeff
        plain = calculator.UFCalculator(model, md_skin=0.0)
        rng = np.random.default_rng(4)
        results = {}
        for label, env in (("cw", {}), ("nocw", {"UF3_EVAL_NO_CW": "1"})):
            os.environ.update(env)
            try:
                calc = calculator.UFCalculator(model, md_skin=0.5)
                a = atoms.copy()
                out = []
                for step in range(4):
                    a.positions = a.positions + np.random.default_rng(10 + step).uniform(-0.02, 0.02, a.positions.shape)
                    r = calc.evaluate_frames([a], virial=(step % 2 == 1))
                    out.append((r[0], r[1], None, r[3] if len(r) > 3 else None))
                results[label] = (out, a)
            finally:
                for k in env:
                    os.environ.pop(k, None)
        for step in range(4):
            e1, f1, _, v1 = results["cw"][0][step]
            e0, f0, _, v0 = results["nocw"][0][step]
            assert abs(e1[0] - e0[0]) <= 1e-12 * abs(e0[0]) and rel_err(f1, f0) < 1e-12
            if v0 is not None:
                assert np.allclose(v1, v0, rtol=1e-10, atol=1e-10 * np.abs(v0).max())
        a = results["cw"][1]
        e1, f1, _, _ = results["cw"][0][3]
        ep, fp, _, vp = plain.evaluate_frames([a], virial=True)
        assert abs(e1[0] - ep[0]) <= 1e-12 * abs(ep[0]) and rel_err(f1, fp) < 1e-12
        assert np.allclose(results["cw"][0][3][3], vp, rtol=1e-10, atol=1e-10 * np.abs(vp).max())
        e_ref, f_ref = O.evaluate(O.OracleBasis(basis), a, coeff)
        assert abs(e1[0] - e_ref) <= TOL * abs(e_ref) and worst_elementwise(f1, f_ref) <= 1.0


def test_fit_chunk_plan_ramps_then_spreads_the_rest_evenly():
    """uf3_fit_add plans a call's chunks ahead (round 6): a ramp from an eighth of the limit, doubling, then the remaining frames
    spread evenly over the fewest chunks of at most the limit -- no small tail chunk -- and the pieces do not depend on the plan;
    frames without atoms are refused."""
    from uf3_amd import pipeline
    basis = synthetic.notebook_basis(['W'])
    frames = [synthetic.lattice_frame("bcc", (2, 2, 2), 3.165, [74], seed=500 + k) for k in range(128)]       # 16 atoms each
    rng = np.random.default_rng(1)
    energies = rng.normal(size=len(frames))
    forces = [rng.normal(size=(16, 3)) for _ in frames]
    model = ls.WeightedLinearModel(basis)
    fz = process.BasisFeaturizer(basis)
    planned = pipeline.NativeFitAccumulator(model, fz, max_atoms_per_chunk=512)          # 32 frames per full chunk
    planned.add_frames(frames, energies, forces)
    assert planned.n_chunks == 7                    # 4 + 8 + 16 frames, then 100 frames as 4 x 25 (not 3 x 32 + 4)
    one = pipeline.NativeFitAccumulator(model, fz, max_atoms_per_chunk=1 << 20)
    one.ctx.check(one.ctx.lib.uf3_fit_first_chunk(one.handle, 1.0))
    one.add_frames(frames, energies, forces)
    assert one.n_chunks == 1
    a, b = planned.pieces(), one.pieces()
    for key in a:
        assert rel_err(a[key], b[key]) < 1e-11, key
    odd = pipeline.NativeFitAccumulator(model, fz, max_atoms_per_chunk=512)              # 37 frames: 4 + 8 + 16 + 9
    odd.add_frames(frames[:37], energies[:37], forces[:37])
    assert odd.n_chunks == 4
    from uf3_amd.data.atoms import Atoms as _A
    empty = _A(numbers=np.zeros(0, dtype=int), positions=np.zeros((0, 3)), cell=np.eye(3) * 5, pbc=True)
    with pytest.raises(_lib.UF3Error):
        odd.add_frames([frames[0], empty], energies[:2], [forces[0], np.zeros((0, 3))])


def test_md_route_frame_sums_from_the_collection_pass_on_a_large_frame():
    """One whole frame of >= 8192 atoms on the MD route: the collection pass leaves per-workgroup sums of the atoms' energies and
    strain derivatives and k_frame_sum adds those (round 6) -- energy, forces and strain derivative against the rebuild-everything
    route (per-atom sums) and the oracle, on moved atoms."""
    atoms, basis = synthetic.config_c4(frame=2)
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(8).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    md = calculator.UFCalculator(model, md_skin=0.5)
    plain = calculator.UFCalculator(model, md_skin=0.0)
    a = atoms.copy()
    for step in range(3):
        a.positions = a.positions + np.random.default_rng(40 + step).uniform(-0.02, 0.02, a.positions.shape)
        e1, f1, _, v1 = md.evaluate_frames([a], virial=True)
    e0, f0, _, v0 = plain.evaluate_frames([a], virial=True)
    assert abs(e1[0] - e0[0]) <= 1e-12 * abs(e0[0]) and rel_err(f1, f0) < 1e-12
    assert np.allclose(v1, v0, rtol=1e-10, atol=1e-10 * np.abs(v0).max())
    e2, f2, _ = md.evaluate_frames([a])                       # (without the strain derivative: the energy sums alone)
    assert abs(e2[0] - e0[0]) <= 1e-12 * abs(e0[0]) and rel_err(f2, f0) < 1e-12
    e_ref, f_ref = O.evaluate(O.OracleBasis(basis), a, coeff)
    assert abs(e1[0] - e_ref) <= TOL * abs(e_ref) and worst_elementwise(f1, f_ref) <= 1.0
