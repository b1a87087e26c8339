"""CPU-side checks: the C ABI library loads and exports the declared symbols; host logic."""
import os
import re
import json

import numpy as np
import pytest

from uf3_amd import _lib
from uf3_amd.data.atoms import Atoms
from uf3_amd.data import geometry
from uf3_amd.regression import least_squares as ls
from uf3_amd.representation import process
from oracle import oracle as O
from _util import GOLDEN, basis_from_meta, load_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "uf3_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(uf3_[a-z0-9_]+)\s*\(", header)))
    assert set(declared) == set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_binary_carries_the_hash_of_the_sources_it_was_built_from():
    """uf3_build_id(): the first 16 hex digits of the sha256 over uf3_hip.hip + headers the binary was compiled from (the Makefile
    passes it in).  After __graft_entry__.build() it is the hash of the tree's sources -- a pushed .so that no longer matches its
    sources cannot pass for a build of them."""
    want = _lib.source_build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", want)
    import __graft_entry__ as g
    so = os.path.join(ROOT, "uf3_amd", "csrc", "libuf3hip.so")
    have = g._binary_build_id(so)
    assert have == _lib.build_id()
    assert have == want, f"libuf3hip.so was built from other sources ({have}) than the tree holds ({want}): run __graft_entry__.build()"
    # a changed source changes the id
    import hashlib
    h = hashlib.sha256()
    for name in _lib.SOURCES:
        h.update(open(os.path.join(ROOT, "uf3_amd", "csrc", name), "rb").read())
    assert h.hexdigest()[:16] == want
    h.update(b" ")
    assert h.hexdigest()[:16] != want


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d, meta, atoms = load_case("case_h2o")
    fz = process.BasisFeaturizer(basis_from_meta(meta))
    with pytest.raises(_lib.HipUnavailable):
        fz.featurize_energy_2B(atoms)
    with pytest.raises(_lib.HipUnavailable):
        ls.WeightedLinearModel(basis_from_meta(meta)).fit(np.ones((3, 196)), np.ones(3))


def test_flatten_basis_lut_matches_compress():
    d, meta, atoms = load_case("case_steel")
    basis = basis_from_meta(meta)
    spec, keep = _lib.flatten_basis(basis)
    assert spec.n_feat == 609 and spec.n_trios == 6 and spec.n_pairs == 3
    off = 0
    rng = np.random.default_rng(3)
    for k, trio in enumerate(keep["trios"]):
        shape = basis.templates[trio].shape
        n = int(np.prod(shape))
        lut = keep["trio_lut"][off:off + n]
        off += n
        grid = rng.random(shape)
        got = np.zeros(keep["trio_ncol"][k])
        np.add.at(got, lut[lut >= 0], grid.ravel()[lut >= 0])
        assert np.allclose(got, basis.compress_3B(grid, trio), rtol=1e-14)


def test_supercell_order_matches_oracle():
    rng = np.random.default_rng(0)
    cell = np.array([[6.0, 0.5, 0.0], [0.3, 5.0, 0.2], [0.0, 0.4, 7.0]])
    atoms = Atoms(numbers=[74] * 5, positions=rng.uniform(0, 5, (5, 3)), cell=cell, pbc=[True, True, False])
    sup = geometry.get_supercell(atoms, r_cut=5.5)
    pos, z, shifts = O.supercell(atoms, 5.5)
    assert np.allclose(sup.get_positions(), pos, rtol=0, atol=1e-12)
    assert np.array_equal(geometry.image_shifts(cell, atoms.get_pbc(), 5.5), shifts)


def test_fit_with_gram_matches_reference_capture():
    """Host part of the fit (freeze, regulariser, solve, coverage) on Gram pieces built in the test."""
    d = np.load(os.path.join(GOLDEN, "fit_case.npz"))
    basis = basis_from_meta(json.loads(str(d["meta"])))
    model = ls.WeightedLinearModel(basis, regularizer=d["regularizer"])
    pieces = dict(gram_e=d["gram_e"], ord_e=d["ord_e"], gram_f=d["gram_f"], ord_f=d["ord_f"],
                  m_e=ls.moments(d["y_e"]), m_f=ls.moments(d["y_f"]))
    w = ls.calc_E_F_weights(len(d["y_e"]), len(d["y_f"]), ls.std_from_moments(pieces["m_e"]),
                            ls.std_from_moments(pieces["m_f"]))
    assert np.allclose(w, d["weights"], rtol=1e-12)
    model.fit_from_pieces(pieces, weight=float(d["kappa"][0]))
    assert np.allclose(model.coefficients, d["coefficients"], rtol=1e-7, atol=1e-9)
    assert np.array_equal(model.data_coverage, d["data_coverage"])
    assert np.allclose(model.predict(d["x_e"]), d["predict_e"], rtol=1e-8, atol=1e-8)


def test_model_json_roundtrip(tmp_path):
    model = ls.WeightedLinearModel.from_json(os.path.join(GOLDEN, "model_unary.json"))
    out = tmp_path / "m.json"
    model.to_json(str(out))
    again = ls.WeightedLinearModel.from_json(str(out))
    assert np.allclose(again.coefficients, model.coefficients, rtol=1e-15)
    assert again.bspline_config.get_column_names() == model.bspline_config.get_column_names()


def test_unknown_element_warns_and_returns_empty():
    d, meta, atoms = load_case("case_h2o")
    fz = process.BasisFeaturizer(basis_from_meta(meta))
    other = Atoms("Ar2", positions=[[0, 0, 0], [3, 0, 0]])
    with pytest.warns(RuntimeWarning):
        assert fz.evaluate_configuration(other, energy=1.0) == {}


def test_supercell_argument_accepts_lattice_tilings_only():
    """``featurize_*(geom, supercell)``: the reference takes any supercell; features only depend on it through the images
    inside the cut-off, so every tiling by whole lattice images covering ``r_cut`` is accepted (larger r_cut, sorted
    images), anything else is refused (host logic, no GPU involved)."""
    import numpy as np
    import pytest
    from uf3_amd import synthetic
    from uf3_amd.data import geometry
    from uf3_amd.data.atoms import Atoms
    from uf3_amd.representation import process
    atoms = synthetic.lattice_frame("bcc", (2, 2, 2), 3.165, [74], seed=4)
    fz = process.BasisFeaturizer(synthetic.notebook_basis(['W']))
    check = fz._supercell_means_periodic
    assert check(atoms, None) is False and check(atoms, atoms) is False and check(atoms, atoms.copy()) is False
    ref = geometry.get_supercell(atoms, r_cut=fz.r_cut)
    assert check(atoms, ref) is None
    assert check(atoms, geometry.get_supercell(atoms)) is None                                  # r_cut = 10
    assert check(atoms, geometry.get_supercell(atoms, r_cut=fz.r_cut, sort_indices=True)) is None
    n = len(atoms)
    with pytest.raises(ValueError):                                                              # too few images
        check(atoms, Atoms(numbers=ref.get_atomic_numbers()[:5 * n], positions=ref.get_positions()[:5 * n]))
    with pytest.raises(ValueError):
        check(atoms, Atoms(numbers=ref.get_atomic_numbers()[:-1], positions=ref.get_positions()[:-1]))
    moved = ref.get_positions().copy()
    moved[len(atoms) + 3] += 0.01                                                                # not a rigid image
    with pytest.raises(ValueError):
        check(atoms, Atoms(numbers=ref.get_atomic_numbers(), positions=moved))
    # atoms outside the cell: the reference's rows depend on the image range, so only its own range is accepted
    pos = atoms.get_positions().copy()
    pos[0] += np.array(atoms.get_cell())[0] * 1.5
    out = Atoms(numbers=atoms.get_atomic_numbers(), positions=pos, cell=atoms.get_cell(), pbc=True)
    assert check(out, geometry.get_supercell(out, r_cut=fz.r_cut)) is None
    with pytest.raises(ValueError):
        check(out, geometry.get_supercell(out, r_cut=3 * fz.r_cut))


def test_bench_line_keeps_the_other_configurations_where_the_driver_looks():
    """bench.py: the sub-lines of the other BASELINE configurations are repeated, compact, inside `roofline` (the driver's
    parser drops `extra`), and the CPU baseline carries the calibrated estimate for the NumPy reference."""
    import bench
    extra = {"fit_w": dict(value=6900.0, unit="frames/s", ms_per_step=18.5,
                           roofline=dict(fp64=dict(frac=0.27), hbm=dict(frac=0.0004))),
             "eval_50k": dict(value=1.1e8, unit="atom-steps/s", ms_per_step=0.45, roofline=dict(fp64=dict(frac=0.1), hbm=dict(frac=0.001)),
                              cpu_baseline=dict(value=5700, unit="atom-steps/s")),
             "lead0": dict(error="RuntimeError: boom")}
    table = bench.compact_configs(extra)
    assert table["fit_w"] == [6900.0, "frames/s", 18.5, 0.27, 0.0004]
    assert table["eval_50k_cpu_port"][:2] == [5700, "atom-steps/s"] and table["lead0"][0] == "error"
    assert table["columns"] == ["value", "unit", "ms_per_step", "fp64_frac", "hbm_frac"]
    est = bench.reference_estimate(0.4, 4.8, 32)
    assert est is None or (est["label"] == "estimate" and est["frames_per_s_all_cores"] == pytest.approx(4.8 / est["reference_to_port_ratio"]))


def test_single_frame_batch_is_refreshed_in_place():
    """_lib.FrameBatch.refresh (what UFCalculator does between the calls of an MD loop): positions, species, cell and boundary
    flags of the same-sized frame land in the arrays the C struct already points to; another atom count asks for a new batch."""
    from uf3_amd import synthetic
    from uf3_amd.data.atoms import Atoms
    a = synthetic.lattice_frame("bcc", (2, 2, 2), 3.165, [42, 74], seed=1)
    batch = _lib.FrameBatch([a])
    where = (batch.pos.ctypes.data, batch.z.ctypes.data, batch.cells.ctypes.data, batch.pbc.ctypes.data)
    b = Atoms(numbers=a.get_atomic_numbers()[::-1], positions=a.get_positions() + 0.25, cell=np.asarray(a.get_cell()) * 1.1, pbc=[True, False, True])
    assert batch.refresh(b)
    assert (batch.pos.ctypes.data, batch.z.ctypes.data, batch.cells.ctypes.data, batch.pbc.ctypes.data) == where
    fresh = _lib.FrameBatch([b])
    for name in ("pos", "z", "cells", "pbc", "offsets"):
        assert np.array_equal(getattr(batch, name), getattr(fresh, name)), name
    assert batch.z.dtype == np.int32 and batch.struct.n_frames == 1
    assert not batch.refresh(synthetic.lattice_frame("bcc", (2, 2, 3), 3.165, [42, 74], seed=1))      # another size
    assert not _lib.FrameBatch([a, b]).refresh(a)                                                         # not a single frame
