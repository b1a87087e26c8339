#!/usr/bin/env python3
"""
Golden-vector capture.  Runs ONLY in the build container, where the Python
reference is mounted at /root/reference (it never travels to the GPU box):

    python tests/golden/make_golden.py

It imports the reference behind the stand-ins in tests/golden/_standins (ase /
numba / tables / ndsplines are not installed here; see the README there), calls
the reference's own functions on small seeded inputs and writes inputs +
expected outputs as data files next to this script.  It also copies, verbatim,
the data fixtures the reference's tests hold (rattled_steel_features.json, the
model JSONs, test.xyz) and extracts the literal known-answer vectors of
tests/test_representation.py and tests/test_calculator.py.

Nothing in the product, the GPU tests, smoke() or bench.py imports this script
or reads /root/reference.
"""
import json
import os
import shutil
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, REF)
warnings.simplefilter("ignore")

import ase  # noqa: E402  (stand-in)
from uf3.data import composition as rc  # noqa: E402
from uf3.data import geometry as rg  # noqa: E402
from uf3.representation import bspline as rb, process as rp, distances as rd, angles as ra  # noqa: E402
from uf3.regression import least_squares as rl  # noqa: E402
from uf3.forcefield import calculator as rcalc  # noqa: E402
from ase import symbols as asym  # noqa: E402


def jkey(t):
    return "-".join(t) if isinstance(t, tuple) else t


def basis_kwargs_json(kw):
    out = {}
    for k, v in kw.items():
        if isinstance(v, dict):
            out[k] = {jkey(a) if not isinstance(a, int) else str(a): (list(b) if isinstance(b, (list, tuple)) else b)
                      for a, b in v.items()}
        else:
            out[k] = v
    return out


def read_xyz(path):
    frames = []
    with open(path) as f:
        lines = f.read().splitlines()
    i = 0
    while i < len(lines) and lines[i].strip():
        n = int(lines[i])
        hdr = lines[i + 1]
        lat = np.array(hdr.split('Lattice="')[1].split('"')[0].split(), dtype=float).reshape(3, 3)
        energy = float(hdr.split("energy=")[1].split()[0])
        rows = [ln.split() for ln in lines[i + 2:i + 2 + n]]
        sym = [r[0] for r in rows]
        pos = np.array([[float(x) for x in r[1:4]] for r in rows])
        frc = np.array([[float(x) for x in r[4:7]] for r in rows])
        frames.append(dict(symbols=sym, positions=pos, cell=lat, energy=energy, forces=frc))
        i += 2 + n
    return frames


def reference_rows(geom, basis, forces=True):
    """Energy row / force rows exactly as evaluate_configuration assembles them (no y)."""
    fz = rp.BasisFeaturizer(basis)
    sup = rg.get_supercell(geom, r_cut=basis.r_cut) if any(geom.pbc) else geom
    parts = [basis.chemical_system.get_composition_tuple(geom).astype(float),
             fz.featurize_energy_2B(geom, sup)]
    if basis.degree > 2:
        parts.append(fz.featurize_energy_3B(geom, sup))
    out = dict(xe=np.concatenate(parts))
    if forces:
        parts = [np.zeros((len(geom), 3, len(basis.element_list))), fz.featurize_force_2B(geom, sup)]
        if basis.degree > 2:
            parts.append(fz.featurize_force_3B(geom, sup))
        out["xf"] = np.concatenate(parts, axis=2)
    # neighbour indices on the untrimmed supercell
    dm = rd.get_distance_matrix(geom, sup)
    gz, sz = np.array(geom.get_atomic_numbers()), np.array(sup.get_atomic_numbers())
    for p, pair in enumerate(basis.interactions_map[2]):
        cm = rd.mask_matrix_by_pair_interaction(asym.symbols2numbers(pair), gz, sz)
        cut = (dm > max(basis.r_min_map[pair], 0)) & (dm < basis.r_max_map[pair])
        i, j = np.where(cm & cut)
        out[f"pair{p}_ij"] = np.stack([i, j], axis=1).astype(np.int64)
    if basis.degree > 2:
        ks = [basis.knots_map[t] for t in basis.interactions_map[3]]
        _, iw, jw = ra.identify_ij(geom, ks, sup)
        out["n3_ij"] = np.stack([iw, jw], axis=1).astype(np.int64)
    out["n_supercell"] = np.array([len(sup)], dtype=np.int64)
    return out


def save_case(name, geom, els, degree, kw, forces=True, extra=None):
    basis = rb.BSplineBasis(rc.ChemicalSystem(els, degree), **kw)
    out = reference_rows(geom, basis, forces=forces)
    meta = dict(element_list=list(els), degree=degree, basis_kwargs=basis_kwargs_json(kw),
                partition_sizes=[int(x) for x in basis.partition_sizes], r_cut=float(basis.r_cut))
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        positions=geom.get_positions(), numbers=geom.get_atomic_numbers(),
                        cell=np.array(geom.get_cell()), pbc=np.array(geom.get_pbc()),
                        meta=json.dumps(meta), **out, **(extra or {}))
    print("wrote", name, "F =", len(out["xe"]), "atoms =", len(geom))
    return basis, out


def w_notebook_kwargs(lead3=3):
    return dict(r_min_map={('W', 'W'): 0.001, ('W', 'W', 'W'): [1.5, 1.5, 1.5]},
                r_max_map={('W', 'W'): 5.5, ('W', 'W', 'W'): [3.5, 3.5, 7.0]},
                resolution_map={('W', 'W'): 15, ('W', 'W', 'W'): [6, 6, 12]},
                leading_trim={2: 0, 3: lead3}, trailing_trim={2: 3, 3: 3})


def bcc_cell(rng, reps, a=3.165, rattle=0.08, strain=0.01):
    base = np.array([[0, 0, 0], [0.5, 0.5, 0.5]])
    pts = []
    for i in range(reps[0]):
        for j in range(reps[1]):
            for k in range(reps[2]):
                pts.extend((base + [i, j, k]) * a)
    pos = np.array(pts) + rng.normal(0, rattle, (len(pts), 3))
    cell = np.diag(np.array(reps) * a) @ (np.eye(3) + rng.uniform(-strain, strain, (3, 3)))
    return pos, cell


def main():
    # ---------------------------------------------------------------- verbatim data fixtures
    for src in ["tests/data/precalculated_ref/rattled_steel_features.json",
                "tests/data/precalculated_ref/model_unary.json",
                "tests/data/precalculated_ref/model_binary.json",
                "tests/data/extended_xyz/test.xyz",
                "examples/tungsten_extxyz/model_2and3.json"]:
        shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, os.path.basename(src)))

    # ---------------------------------------------------------------- literal vectors of the reference tests
    sys.path.insert(0, os.path.join(REF, "tests"))
    import test_representation as tr

    def fixture_value(name):
        fx = getattr(tr, name)
        fn = getattr(fx, "_get_wrapped_function", None)
        fn = fn() if fn else getattr(fx, "__wrapped__", fx)
        return next(fn())

    lit = {}
    for name in ["strained_H2O_molecule_feature", "strained_H2O_molecule_feature_old",
                 "methane_feature", "methane_feature_old"]:
        v = fixture_value(name)
        enc = {"2": {jkey(k): np.asarray(a).tolist() for k, a in v[2].items()},
               "3": {jkey(k): {"position": np.asarray(a["position"]).tolist(),
                               "value": np.asarray(a["value"]).tolist()} for k, a in v[3].items()}}
        lit[name] = enc
    json.dump(lit, open(os.path.join(HERE, "literal_features.json"), "w"))
    print("wrote literal_features.json")

    # ---------------------------------------------------------------- host basis layer
    host = {}
    cfgs = [("W2", ['W'], 2, {}), ("W3", ['W'], 3, {}), ("W3_notebook", ['W'], 3, w_notebook_kwargs()),
            ("W3_lead0", ['W'], 3, w_notebook_kwargs(0)),
            ("NeXe3", ['Ne', 'Xe'], 3, {}), ("NeXe3_lead0", ['Ne', 'Xe'], 3, dict(leading_trim=0, trailing_trim=3)),
            ("AlCuZr3", ['Al', 'Cu', 'Zr'], 3, {}),
            ("W3_sym3", ['W'], 3, dict(r_min_map={('W', 'W', 'W'): [1.5] * 3}, r_max_map={('W', 'W', 'W'): [4.0] * 3},
                                       resolution_map={('W', 'W', 'W'): [5, 5, 5]}, leading_trim=0, trailing_trim=3)),
            ("W3_sym1", ['W'], 3, dict(r_min_map={('W', 'W', 'W'): [1.5] * 3}, r_max_map={('W', 'W', 'W'): [3.0, 4.0, 6.0]},
                                       resolution_map={('W', 'W', 'W'): [4, 5, 7]}, leading_trim=0, trailing_trim=3)),
            ("NeXe2_lammps", ['Ne', 'Xe'], 2, dict(knot_strategy='lammps'))]
    arrays = {}
    for name, els, deg, kw in cfgs:
        b = rb.BSplineBasis(rc.ChemicalSystem(els, deg), **kw)
        host[name] = dict(element_list=els, degree=deg, basis_kwargs=basis_kwargs_json(kw),
                          interactions=[jkey(i) for i in b.interactions],
                          partition_sizes=[int(x) for x in b.partition_sizes], r_cut=float(b.r_cut),
                          columns=b.get_column_names(), col_idx=b.col_idx.tolist(),
                          symmetry={jkey(k): int(v) for k, v in b.symmetry.items()},
                          hashes={str(d): [int(h) for h in b.chemical_system.interaction_hashes[d]]
                                  for d in range(2, deg + 1)})
        for k, v in b.knots_map.items():
            arrays[f"{name}|knots|{jkey(k)}"] = np.array(v if not isinstance(v, list) else np.concatenate(v))
        for k in b.symmetry:
            arrays[f"{name}|mask|{jkey(k)}"] = b.template_mask[k]
            arrays[f"{name}|weights|{jkey(k)}"] = b.flat_weights[k]
        reg = b.get_regularization_matrix(ridge_map={}, curvature_map={}, ridge_1b=1e-8, ridge_2b=0.0,
                                          ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=1e-6)
        if reg.size < 400000:
            arrays[f"{name}|regularizer"] = reg
    json.dump(host, open(os.path.join(HERE, "host_basis.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "host_basis.npz"), **arrays)
    print("wrote host_basis")

    # known-answer B-spline values (tests/test_bsplines.py:529-547 style): basis elements via scipy
    t = rb.generate_uniform_knots(1.0, 6.0, 10)
    x = np.linspace(0.5, 6.5, 241)
    vals = np.array([[np.nan_to_num(bf(x, nu=nu)) for bf in rb.generate_basis_functions(rb.get_knot_subintervals(t))]
                     for nu in (0, 1)])
    np.savez_compressed(os.path.join(HERE, "bspline_values.npz"), knots=t, x=x, values=vals)

    # ---------------------------------------------------------------- feature rows + neighbour indices
    rng = np.random.default_rng(20240928)
    # (a) rattled steel: inputs of tests/test_representation.py:605-648 (expected = the copied JSON)
    steel_pos = [[1.99342831e-01, 7.23471398e-02, 2.29537708e-01], [3.27460597e+00, 3.16932506e-03, -9.68273914e-02],
                 [3.65842563e-01, 3.07348695e+00, -1.43894877e-01], [3.02851201e+00, 2.85731646e+00, 6.85404929e-03],
                 [-1.60754569e-03, -3.82656049e-01, 2.57501643e+00], [2.80754249e+00, -3.02566224e-01, 2.88284947e+00],
                 [-8.16048151e-02, 2.53753926e+00, 3.26312975e+00], [2.92484474e+00, 2.93350564e+00, 2.58505036e+00],
                 [1.32612346e+00, 1.45718452e+00, -1.80198715e-01], [1.51013960e+00, -7.01277380e-02, 1.37666125e+00],
                 [-7.03413224e-02, 1.80545564e+00, 1.43230056e+00]]
    steel = ase.Atoms('Fe8C3', positions=steel_pos, pbc=True, cell=[5.74, 5.74, 5.74])
    trios = [('Fe', 'Fe', 'Fe'), ('Fe', 'Fe', 'C'), ('Fe', 'C', 'C'), ('C', 'Fe', 'Fe'), ('C', 'Fe', 'C'), ('C', 'C', 'C')]
    pairs = [('Fe', 'Fe'), ('Fe', 'C'), ('C', 'C')]
    steel_kw = dict(r_min_map={**{p: 0.1 for p in pairs}, **{t: [1.5] * 3 for t in trios}},
                    r_max_map={**{p: 6.0 for p in pairs}, **{t: [5.0, 5.0, 10.0] for t in trios}},
                    resolution_map={**{p: 12 for p in pairs}, **{t: [4, 4, 8] for t in trios}},
                    knot_strategy='linear', offset_1b=True, leading_trim=0, trailing_trim=3)
    save_case("case_steel", steel, ['Fe', 'C'], 3, steel_kw)

    # (b) molecules of the reference tests
    h2o = ase.Atoms('H2O', positions=[[0, 0, 0], [1.5, 0.0, 0.0], [0, 2.0, 0]], pbc=False, cell=None)
    save_case("case_h2o", h2o, ['H', 'O'], 3, {})
    save_case("case_h2o_lead0", h2o, ['H', 'O'], 3, dict(leading_trim={2: 0, 3: 0}, trailing_trim={2: 3, 3: 3}))
    ch4 = ase.Atoms("CH4", positions=[[15.000000000, 15.000000000, 15.000010729], [15.629117489, 15.629117489, 15.629128218],
                                      [14.370881617, 14.370881617, 15.629128218], [15.629117489, 14.370881617, 14.370892346],
                                      [14.370881617, 15.629117489, 14.370892346]], pbc=True, cell=[30, 30, 30])
    save_case("case_ch4", ch4, ['H', 'C'], 3, {})
    save_case("case_ch4_lead0", ch4, ['H', 'C'], 3, dict(leading_trim={2: 0, 3: 0}, trailing_trim={2: 3, 3: 3}))

    # (c) the five 128-atom W frames of tests/data/extended_xyz/test.xyz: energy rows, notebook basis
    frames = read_xyz(os.path.join(HERE, "test.xyz"))
    for lead in (3, 0):
        rows, idx = [], []
        for fr in frames:
            g = ase.Atoms(fr["symbols"], positions=fr["positions"], pbc=True, cell=fr["cell"])
            b = rb.BSplineBasis(rc.ChemicalSystem(['W'], 3), **w_notebook_kwargs(lead))
            o = reference_rows(g, b, forces=False)
            rows.append(o["xe"])
            idx.append(o)
        meta = dict(element_list=['W'], degree=3, basis_kwargs=basis_kwargs_json(w_notebook_kwargs(lead)))
        np.savez_compressed(os.path.join(HERE, f"case_w128_energy_lead{lead}.npz"), xe=np.array(rows),
                            meta=json.dumps(meta), pair0_ij_frame0=idx[0]["pair0_ij"], n3_ij_frame0=idx[0]["n3_ij"])
        print("wrote case_w128_energy", lead)
    # config 1 (2-body only, trims (0,3)) on the same frames, forces included (2-body force path is fast)
    kw2 = dict(r_min_map={('W', 'W'): 0.001}, r_max_map={('W', 'W'): 5.5}, resolution_map={('W', 'W'): 15},
               leading_trim=0, trailing_trim=3)
    g0 = ase.Atoms(frames[0]["symbols"], positions=frames[0]["positions"], pbc=True, cell=frames[0]["cell"])
    save_case("case_w128_2body", g0, ['W'], 2, kw2, forces=True)

    # (d) rattled bcc-W cells with forces, both trims
    pos, cell = bcc_cell(rng, (2, 2, 2))
    g16 = ase.Atoms(['W'] * 16, positions=pos, pbc=True, cell=cell)
    save_case("case_w16", g16, ['W'], 3, w_notebook_kwargs(3))
    save_case("case_w16_lead0", g16, ['W'], 3, w_notebook_kwargs(0))
    pos, cell = bcc_cell(rng, (3, 3, 3))
    g54 = ase.Atoms(['W'] * 54, positions=pos, pbc=True, cell=cell)
    save_case("case_w54", g54, ['W'], 3, w_notebook_kwargs(3))

    # (e) binary Ne-Xe random fcc-ish cell, ragged neighbours, triclinic strain, atoms outside the cell
    def binary_kwargs(els):
        cs = rc.ChemicalSystem(els, 3)
        return dict(r_min_map={**{p: 0.5 for p in cs.interactions_map[2]}, **{t: [1.5] * 3 for t in cs.interactions_map[3]}},
                    r_max_map={**{p: 6.0 for p in cs.interactions_map[2]}, **{t: [4.5, 4.5, 9.0] for t in cs.interactions_map[3]}},
                    resolution_map={**{p: 15 for p in cs.interactions_map[2]}, **{t: [6, 6, 12] for t in cs.interactions_map[3]}})
    a = 5.0
    fcc = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    pts = np.array([(fcc + [i, j, k]) * a for i in range(2) for j in range(2) for k in range(2)]).reshape(-1, 3)
    cellb = np.diag([2 * a] * 3) @ (np.eye(3) + rng.uniform(-0.03, 0.03, (3, 3)))
    posb = pts + rng.normal(0, 0.15, pts.shape)
    zb = rng.choice(['Ne', 'Xe'], len(pts))
    gb = ase.Atoms(list(zb), positions=posb, pbc=True, cell=cellb)
    save_case("case_nexe32", gb, ['Ne', 'Xe'], 3, binary_kwargs(['Ne', 'Xe']))
    save_case("case_nexe32_lead0", gb, ['Ne', 'Xe'], 3, dict(leading_trim=0, trailing_trim=3, **binary_kwargs(['Ne', 'Xe'])),
              forces=True)
    # partially periodic slab + ternary
    post = rng.uniform(0, 1, (24, 3)) @ (np.diag([7.0, 8.0, 9.0]) + rng.uniform(-0.8, 0.8, (3, 3)))
    cellt = np.diag([7.0, 8.0, 9.0]) + rng.uniform(-0.8, 0.8, (3, 3))
    post = rng.uniform(-0.2, 1.2, (24, 3)) @ cellt
    zt = rng.choice(['Al', 'Cu', 'Zr'], 24)
    gt = ase.Atoms(list(zt), positions=post, pbc=[True, True, False], cell=cellt)
    cs3 = rc.ChemicalSystem(['Al', 'Cu', 'Zr'], 3)
    kwt = dict(r_min_map={**{p: 0.3 for p in cs3.interactions_map[2]}, **{t: [0.8] * 3 for t in cs3.interactions_map[3]}},
               r_max_map={**{p: 5.0 for p in cs3.interactions_map[2]}, **{t: [3.6, 3.6, 7.2] for t in cs3.interactions_map[3]}},
               resolution_map={**{p: 10 for p in cs3.interactions_map[2]}, **{t: [5, 5, 10] for t in cs3.interactions_map[3]}})
    save_case("case_ternary24_slab", gt, ['Al', 'Cu', 'Zr'], 3, kwt)
    # same-species neighbours with symmetry 1 (leg assignment follows supercell index order)
    kws1 = dict(r_min_map={('W', 'W'): 0.5, ('W', 'W', 'W'): [1.0, 1.0, 1.0]},
                r_max_map={('W', 'W'): 5.0, ('W', 'W', 'W'): [3.2, 3.8, 6.4]},
                resolution_map={('W', 'W'): 10, ('W', 'W', 'W'): [4, 5, 8]}, leading_trim=0, trailing_trim=3)
    save_case("case_w16_sym1", g16, ['W'], 3, kws1)
    kws3 = dict(r_min_map={('W', 'W'): 0.5, ('W', 'W', 'W'): [1.0, 1.0, 1.0]},
                r_max_map={('W', 'W'): 5.0, ('W', 'W', 'W'): [4.0, 4.0, 4.0]},
                resolution_map={('W', 'W'): 10, ('W', 'W', 'W'): [5, 5, 5]}, leading_trim=0, trailing_trim=3)
    save_case("case_w16_sym3", g16, ['W'], 3, kws3)

    # ---------------------------------------------------------------- calculator (energies / forces)
    calc_cases = {}

    def record(name, geom, calc, model_file=None, coefficients=None, basis_meta=None, literal=None):
        e = float(calc.get_potential_energy(geom))
        f = np.array(calc.get_forces(geom))
        if literal is not None:  # the numbers asserted by tests/test_calculator.py
            assert np.isclose(e, literal["energy"]), (name, e, literal["energy"])
            assert np.allclose(f, literal["forces"]), name
        calc_cases[name] = dict(positions=geom.get_positions().tolist(), numbers=geom.get_atomic_numbers().tolist(),
                                cell=np.array(geom.get_cell()).tolist(), pbc=[bool(x) for x in geom.get_pbc()],
                                model_file=model_file, coefficients=coefficients, basis=basis_meta,
                                energy=e, forces=f.tolist(), literal=literal)
        print("calc", name, e)

    # test_unary_dimer (LJ-fitted 2-body, lammps knots)
    cs = rc.ChemicalSystem(['W'])
    kwd = dict(r_min_map={('W', 'W'): 2.0}, r_max_map={('W', 'W'): 6.0}, resolution_map={('W', 'W'): 20}, knot_strategy='lammps')
    bd = rb.BSplineBasis(cs, **kwd)
    model = rl.WeightedLinearModel(bspline_config=bd)
    x = np.linspace(2.0, 6.0, 1000)
    y = 4 * 0.87 * ((2.5 / x) ** 12 - (2.5 / x) ** 6)
    cvec = np.insert(rb.fit_spline_1d(x, y, bd.knots_map[('W', 'W')]), 0, 0)
    model.coefficients = cvec
    calc = rcalc.UFCalculator(model)
    meta = dict(element_list=['W'], degree=2, basis_kwargs=basis_kwargs_json(kwd))
    dimer = ase.Atoms('W2', positions=[[0, 0, 0], [1.5, 1.5, 1.5]], pbc=False, cell=None)
    record("unary_dimer_free", dimer, calc, coefficients=cvec.tolist(), basis_meta=meta,
           literal=dict(energy=-1.21578, forces=[[-3.96244881] * 3, [3.96244881] * 3]))
    dimer.set_pbc([True, True, True])
    dimer.set_cell([[3, 0, 0], [3, 5, 0], [0, 0, 3]])
    record("unary_dimer_pbc", dimer, calc, coefficients=cvec.tolist(), basis_meta=meta,
           literal=dict(energy=-15.33335, forces=[[0, -17.3656864, 0], [0, 17.3656864, 0]]))
    # test_unary_trimer / test_unary_pbc / test_binary with the shipped model files
    mu = rl.WeightedLinearModel.from_json(os.path.join(HERE, "model_unary.json"))
    cu = rcalc.UFCalculator(mu)
    trimer = ase.Atoms("W3", positions=[[0, 0, 0], [2, 0, 0], [0, 3, 0]], pbc=False, cell=None)
    record("unary_trimer", trimer, cu, model_file="model_unary.json",
           literal=dict(energy=-18.79979353611411,
                        forces=[[-12.26367499, 0.15140673, 0.], [12.05608935, 0.31137845, 0.], [0.20758563, -0.46278518, 0.]]))
    w8 = ase.Atoms("W8", positions=[[0.00, 0.00, 0.00], [2.89, 0.12, -0.04], [-0.32, 2.71, -0.11], [2.65, 2.81, 0.37],
                                    [0.00, 0.00, 3.00], [2.64, 0.00, 3.00], [-0.08, 2.94, 3.16], [2.53, 2.87, 3.23]],
                   pbc=True, cell=np.eye(3) * 2.74 * 2)
    record("unary_pbc", w8, cu, model_file="model_unary.json",
           literal=dict(energy=-76.358888229785,
                        forces=[[1.36696442, -0.46307, 1.78573347], [0.20112587, 0.17014795, 1.22172728],
                                [-0.66043959, -1.08374173, 6.78845939], [-1.30913745, 0.36888897, 1.48182124],
                                [-0.33315563, 1.28359885, -1.56572912], [0.01504262, 0.06574851, -2.38044283],
                                [0.25436762, 0.2491558, -7.48063062], [0.46523214, -0.59072835, 0.14906119]]))
    mb = rl.WeightedLinearModel.from_json(os.path.join(HERE, "model_binary.json"))
    cb = rcalc.UFCalculator(mb)
    nexe = ase.Atoms("NeXe", positions=[[0, 0, 0], [3.1, 0, 0]], pbc=False)
    record("binary_dimer", nexe, cb, model_file="model_binary.json",
           literal=dict(energy=0.3464031387757268, forces=[[-0.28138023, 0., 0.], [0.28138023, 0., 0.]]))
    # fitted W 2+3-body model of the demo notebook on bcc-W cells (and energy on a 128-atom MD frame)
    m23 = rl.WeightedLinearModel.from_json(os.path.join(HERE, "model_2and3.json"))
    c23 = rcalc.UFCalculator(m23)
    record("w16_model23", g16, c23, model_file="model_2and3.json")
    record("w54_model23", g54, c23, model_file="model_2and3.json")
    e128 = float(c23._get_potential_energy(g0))
    calc_cases["w128_model23_energy"] = dict(frame=0, model_file="model_2and3.json", energy=e128)
    calc_cases["model23_coefficients"] = m23.coefficients.tolist()
    calc_cases["model_unary_coefficients"] = mu.coefficients.tolist()
    calc_cases["model_binary_coefficients"] = mb.coefficients.tolist()
    json.dump(calc_cases, open(os.path.join(HERE, "calculator_cases.json"), "w"))
    print("wrote calculator_cases.json")

    # ---------------------------------------------------------------- fit (normal equations)
    fit = {}
    b = rb.BSplineBasis(rc.ChemicalSystem(['W'], 3), **w_notebook_kwargs(3))
    nf = int(np.sum(b.partition_sizes))
    r2 = np.random.default_rng(7)
    x_e, x_f = r2.random((40, nf)), r2.random((900, nf)) - 0.5
    c_true = r2.normal(0, 1, nf)
    c_true[b.col_idx] = 0
    y_e, y_f = x_e @ c_true + r2.normal(0, 1e-3, 40), x_f @ c_true + r2.normal(0, 1e-3, 900)
    reg = b.get_regularization_matrix(ridge_map={}, curvature_map={}, ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8,
                                      curvature_2b=1e-8, curvature_3b=0.0)
    m = rl.WeightedLinearModel(b, regularizer=reg)
    m.fit(x_e, y_e, x_f, y_f, weight=0.3, batch_size=250)
    g_e, o_e = rl.batched_moore_penrose(*rl.freeze_columns(x_e, y_e, m.mask, m.frozen_c, m.col_idx))
    g_f, o_f = rl.batched_moore_penrose(*rl.freeze_columns(x_f, y_f, m.mask, m.frozen_c, m.col_idx))
    we, wf = rl.calc_E_F_weights(len(y_e), len(y_f), np.std(y_e), np.std(y_f))
    fit.update(x_e=x_e, y_e=y_e, x_f=x_f, y_f=y_f, regularizer=reg, coefficients=m.coefficients,
               data_coverage=m.data_coverage, gram_e=g_e, ord_e=o_e, gram_f=g_f, ord_f=o_f,
               weights=np.array([we, wf]), kappa=np.array([0.3]), predict_e=m.predict(x_e))
    m2 = rl.WeightedLinearModel(b, regularizer=reg)
    m2.fit(x_e, y_e)
    fit["coefficients_energy_only"] = m2.coefficients
    np.savez_compressed(os.path.join(HERE, "fit_case.npz"), meta=json.dumps(
        dict(element_list=['W'], degree=3, basis_kwargs=basis_kwargs_json(w_notebook_kwargs(3)))), **fit)
    print("wrote fit_case")


if __name__ == "__main__":
    main()
