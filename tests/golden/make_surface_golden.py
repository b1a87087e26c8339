#!/usr/bin/env python3
"""
Golden capture of the reference's MODULE-LEVEL functions (uf3.representation.distances / angles).
Runs only in the build container, like make_golden.py (same stand-ins, reference at /root/reference):

    python tests/golden/make_surface_golden.py

Inputs are the frames and basis settings of the existing cases (case_steel, case_w16, case_nexe32); the outputs of

    distances.distances_by_interaction (plain and atomic), derivatives_by_interaction, get_distance_matrix,
    compute_direction_cosines, angles.identify_ij (both forms), featurize_energy_3b, featurize_force_3b,
    symmetrize_3B

go to tests/golden/surface_<case>.npz.  The dense drij_dR arrays are mostly zeros; they are kept whole (compressed).
"""
import json
import os
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
from uf3.representation import bspline as rb, distances as rd, angles as ra  # noqa: E402


def dekey(k):
    return tuple(k.split("-")) if "-" in k else k


def decode(kw):
    out = {}
    for name, v in kw.items():
        if isinstance(v, dict):
            out[name] = ({int(a): b for a, b in v.items()} if name in ("leading_trim", "trailing_trim")
                         else {dekey(a): b for a, b in v.items()})
        else:
            out[name] = v
    return out


def capture(case, r_cut_sup=None):
    d = np.load(os.path.join(HERE, case + ".npz"))
    meta = json.loads(str(d["meta"]))
    basis = rb.BSplineBasis(rc.ChemicalSystem(meta["element_list"], meta["degree"]), **decode(meta["basis_kwargs"]))
    geom = ase.Atoms(numbers=d["numbers"], positions=d["positions"], cell=d["cell"], pbc=d["pbc"])
    sup = rg.get_supercell(geom, r_cut=r_cut_sup or basis.r_cut)
    out = dict(r_cut_sup=np.array([r_cut_sup or basis.r_cut]))
    pairs = basis.interactions_map[2]
    dist = rd.distances_by_interaction(geom, pairs, basis.r_min_map, basis.r_max_map, supercell=sup)
    dist_at = rd.distances_by_interaction(geom, pairs, basis.r_min_map, basis.r_max_map, supercell=sup, atomic=True)
    dist_cluster = rd.distances_by_interaction(geom, pairs, basis.r_min_map, basis.r_max_map)
    dmap, drmap = rd.derivatives_by_interaction(geom, pairs, basis.r_cut, basis.r_min_map, basis.r_max_map, supercell=sup)
    for p, pair in enumerate(pairs):
        out[f"dist{p}"] = dist[pair]
        out[f"dist_cluster{p}"] = dist_cluster[pair]
        out[f"dist_atomic_len{p}"] = np.array([len(v) for v in dist_at[pair]])
        out[f"dist_atomic_cat{p}"] = np.concatenate(dist_at[pair]) if len(dist_at[pair]) else np.zeros(0)
        out[f"deriv_dist{p}"] = dmap[pair]
        out[f"deriv_drij{p}"] = drmap[pair]
    dm = rd.get_distance_matrix(geom, sup)
    out["dm_geom_sup"] = dm
    # compute_direction_cosines on an explicit index list (every 7th pair inside 4 A of the first atoms)
    dss = rd.get_distance_matrix(sup, sup)
    iw, jw = np.where((dss > 0) & (dss < 4.0))
    pick = np.arange(0, len(iw), 7)[:400]
    out["dc_i"], out["dc_j"] = iw[pick].astype(np.int64), jw[pick].astype(np.int64)
    out["dc_out"] = rd.compute_direction_cosines(sup.get_positions(), dss, iw[pick], jw[pick], len(geom))
    if basis.degree > 2:
        trios = basis.interactions_map[3]
        ks = [basis.knots_map[t] for t in trios]
        bf = [basis.basis_functions[t] for t in trios]
        hashes = basis.chemical_system.interaction_hashes[3]
        _, iw, jw = ra.identify_ij(geom, ks, sup)
        out["ij_i"], out["ij_j"] = iw.astype(np.int64), jw.astype(np.int64)
        _, m2, iw2, jw2 = ra.identify_ij(geom, ks, sup, square=True)
        out["ij_sq_i"], out["ij_sq_j"] = iw2.astype(np.int64), jw2.astype(np.int64)
        out["ij_sq_matrix_rows"] = m2[:len(geom)]
        n_lead, n_trail = basis.leading_trim[3], basis.trailing_trim[3]
        out["trims3"] = np.array([n_lead, n_trail])
        out["hashes3"] = np.array(hashes, dtype=np.int64)
        eg = ra.featurize_energy_3b(geom, ks, bf, hashes, supercell=sup, n_lead=n_lead, n_trail=n_trail)
        for t, g in enumerate(eg):
            out[f"energy_grid{t}"] = g
        eg0 = ra.featurize_energy_3b(geom, ks, bf, hashes, supercell=sup)       # the function's own defaults: no trims
        for t, g in enumerate(eg0):
            out[f"energy_grid_notrim{t}"] = g
        ec = ra.featurize_energy_3b(geom, ks, bf, hashes, n_lead=n_lead, n_trail=n_trail)     # cluster
        for t, g in enumerate(ec):
            out[f"energy_grid_cluster{t}"] = g
        fg = ra.featurize_force_3b(geom, ks, bf, hashes, supercell=sup, n_lead=n_lead, n_trail=n_trail)
        for t, per_atom in enumerate(fg):
            out[f"force_grid{t}"] = np.array([[np.asarray(c) for c in comps] for comps in per_atom])
        sym = basis.symmetry[trios[0]]
        out["sym0"] = np.array([sym])
        out["symmetrized0"] = ra.symmetrize_3B(eg[0], sym)
        out["symmetrized0_s2"] = ra.symmetrize_3B(eg[0][:min(eg[0].shape[:2]), :min(eg[0].shape[:2])], 2)
    np.savez_compressed(os.path.join(HERE, f"surface_{case}.npz"), meta=json.dumps(meta), **out)
    print("wrote surface_" + case, {k: np.asarray(v).shape for k, v in out.items() if k.startswith(("dist0", "force_grid0"))})


if __name__ == "__main__":
    for case in sys.argv[1:] or ["case_steel", "case_w16", "case_nexe32"]:
        capture(case)
