"""Shared helpers: load golden cases and rebuild the basis / atoms they describe."""
import json
import os

import numpy as np

from uf3_amd.data import composition
from uf3_amd.data.atoms import Atoms
from uf3_amd.representation import bspline

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dekey(k):
    return tuple(k.split("-")) if "-" in k else k


def decode_basis_kwargs(kw):
    out = {}
    for name, v in kw.items():
        if isinstance(v, dict):
            if name in ("leading_trim", "trailing_trim"):
                out[name] = {int(a): b for a, b in v.items()}
            else:
                out[name] = {_dekey(a): b for a, b in v.items()}
        else:
            out[name] = v
    return out


def basis_from_meta(meta):
    cs = composition.ChemicalSystem(meta["element_list"], meta["degree"])
    return bspline.BSplineBasis(cs, **decode_basis_kwargs(meta["basis_kwargs"]))


def load_case(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(d["meta"]))
    atoms = None
    if "positions" in d:
        atoms = Atoms(numbers=d["numbers"], positions=d["positions"], cell=d["cell"], pbc=d["pbc"])
    return d, meta, atoms


FEATURE_CASES = ["case_steel", "case_h2o", "case_h2o_lead0", "case_ch4", "case_ch4_lead0",
                 "case_w128_2body", "case_w16", "case_w16_lead0", "case_w54", "case_nexe32",
                 "case_nexe32_lead0", "case_ternary24_slab", "case_w16_sym1", "case_w16_sym3"]


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max())) if a.size else 0.0


def worst_elementwise(a, b, rtol=1e-9, floor=1e-12):
    """north_star's "within 1e-6 relative", literally: max over entries of |a - b| / (rtol |b| + floor max|b|); <= 1 passes.
    Every entry is held to ``rtol`` of ITS OWN magnitude; ``floor`` (relative to the largest entry) only covers entries that
    are sums of cancelling terms around zero.  (``rel_err`` above is the max-norm and lets small columns hide behind large
    ones.)"""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if not a.size:
        return 0.0
    return float((np.abs(a - b) / (rtol * np.abs(b) + floor * max(np.abs(b).max(), 1e-300))).max())
