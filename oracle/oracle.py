"""
ctypes front-end of the CPU restatement (``oracle/uf3_oracle.c``) plus the NumPy
restatement of the normal-equation fit.  TEST INFRASTRUCTURE ONLY: imported by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg.

The basis description is read from any object exposing the reference's
``BSplineBasis`` attribute names (``knots_map``, ``symmetry``, ``template_mask``,
``flat_weights``, ``leading_trim`` ...), so the same wrapper runs on the
reference's own class (golden capture) and on ``uf3_amd``'s.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_SYMBOLS = ("X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga "
            "Ge As Se Br Kr Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd "
            "Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac "
            "Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts "
            "Og").split()
_Z = {s: z for z, s in enumerate(_SYMBOLS)}


class _Spec(C.Structure):
    _fields_ = [("n_species", C.c_int), ("species_z", C.c_void_p),
                ("n_pairs", C.c_int), ("pair_z", C.c_void_p), ("pair_nk", C.c_void_p),
                ("pair_knots", C.c_void_p), ("pair_rmin", C.c_void_p), ("pair_rmax", C.c_void_p),
                ("pair_col", C.c_void_p),
                ("lead2", C.c_int), ("trail2", C.c_int), ("lead3", C.c_int), ("trail3", C.c_int),
                ("n_trios", C.c_int), ("trio_z", C.c_void_p), ("trio_nk", C.c_void_p),
                ("trio_knots", C.c_void_p), ("trio_sym", C.c_void_p), ("trio_col", C.c_void_p),
                ("trio_ncol", C.c_void_p), ("trio_mask", C.c_void_p), ("trio_w", C.c_void_p),
                ("n_feat", C.c_int), ("r_cut", C.c_double)]


class _Frame(C.Structure):
    _fields_ = [("n_atoms", C.c_int), ("pos", C.c_void_p), ("z", C.c_void_p),
                ("cell", C.c_void_p), ("pbc", C.c_void_p)]


def build(force=False):
    """Compile libuf3oracle.so next to the source (gcc, seconds)."""
    so = os.path.join(_HERE, "libuf3oracle.so")
    src = os.path.join(_HERE, "uf3_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libuf3oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.uf3o_featurize.restype = C.c_int
        _LIB.uf3o_eval.restype = C.c_int
        _LIB.uf3o_supercell.restype = C.c_int64
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleBasis:
    """Flat C view of a BSplineBasis-like object (keeps the arrays alive)."""

    def __init__(self, basis):
        self.basis = basis
        cs = basis.chemical_system
        els = list(cs.element_list)
        self.species_z = np.array([_Z[e] for e in els], dtype=np.int32)
        sizes, offsets = basis.get_interaction_partitions()
        pairs = list(cs.interactions_map[2])
        self.pair_z = np.array([[_Z[a], _Z[b]] for a, b in pairs], dtype=np.int32).reshape(-1, 2)
        pk = [np.asarray(basis.knots_map[p], dtype=np.float64) for p in pairs]
        self.pair_nk = np.array([len(k) for k in pk], dtype=np.int32)
        self.pair_knots = np.concatenate(pk) if pk else np.zeros(0)
        self.pair_rmin = np.array([float(basis.r_min_map[p]) for p in pairs])
        self.pair_rmax = np.array([float(basis.r_max_map[p]) for p in pairs])
        self.pair_col = np.array([int(offsets[p]) for p in pairs], dtype=np.int32)
        trios = list(cs.interactions_map.get(3, [])) if cs.degree > 2 else []
        self.trios = trios
        self.pairs = pairs
        self.trio_z = np.array([[_Z[a], _Z[b], _Z[c]] for a, b, c in trios],
                               dtype=np.int32).reshape(-1, 3)
        tk = [[np.asarray(k, dtype=np.float64) for k in basis.knots_map[t]] for t in trios]
        self.trio_nk = np.array([[len(k) for k in ks] for ks in tk], dtype=np.int32).reshape(-1, 3)
        self.trio_knots = (np.concatenate([k for ks in tk for k in ks]) if trios else np.zeros(0))
        self.trio_sym = np.array([basis.symmetry[t] for t in trios], dtype=np.int32)
        self.trio_col = np.array([int(offsets[t]) for t in trios], dtype=np.int32)
        self.trio_ncol = np.array([int(sizes[t]) for t in trios], dtype=np.int32)
        self.trio_mask = (np.concatenate([np.asarray(basis.template_mask[t], dtype=np.int64)
                                          for t in trios]) if trios else np.zeros(0, np.int64))
        self.trio_w = (np.concatenate([np.asarray(basis.flat_weights[t], dtype=np.float64)
                                       for t in trios]) if trios else np.zeros(0))
        self.n_feat = int(np.sum(basis.get_feature_partition_sizes()))
        self.grid_shapes = [tuple(int(n) - 4 for n in row) for row in self.trio_nk]
        s = _Spec()
        s.n_species = len(els)
        s.species_z = _ptr(self.species_z)
        s.n_pairs = len(pairs)
        s.pair_z, s.pair_nk, s.pair_knots = _ptr(self.pair_z), _ptr(self.pair_nk), _ptr(self.pair_knots)
        s.pair_rmin, s.pair_rmax, s.pair_col = _ptr(self.pair_rmin), _ptr(self.pair_rmax), _ptr(self.pair_col)
        s.lead2, s.trail2 = int(basis.leading_trim[2]), int(basis.trailing_trim[2])
        s.lead3, s.trail3 = int(basis.leading_trim.get(3, 0)), int(basis.trailing_trim.get(3, 0))
        s.n_trios = len(trios)
        s.trio_z, s.trio_nk, s.trio_knots = _ptr(self.trio_z), _ptr(self.trio_nk), _ptr(self.trio_knots)
        s.trio_sym, s.trio_col, s.trio_ncol = _ptr(self.trio_sym), _ptr(self.trio_col), _ptr(self.trio_ncol)
        s.trio_mask, s.trio_w = _ptr(self.trio_mask), _ptr(self.trio_w)
        s.n_feat = self.n_feat
        s.r_cut = float(basis.r_cut)
        self.spec = s


class _FrameView:
    def __init__(self, atoms):
        self.pos = np.ascontiguousarray(atoms.get_positions(), dtype=np.float64)
        self.z = np.ascontiguousarray(atoms.get_atomic_numbers(), dtype=np.int32)
        cell = atoms.get_cell()
        self.cell = np.ascontiguousarray(np.array(cell, dtype=np.float64).reshape(3, 3))
        pbc = np.zeros(3, dtype=np.int32)
        pbc[:] = np.asarray(atoms.get_pbc() if hasattr(atoms, "get_pbc") else atoms.pbc)
        self.pbc = pbc
        f = _Frame()
        f.n_atoms = len(self.z)
        f.pos, f.z, f.cell, f.pbc = _ptr(self.pos), _ptr(self.z), _ptr(self.cell), _ptr(self.pbc)
        self.c = f


def featurize(ob, atoms, energy=True, forces=True, indices=False):
    """
    Feature rows of one frame, reference column order without the ``y`` column.

    Returns dict(xe [F] | None, xf [N,3,F] | None) and, with ``indices=True``,
    ``pairs`` {pair: (n,2) int64 (i, j)} in np.where order, ``n3`` (n,2) int64
    (identify_ij, square=False) and ``supercell`` info.
    """
    fv = _FrameView(atoms)
    n = fv.c.n_atoms
    F = ob.n_feat
    xe = np.zeros(F) if energy else None
    xf = np.zeros((n, 3, F)) if forces else None
    out = {}
    pair_cnt = pair_ij = n3_cnt = n3_ij = sc_info = None
    cap2 = cap3 = 0
    if indices:
        sc_info = np.zeros(8, dtype=np.int64)
        pair_cnt = np.zeros(max(1, ob.spec.n_pairs), dtype=np.int64)
        n3_cnt = np.zeros(1, dtype=np.int64)
        rc = lib().uf3o_featurize(C.byref(ob.spec), C.byref(fv.c), None, None,
                                  _ptr(pair_cnt), None, C.c_int64(0),
                                  _ptr(n3_cnt), None, C.c_int64(0), _ptr(sc_info))
        if rc:
            raise RuntimeError(f"uf3o_featurize rc={rc}")
        cap2 = int(pair_cnt.max()) if len(pair_cnt) else 0
        cap3 = int(n3_cnt[0])
        pair_ij = np.zeros((max(1, ob.spec.n_pairs), max(1, cap2), 2), dtype=np.int64)
        n3_ij = np.zeros((max(1, cap3), 2), dtype=np.int64)
    rc = lib().uf3o_featurize(C.byref(ob.spec), C.byref(fv.c), _ptr(xe), _ptr(xf),
                              _ptr(pair_cnt), _ptr(pair_ij), C.c_int64(cap2),
                              _ptr(n3_cnt), _ptr(n3_ij), C.c_int64(cap3), _ptr(sc_info))
    if rc:
        raise RuntimeError(f"uf3o_featurize rc={rc}")
    out["xe"], out["xf"] = xe, xf
    if indices:
        out["pairs"] = {p: pair_ij[k, :pair_cnt[k]].copy() for k, p in enumerate(ob.pairs)}
        out["n3"] = n3_ij[:cap3].copy()
        out["supercell"] = dict(n_img=int(sc_info[0]), m=int(sc_info[1]),
                                factors=sc_info[2:5].tolist(), counts=sc_info[5:8].tolist())
    return out


def supercell(atoms, r_cut):
    fv = _FrameView(atoms)
    m = lib().uf3o_supercell(C.byref(fv.c), C.c_double(r_cut), None, None, None)
    pos = np.zeros((m, 3))
    z = np.zeros(m, dtype=np.int32)
    n_img = m // max(1, fv.c.n_atoms)
    shift = np.zeros((max(1, n_img), 3), dtype=np.int32)
    lib().uf3o_supercell(C.byref(fv.c), C.c_double(r_cut), _ptr(pos), _ptr(z), _ptr(shift))
    return pos, z, shift


def split_coefficients(ob, coefficients):
    """Flat model coefficients -> (c1, c2 concatenated, c3 concatenated full grids)."""
    basis = ob.basis
    coefficients = np.asarray(coefficients, dtype=np.float64)
    sizes, offsets = basis.get_interaction_partitions()
    n_el = len(basis.chemical_system.element_list)
    c1 = np.ascontiguousarray(coefficients[:n_el])
    c2 = [coefficients[offsets[p]:offsets[p] + sizes[p]] for p in ob.pairs]
    c3 = [basis.decompress_3B(coefficients[offsets[t]:offsets[t] + sizes[t]], t).ravel()
          for t in ob.trios]
    return (c1, np.ascontiguousarray(np.concatenate(c2)) if c2 else np.zeros(0),
            np.ascontiguousarray(np.concatenate(c3)) if c3 else np.zeros(0))


def evaluate(ob, atoms, coefficients, forces=True):
    """Energy (and forces) of a model, calculator.py:156-343 restated."""
    c1, c2, c3 = split_coefficients(ob, coefficients)
    fv = _FrameView(atoms)
    e = C.c_double(0.0)
    f = np.zeros((fv.c.n_atoms, 3)) if forces else None
    rc = lib().uf3o_eval(C.byref(ob.spec), C.byref(fv.c), _ptr(c1), _ptr(c2), _ptr(c3),
                         C.byref(e), _ptr(f))
    if rc:
        raise RuntimeError(f"uf3o_eval rc={rc}")
    return e.value, f


# ---------------------------------------------------------------------------
# NumPy restatement of the weighted normal-equation fit
# (uf3/regression/least_squares.py:274-353, 248-272, 716-771, 817-890, 1147-1169)
# ---------------------------------------------------------------------------
def fit(basis, regularizer, x_e, y_e, x_f=None, y_f=None, weight=0.5):
    n_feats = int(np.sum(basis.get_feature_partition_sizes()))
    col_idx, frozen_c = np.asarray(basis.col_idx, dtype=int), np.asarray(basis.frozen_c, float)
    mask = np.setdiff1d(np.arange(n_feats), col_idx)

    def freeze(x, y):
        return x[:, mask], y - x[:, col_idx] @ frozen_c

    x_e, y_e0 = np.asarray(x_e, float), np.asarray(y_e, float)
    xe, ye = freeze(x_e, y_e0)
    g, o = xe.T @ xe, xe.T @ ye
    parts = dict(gram_e=g.copy(), ord_e=o.copy())
    if x_f is not None:
        x_f, y_f0 = np.asarray(x_f, float), np.asarray(y_f, float)
        se, sf = np.std(y_e0), np.std(y_f0)
        if se == 0:
            we, wf = 1.0, 1 / np.sqrt(len(y_f0))
        else:
            we, wf = 1 / np.sqrt(len(y_e0)) / se, 1 / np.sqrt(len(y_f0)) / sf
        xf, yf = freeze(x_f, y_f0)
        gf, of = xf.T @ xf, xf.T @ yf
        parts.update(gram_f=gf.copy(), ord_f=of.copy(), energy_weight=we, force_weight=wf)
        g = weight * we ** 2 * g + (1 - weight) * wf ** 2 * gf
        o = weight * we ** 2 * o + (1 - weight) * wf ** 2 * of
    cov = np.zeros(n_feats, dtype=bool)
    cov[mask] = np.sum(g, axis=0) != 0
    r = np.asarray(regularizer, float)[:, mask]
    sol = np.linalg.solve(g + r.T @ r, o)
    full = np.zeros(n_feats)
    full[mask] = sol
    full[col_idx] = frozen_c
    parts.update(gram=g, ordinate=o, coefficients=full, data_coverage=cov)
    return parts
