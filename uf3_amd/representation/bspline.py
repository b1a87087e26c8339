"""
Cubic B-spline basis definition for the UF3 hot path (host side).

``BSplineBasis`` turns per-interaction (r_min, r_max, resolution) settings or
explicit knot sequences into everything the device kernels and the solver need:
knot vectors, the L x M x N template of each three-body grid with its symmetry
class, the compressed column layout (``template_mask`` / ``flat_weights``),
column offsets, frozen columns, the regulariser, and ``r_cut``.

Public surface follows the reference's ``uf3/representation/bspline.py``
(class :20-719, ``find_symmetry_3B`` :723-763, knot generators :1011-1124,
``find_spline_indices`` :950-974, trims :1162-1185) and the symmetry template of
``uf3/representation/angles.py:677-735``.  Own implementation: everything here
is integer / rational bookkeeping plus ``numpy.linspace`` knots, so equality
with the reference is exact (checked against captured fixtures in
``tests/test_basis_host.py``).

The only floating-point evaluation in this module, ``basis_values`` (de Boor-Cox
on the global knot vector), is a host convenience for ``fit_spline_1d`` /
``basis_functions``; feature rows come from the HIP library.
"""
import itertools
import os
import re
import warnings

import numpy as np

from uf3_amd.data import composition
from uf3_amd.regression import regularize
from uf3_amd.util import json_io

_NUMERIC = (float, np.floating, int, np.integer)


# --------------------------------------------------------------------------
# knots
# --------------------------------------------------------------------------
def knot_sequence_from_points(knot_points):
    """Repeat both end points three extra times (4-fold end knots)."""
    p = np.asarray(knot_points, dtype=float)
    return np.concatenate([np.repeat(p[0], 3), p, np.repeat(p[-1], 3)])


def get_knot_subintervals(knots):
    return [knots[i:i + 5] for i in range(len(knots) - 4)]


def generate_uniform_knots(r_min, r_max, n_intervals, sequence=True, offset=3):
    if r_min is None:
        r_min = -offset * (r_max - 0.0) / (n_intervals - offset)
    knots = np.linspace(r_min, r_max, n_intervals + 1)
    if sequence:
        knots = knot_sequence_from_points(knots)
    return np.round(knots, 10)


def _need_rmin(r_min):
    if r_min is None:
        raise ValueError("Automatic lower-bound is WIP for this knot spacing scheme.")


def generate_inv_knots(r_min, r_max, n_intervals, sequence=True):
    _need_rmin(r_min)
    knots = np.linspace(1 / r_min, 1 / r_max, n_intervals + 1) ** -1
    return knot_sequence_from_points(knots) if sequence else knots


def generate_geometric_knots(r_min, r_max, n_intervals, sequence=True):
    _need_rmin(r_min)
    knots = np.geomspace(r_min, r_max, n_intervals + 1)
    return knot_sequence_from_points(knots) if sequence else knots


def generate_lammps_knots(r_min, r_max, n_intervals, sequence=True):
    _need_rmin(r_min)
    knots = np.linspace(r_min ** 2, r_max ** 2, n_intervals + 1) ** 0.5
    return knot_sequence_from_points(knots) if sequence else knots


_SPACERS = dict(lammps=generate_lammps_knots, linear=generate_uniform_knots,
                geometric=generate_geometric_knots, inverse=generate_inv_knots)


def get_knot_spacer(knot_strategy):
    if knot_strategy not in _SPACERS:
        raise ValueError('Invalid value of knot_strategy:', knot_strategy)
    return _SPACERS[knot_strategy]


# --------------------------------------------------------------------------
# host-side evaluation of the four non-zero cubic B-splines (convenience only)
# --------------------------------------------------------------------------
def find_interval(knots, points):
    """i such that t[i] < x <= t[i+1] (``searchsorted(..., 'left') - 1``)."""
    return np.searchsorted(knots, points, side='left') - 1


def basis_values(knots, points):
    """
    Values and first derivatives of the (up to) four non-zero cubic B-splines
    at each point, by the de Boor-Cox triangle on the global knot vector.

    Returns (first, values[n,4], derivs[n,4]); ``first`` = index of the first
    basis function (interval - 3).  Points outside (t[0], t[-1]] get zeros.
    """
    t = np.asarray(knots, dtype=float)
    x = np.atleast_1d(np.asarray(points, dtype=float))
    i = find_interval(t, x)
    ok = (i >= 3) & (i <= len(t) - 5) & (x <= t[-1])
    ii = np.where(ok, i, int(np.argmax(np.diff(t) > 0)))  # any non-degenerate interval for masked points
    n = np.zeros((4, len(x)))
    n[0] = 1.0
    left = np.zeros((4, len(x)))
    right = np.zeros((4, len(x)))
    n2 = None
    for j in range(1, 4):
        left[j] = x - t[ii + 1 - j]
        right[j] = t[ii + j] - x
        saved = np.zeros(len(x))
        for r in range(j):
            temp = n[r] / (right[r + 1] + left[j - r])
            n[r] = saved + right[r + 1] * temp
            saved = left[j - r] * temp
        n[j] = saved
        if j == 2:
            n2 = n[:3].copy()
    d = np.zeros((4, len(x)))
    for a in range(4):
        if a >= 1:
            d[a] += 3.0 * n2[a - 1] / (t[ii + a] - t[ii + a - 3])
        if a <= 2:
            d[a] -= 3.0 * n2[a] / (t[ii + a + 1] - t[ii + a - 2])
    n[:, ~ok] = 0.0
    d[:, ~ok] = 0.0
    return i - 3, n.T, d.T


class BasisFunction:
    """One cubic B-spline basis element on 5 knots; callable like scipy's
    ``BSpline.basis_element(..., extrapolate=False)`` with NaN replaced by 0
    (Cox-de Boor recursion, 0/0 := 0, half-open knot intervals: like scipy, the element
    evaluates to 0 at its own last knot, also when that knot is the 4-fold end knot)."""

    def __init__(self, knots5):
        self.t = np.asarray(knots5, dtype=float)

    def _b(self, s, k, x):
        t = self.t
        if k == 0:
            return ((t[s] <= x) & (x < t[s + 1])).astype(float)
        out = np.zeros_like(x)
        d1, d2 = t[s + k] - t[s], t[s + k + 1] - t[s + 1]
        if d1 > 0:
            out += (x - t[s]) / d1 * self._b(s, k - 1, x)
        if d2 > 0:
            out += (t[s + k + 1] - x) / d2 * self._b(s + 1, k - 1, x)
        return out

    def __call__(self, points, nu=0):
        x = np.atleast_1d(np.asarray(points, dtype=float))
        t = self.t
        if nu == 0:
            out = self._b(0, 3, x)
        elif nu == 1:
            out = np.zeros_like(x)
            if t[3] > t[0]:
                out += 3.0 / (t[3] - t[0]) * self._b(0, 2, x)
            if t[4] > t[1]:
                out -= 3.0 / (t[4] - t[1]) * self._b(1, 2, x)
        elif nu == 2:
            out = self._d(0, 3, x, 2)
        else:
            raise NotImplementedError("nu in (0, 1, 2)")
        out[(x < t[0]) | (x > t[4])] = 0.0
        return out

    def _d(self, s, k, x, nu):
        """nu-th derivative of the degree-k element starting at knot s: d/dx B_{s,k} = k (B_{s,k-1} / (t_{s+k} - t_s) -
        B_{s+1,k-1} / (t_{s+k+1} - t_{s+1})), terms over a zero knot span dropped like the recursion's 0/0."""
        if nu == 0:
            return self._b(s, k, x)
        t = self.t
        out = np.zeros_like(x)
        d1, d2 = t[s + k] - t[s], t[s + k + 1] - t[s + 1]
        if d1 > 0:
            out += k / d1 * self._d(s, k - 1, x, nu - 1)
        if d2 > 0:
            out -= k / d2 * self._d(s + 1, k - 1, x, nu - 1)
        return out


def generate_basis_functions(knot_subintervals):
    return [BasisFunction(k) for k in knot_subintervals]


def evaluate_basis_functions(points, basis_functions, nu=0, n_lead=0, n_trail=0, flatten=True):
    """Host restatement of the per-basis sum (bspline.py:810-849); tests / small inputs."""
    n = len(basis_functions)
    vals = [0] * n
    for b in range(n_lead, n - n_trail):
        vals[b] = basis_functions[b](points, nu=nu)
    if not flatten:
        return vals
    return np.array([np.sum(v) for v in vals])


def find_spline_indices(points, knot_sequence):
    """Each point repeated 4x with the indices of its four non-zero splines."""
    points = np.asarray(points)
    idx = np.searchsorted(knot_sequence, points, side='left') - 4
    idx = np.repeat(idx, 4) + np.tile(np.arange(4, dtype=np.int64), len(points))
    return np.repeat(points, 4), idx


def fit_spline_1d(x, y, knot_sequence):
    """Least-squares cubic spline coefficients of samples (x, y) on ``knot_sequence``."""
    from scipy import interpolate
    b_min, b_max = knot_sequence[0], knot_sequence[-1]
    keep = (x > b_min) & (x < b_max)
    x, y = x[keep], y[keep]
    lo, hi = np.argmin(x), np.argmax(x)
    x_min, y_min, x_max, y_max = x[lo], y[lo], x[hi], y[hi]
    uniq = np.unique(knot_sequence)
    for i in range(len(uniq) - 1):
        mid = 0.5 * (uniq[i] + uniq[i + 1])
        if x_min > uniq[i]:
            x, y = np.insert(x, 0, mid), np.insert(y, 0, y_min)
        elif x_max < uniq[i]:
            x, y = np.insert(x, -1, mid), np.insert(y, -1, y_max)
    order = np.argsort(x)
    x, y = x[order], y[order]
    interior = knot_sequence[4:-4] if knot_sequence[0] == knot_sequence[3] else knot_sequence[1:-1]
    lsq = interpolate.LSQUnivariateSpline(x, y, interior, bbox=(b_min, b_max))
    return lsq.get_coeffs()


# --------------------------------------------------------------------------
# three-body symmetry
# --------------------------------------------------------------------------
def find_symmetry_3B(trio, r_min, r_max, resolution):
    """1: legs ij/ik distinguishable; 2: j<->k mirror; 3: full i,j,k permutation."""
    if trio[1] != trio[2]:
        return 1
    legs = list(zip(r_min, r_max, resolution))
    if legs[0] == legs[1] == legs[2]:
        return 3 if trio[0] == trio[1] else 2
    return 2 if legs[0] == legs[1] else 1


def get_symmetry_weights(symmetry, l_space, m_space, n_space, n_lead=0, n_trail=3):
    """
    Weight template of the L x M x N grid (angles.py:677-735): redundant images
    under the symmetry get 0, bins on mirror planes 1/2 (1/6 on the body
    diagonal for symmetry 3), bins whose knot supports cannot close a triangle
    get 0, trimmed planes get 0.
    """
    L, M, N = len(l_space) - 4, len(m_space) - 4, len(n_space) - 4
    li, mi, ni = np.meshgrid(np.arange(L), np.arange(M), np.arange(N), indexing='ij')
    w = np.ones((L, M, N))
    if symmetry == 2:
        w[li > mi] = 0.0
        w[li == mi] = 0.5
    elif symmetry == 3:
        any_eq = (li == mi) | (li == ni) | (mi == ni)
        w[any_eq] = 0.5
        w[(li > mi) | (mi > ni)] = 0.0
        w[(li == mi) & (li == ni)] = 1 / 6
    l_lo, l_hi = np.asarray(l_space)[li], np.asarray(l_space)[li + 4]
    m_lo, m_hi = np.asarray(m_space)[mi], np.asarray(m_space)[mi + 4]
    n_lo, n_hi = np.asarray(n_space)[ni], np.asarray(n_space)[ni + 4]
    w[(l_hi + m_hi <= n_lo) | (l_hi + n_hi <= m_lo) | (m_hi + n_hi <= l_lo)] = 0.0
    for t in range(n_lead):
        w[t, :, :] = 0
        w[:, t, :] = 0
        w[:, :, t] = 0
    for t in range(1, n_trail + 1):
        w[-t, :, :] = 0
        w[:, -t, :] = 0
        w[:, :, -t] = 0
    return w


_PERMS = {1: [(0, 1, 2)],
          2: [(0, 1, 2), (1, 0, 2)],
          3: [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]}


def process_trim_values(user_input, default_trim):
    if user_input is None:
        return dict(default_trim)
    if isinstance(user_input, int):
        return {k: user_input for k in default_trim}
    if isinstance(user_input, dict):
        if not all(isinstance(k, int) for k in user_input):
            raise ValueError("Keys of the trimming values (order of interaction) must be integers.")
        if not all(isinstance(v, int) for v in user_input.values()):
            raise ValueError("Values of the trimming values must be integers.")
        return dict(user_input)
    raise ValueError("Invalid input for trimming values. Must be None, int, or a dict.")


def tuple_consistency_check(map_, interaction_map):
    known = [i for group in interaction_map.values() for i in group]
    for entry in map_:
        if entry not in known:
            warnings.warn(f"{entry} specification unused.")


# --------------------------------------------------------------------------
class BSplineBasis:
    """Basis-set handler; constructor and attributes as in the reference (:26-88)."""

    def __init__(self, chemical_system, r_min_map=None, r_max_map=None, resolution_map=None,
                 knot_strategy='linear', offset_1b=True, leading_trim=None, trailing_trim=None,
                 knots_map=None):
        self.chemical_system = chemical_system
        self.knot_strategy = knot_strategy
        self.offset_1b = offset_1b
        self.leading_trim = process_trim_values(leading_trim, {2: 0, 3: 3})
        self.trailing_trim = process_trim_values(trailing_trim, {2: 3, 3: 3})
        self.r_min_map, self.r_max_map, self.resolution_map = {}, {}, {}
        self.knots_map, self.knot_subintervals = {}, {}
        self._basis_functions = None
        self.symmetry, self.flat_weights, self.template_mask, self.templates = {}, {}, {}, {}
        self.partition_sizes, self.frozen_c, self.col_idx = [], [], []
        self.r_cut = 0.0
        self.update_knots(r_max_map, r_min_map, resolution_map, knots_map)
        self.knot_spacer = get_knot_spacer(self.knot_strategy)
        self.update_basis_functions()

    # -- construction helpers ------------------------------------------------
    @staticmethod
    def from_config(config):
        return BSplineBasis.from_dict(config)

    @staticmethod
    def from_dict(config):
        chemical_system = composition.ChemicalSystem.from_dict(config)
        settings = {}
        if "knots_path" in config and config.get("load_knots"):
            fname = config["knots_path"]
            if os.path.isfile(fname):
                try:
                    settings["knots_map"] = json_io.load_interaction_map(fname)["knots"]
                except (ValueError, KeyError, IOError):
                    settings["knots_map"] = None
        aliases = dict(r_min="r_min_map", r_max="r_max_map", resolution="resolution_map",
                       fit_offsets="offset_1b")
        for key, alias in aliases.items():
            if key in config:
                settings[alias] = config[key]
            if alias in config:
                settings[alias] = config[alias]
        for k in ("r_min_map", "r_max_map", "resolution_map", "knot_strategy", "offset_1b",
                  "leading_trim", "trailing_trim", "knots_map"):
            if k in config:
                settings[k] = config[k]
        for k in ("leading_trim", "trailing_trim"):  # JSON stores the orders as strings
            if isinstance(settings.get(k), dict):
                settings[k] = {int(o): v for o, v in settings[k].items()}
        basis = BSplineBasis(chemical_system, **settings)
        if "knots_path" in config and config.get("dump_knots"):
            json_io.dump_interaction_map(dict(knots=basis.knots_map),
                                         filename=config["knots_path"], write=True)
        return basis

    def as_dict(self):
        return dict(knot_strategy=self.knot_strategy,
                    offset_1b=self.offset_1b,
                    leading_trim={str(k): v for k, v in self.leading_trim.items()},
                    trailing_trim={str(k): v for k, v in self.trailing_trim.items()},
                    knots_map=self.knots_map,
                    **self.chemical_system.as_dict())

    degree = property(lambda self: self.chemical_system.degree)
    element_list = property(lambda self: self.chemical_system.element_list)
    interactions_map = property(lambda self: self.chemical_system.interactions_map)
    interactions = property(lambda self: self.chemical_system.interactions)

    @property
    def n_feats(self):
        return int(np.sum(self.get_feature_partition_sizes()))

    @property
    def basis_functions(self):
        """Per-interaction lists of callable basis elements (host convenience, lazy)."""
        if self._basis_functions is None:
            bf = {}
            for pair in self.interactions_map.get(2, []):
                bf[pair] = generate_basis_functions(self.knot_subintervals[pair])
            if self.degree > 2:
                for trio in self.interactions_map.get(3, []):
                    bf[trio] = [generate_basis_functions(s) for s in self.knot_subintervals[trio]]
            self._basis_functions = bf
        return self._basis_functions

    def __repr__(self):
        sizes = self.get_interaction_partitions()[0]
        lines = ["BSplineBasis:", "    Basis functions:"]
        for n in range(2, self.degree + 1):
            for interaction in self.interactions_map[n]:
                lines.append(" " * 8 + f"{str(interaction)}: {sizes[interaction]:d}")
        lines.append(repr(self.chemical_system))
        return "\n".join(lines)

    def get_cutoff(self):
        """max over pair r_max and, for trios, over the two centre legs' r_max."""
        values = []
        for interaction, r_max in self.r_max_map.items():
            if isinstance(r_max, _NUMERIC):
                values.append(r_max)
            else:
                values.append(max(r_max[:len(interaction) - 1]))
        return max(values)

    def update_knots(self, r_max_map=None, r_min_map=None, resolution_map=None, knots_map=None):
        self._tables_version = getattr(self, "_tables_version", 0) + 1      # device tables of the old knots are stale
        r_min_map = composition.sort_interaction_map(r_min_map or {})
        r_max_map = composition.sort_interaction_map(r_max_map or {})
        resolution_map = composition.sort_interaction_map(resolution_map or {})
        self.r_min_map.update(r_min_map)
        self.r_max_map.update(r_max_map)
        self.resolution_map.update(resolution_map)
        if knots_map is not None:
            self.update_knots_from_dict(composition.sort_interaction_map(knots_map))
        for map_ in (self.r_min_map, self.r_max_map, self.resolution_map):
            tuple_consistency_check(map_, self.interactions_map)
        for pair in self.interactions_map.get(2, []):
            self.r_min_map.setdefault(pair, 1.0)
            self.r_max_map.setdefault(pair, 8.0)
            self.resolution_map.setdefault(pair, 15)
        for trio in self.interactions_map.get(3, []):
            # defaults derive from the *user-supplied* pair entries only (:247-252)
            legs = list(itertools.combinations(trio, 2))
            lo = np.min([r_min_map.get(k, 1.0) for k in legs])
            hi = np.max([r_max_map.get(k, 4.0) for k in legs])
            self.r_min_map.setdefault(trio, [lo, lo, lo])
            self.r_max_map.setdefault(trio, [hi, hi, 2 * hi])
            self.resolution_map.setdefault(trio, [5, 5, 10])
            self.symmetry[trio] = find_symmetry_3B(trio, self.r_min_map[trio],
                                                   self.r_max_map[trio],
                                                   self.resolution_map[trio])
        self.r_cut = self.get_cutoff()

    def update_knots_from_dict(self, knots_map):
        for pair in self.interactions_map.get(2, []):
            if pair not in knots_map:
                warnings.warn(f"{pair} specification unused.")
                continue
            seq = np.array(knots_map[pair])
            self.knots_map[pair] = seq
            self.r_min_map[pair] = seq[0]
            self.r_max_map[pair] = seq[-1]
            self.resolution_map[pair] = len(seq) - 7
        for trio in self.interactions_map.get(3, []):
            if trio not in knots_map:
                warnings.warn(f"{trio} specification unused.")
                continue
            seqs = knots_map[trio]
            if isinstance(seqs[0], _NUMERIC):      # one sequence: full symmetry
                sym, lmn = 3, [seqs, seqs, seqs]
            elif len(seqs) == 2:                    # (l = m, n): one mirror plane
                sym, lmn = 2, [seqs[0], seqs[0], seqs[1]]
            else:
                if len(seqs) > 3:
                    warnings.warn("More than three knot sequences provided "
                                  "for {} interaction.".format(trio), RuntimeWarning)
                sym, lmn = 1, list(seqs[:3])
            lmn = [np.array(s) for s in lmn]
            self.symmetry[trio] = sym
            self.knots_map[trio] = lmn
            self.r_min_map[trio] = [s[0] for s in lmn]
            self.r_max_map[trio] = [s[-1] for s in lmn]
            self.resolution_map[trio] = [len(s) - 7 for s in lmn]

    def update_basis_functions(self):
        self._tables_version = getattr(self, "_tables_version", 0) + 1      # (uf3_amd._lib.device_basis)
        self._basis_functions = None
        for pair in self.interactions_map.get(2, []):
            if pair not in self.knots_map:
                seq = self.knot_spacer(self.r_min_map[pair], self.r_max_map[pair],
                                       self.resolution_map[pair])
                if self.r_min_map[pair] is None:
                    self.r_min_map[pair] = seq[0]
                self.knots_map[pair] = seq
            self.knot_subintervals[pair] = get_knot_subintervals(self.knots_map[pair])
        if self.degree > 2:
            for trio in self.interactions_map.get(3, []):
                if trio not in self.knots_map:
                    self.knots_map[trio] = [
                        self.knot_spacer(self.r_min_map[trio][i], self.r_max_map[trio][i],
                                         self.resolution_map[trio][i]) for i in range(3)]
                self.knot_subintervals[trio] = [get_knot_subintervals(s)
                                                for s in self.knots_map[trio]]
            self.set_flatten_template_3B()
        self.partition_sizes = self.get_feature_partition_sizes()
        self.col_idx, self.frozen_c = self.generate_frozen_indices(
            offset_1b=self.offset_1b, n_lead=self.leading_trim, n_trail=self.trailing_trim)

    # -- regulariser ----------------------------------------------------------
    def get_regularization_matrix(self, ridge_map=None, curvature_map=None, **kwargs):
        """Block matrix R (rows = penalties, columns = features); lambda enters as sqrt."""
        ridge_map = dict(ridge_map or {})
        curvature_map = dict(curvature_map or {})
        for k, v in kwargs.items():  # e.g. ridge_1b=..., curvature_2b=...
            order = int(re.sub('[^0-9]', '', k))
            if k.lower()[0] == 'r':
                ridge_map[order] = float(v)
            elif k.lower()[0] == 'c':
                curvature_map[order] = float(v)
        d = regularize.DEFAULT_REGULARIZER_GRID
        ridge_map = {1: d["ridge_1b"], 2: d["ridge_2b"], 3: d["ridge_3b"], **ridge_map}
        curvature_map = {1: 0.0, 2: d["curve_2b"], 3: d["curve_3b"], **curvature_map}
        matrices = [self.get_regularization_matrix_1b(len(self.element_list), ridge_map[1])]
        for degree in range(2, self.degree + 1):
            for interaction in self.interactions_map[degree]:
                if degree == 2:
                    m = self.get_regularization_matrix_2b(interaction, ridge_map[2],
                                                          curvature_map[2])
                elif degree == 3:
                    m = self.get_regularization_matrix_3b(interaction, ridge_map[3],
                                                          curvature_map[3])
                else:
                    raise ValueError("Four-body terms and beyond are not yet implemented.")
                matrices.append(m)
        return regularize.combine_regularizer_matrices(matrices)

    def get_regularization_matrix_1b(self, n_elements, ridge):
        return regularize.get_ridge_penalty_matrix(n_elements) * np.sqrt(ridge)

    def get_regularization_matrix_2b(self, interaction, ridge, curvature):
        n = self.resolution_map[interaction] + 3
        m = regularize.get_ridge_penalty_matrix(n) * np.sqrt(ridge)
        if curvature > 0:
            m = np.vstack((m, regularize.get_curvature_penalty_matrix_1D(n) * np.sqrt(curvature)))
        return m

    def get_regularization_matrix_3b(self, interaction, ridge, curvature):
        mask = self.template_mask[interaction]
        m = regularize.get_ridge_penalty_matrix(len(mask)) * np.sqrt(ridge)
        if curvature > 0:
            L, M, N = (r + 3 for r in self.resolution_map[interaction])
            lap = regularize.get_curvature_penalty_matrix_3D(L, M, N, flatten=False)
            rows = np.array([self.compress_3B(lap[u], interaction) for u in mask])
            m = np.vstack((m, rows.reshape(len(mask), len(mask)) * np.sqrt(curvature)))
        return m

    # -- column layout ----------------------------------------------------------
    def get_feature_partition_sizes(self):
        sizes = [1] * len(self.element_list)
        for degree in range(2, self.degree + 1):
            for interaction in self.interactions_map[degree]:
                if degree == 2:
                    sizes.append(self.resolution_map[interaction] + 3)
                elif degree == 3:
                    sizes.append(int(np.count_nonzero(self.flat_weights[interaction] > 0)))
                else:
                    raise ValueError("Four-body terms and beyond are not yet implemented.")
        self.partition_sizes = sizes
        return sizes

    def get_interaction_partitions(self):
        sizes = self.get_feature_partition_sizes()
        offsets = np.insert(np.cumsum(sizes), 0, 0)
        names = self.interactions
        return ({names[j]: sizes[j] for j in range(len(names))},
                {names[j]: offsets[j] for j in range(len(names))})

    def get_column_names(self):
        sizes = self.get_interaction_partitions()[0]
        cols = ["y"] + ['n_{}'.format(el) for el in self.element_list]
        for n in range(2, self.degree + 1):
            for interaction in self.interactions_map[n]:
                cols.extend("".join(interaction) + str(i) for i in range(sizes[interaction]))
        return cols

    def generate_frozen_indices(self, offset_1b=True, n_lead=None, n_trail=None, value=0.0):
        """Columns pinned to ``value``: trimmed ends of every pair block (+1-body if asked)."""
        n_lead = self.leading_trim if n_lead is None else n_lead
        n_trail = self.trailing_trim if n_trail is None else n_trail
        sizes, offsets = self.get_interaction_partitions()
        col_idx = []
        for pair in self.interactions_map.get(2, []):
            off, size = offsets[pair], sizes[pair]
            col_idx.extend(off + t for t in range(n_lead[2]))
            col_idx.extend(off + size - t for t in range(1, n_trail[2] + 1))
        for trio in self.interactions_map.get(3, []) if self.degree > 2 else []:
            # trimmed planes are already absent from template_mask, so this set is
            # empty in practice; kept (un-offset, as in the reference :614-628).
            t = np.zeros_like(self.templates[trio])
            for k in range(n_lead[3]):
                t[k, :, :] = t[:, k, :] = t[:, :, k] = 1
            for k in range(1, n_trail[3] + 1):
                t[-k, :, :] = t[:, -k, :] = t[:, :, -k] = 1
            col_idx.extend(np.where(self.compress_3B(t, trio) > 0)[0].tolist())
        frozen_c = [value] * len(col_idx)
        if not offset_1b:
            n_el = len(self.element_list)
            col_idx = list(range(n_el - 1, -1, -1)) + col_idx
            frozen_c = [0] * n_el + frozen_c
        return np.array(col_idx, dtype=int), np.array(frozen_c)

    # -- three-body template / compression --------------------------------------
    def set_flatten_template_3B(self):
        for trio in self.interactions_map[3]:
            l_space, m_space, n_space = self.knots_map[trio]
            template = get_symmetry_weights(self.symmetry[trio], l_space, m_space, n_space,
                                            self.leading_trim[3], self.trailing_trim[3])
            flat = template.flatten()
            mask, = np.where(flat > 0)
            self.template_mask[trio] = mask
            self.flat_weights[trio] = flat[mask]
            self.templates[trio] = template

    def compress_3B(self, grid, interaction, fitting=True):
        """Fold an L x M x N grid over its symmetry images and keep the template bins."""
        sym = self.symmetry[interaction]
        vec = sum(grid.transpose(p) for p in _PERMS[sym]) if sym > 1 else grid
        weight = self.flat_weights[interaction] if fitting else 1.0 / len(_PERMS[sym])
        return vec.flat[self.template_mask[interaction]] * weight

    def decompress_3B(self, vec, interaction):
        L, M, N = (len(s) - 4 for s in self.knots_map[interaction])
        grid = np.zeros((L, M, N))
        grid.flat[self.template_mask[interaction]] = vec * self.flat_weights[interaction]
        sym = self.symmetry[interaction]
        if sym > 1:
            grid = sum(grid.transpose(p) for p in _PERMS[sym])
        return grid

    # -- device view --------------------------------------------------------------
    def column_sources(self, interaction):
        """
        For a trio block: ``lut[L*M*N] -> compressed column (or -1)`` and the weight
        with which a raw (l, m, n) product enters that column, i.e. compress_3B
        applied to a unit grid.  Each raw bin feeds at most one column.
        """
        L, M, N = (len(s) - 4 for s in self.knots_map[interaction])
        size = L * M * N
        mask = self.template_mask[interaction]
        col_of = -np.ones(size, dtype=np.int64)
        col_of[mask] = np.arange(len(mask))
        raw = np.arange(size).reshape(L, M, N)
        lut = -np.ones(size, dtype=np.int32)
        wgt = np.zeros(size)
        for p in _PERMS[self.symmetry[interaction]]:
            # vec[pos] += grid.transpose(p)[pos]  => raw bin `src` lands on position `pos`
            src = raw.transpose(p).reshape(-1)
            pos = np.arange(size)
            hit = col_of[pos] >= 0
            s, c = src[hit], col_of[pos[hit]]
            clash = (lut[s] >= 0) & (lut[s] != c)
            if np.any(clash):
                raise ValueError(f"raw bin feeds two columns for {interaction}")
            lut[s] = c
            np.add.at(wgt, s, self.flat_weights[interaction][c])
        return lut, wgt


def parse_knots_file(filename, chemical_system):
    data = json_io.load_interaction_map(filename)
    out = {}
    for d in range(2, chemical_system.degree + 1):
        for interaction in chemical_system.interactions_map[d]:
            if interaction in data:
                a = data[interaction]
                if np.ptp(a[:4]) == 0 and np.ptp(a[-4:]) == 0 and np.all(np.gradient(a) >= 0):
                    out[interaction] = a
    return out
