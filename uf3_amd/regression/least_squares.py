"""
``WeightedLinearModel``: regularised normal equations for UF3 coefficients.

Surface follows the reference's ``uf3/regression/least_squares.py``
(``WeightedLinearModel`` :144-621, gram helpers :716-760, frozen columns :817-890,
weights :1147-1169).  The heavy part -- ``G = X^T X`` and ``o = X^T y`` over all
energy and force rows -- runs on the MI355X matrix cores (``uf3_gram`` in
``libuf3hip.so``; fp64 MFMA).  The F' x F' solve stays on the host (LAPACK dgesv via
``numpy.linalg.solve``, exactly as the reference, :763-771); it is O(F'^3) once per fit.

``gram_pieces`` / ``fit_from_pieces`` expose the per-shard pieces
``{G_e, G_f, o_e, o_f}`` + target moments, i.e. what a multi-GPU fit sum-reduces
(``uf3_amd.parallel``).  No CPU fallback for the Gram accumulation.
"""
import warnings

import numpy as np

from uf3_amd import _lib
from uf3_amd.data import composition
from uf3_amd.representation import bspline
from uf3_amd.util import json_io


class VarianceRecorder:
    """Streaming mean / population std (least_squares.py:19-67)."""

    def __init__(self, mean=0, std=0, n=0):
        self.mean, self.std, self.n = mean, std, int(n)

    def update(self, batch):
        batch = np.asarray(batch, dtype=float)
        if self.n == 0:
            self.mean, self.std, self.n = np.mean(batch, axis=0), np.std(batch, axis=0), len(batch)
        else:
            m, n = float(self.n), len(batch)
            b_std, b_mean = np.std(batch, axis=0), np.mean(batch, axis=0)
            var = (m / (m + n) * self.std ** 2 + n / (m + n) * b_std ** 2
                   + m * n / (m + n) ** 2 * (self.mean - b_mean) ** 2)
            self.std = np.sqrt(var)
            self.mean = m / (m + n) * self.mean + n / (m + n) * b_mean
            self.n += n
        return self.mean, self.std, self.n

    def update_with_components(self, df, keys=None):
        """One update from force components spread over columns (default fx, fy, fz): every row contributes the
        concatenation of its cells (scalars or per-atom lists); rows with a missing cell are left out (:55-67)."""
        columns = ["fx", "fy", "fz"] if keys is None else list(keys)
        values = []
        for cells in df[columns].itertuples(index=False, name=None):
            if any(cell is np.nan for cell in cells):
                continue
            values.extend(np.ravel(np.concatenate([np.atleast_1d(cell) for cell in cells])).tolist())
        self.update(values)
        return self.mean, self.std, self.n


def moments(y):
    """(n, sum, sum of squares): the additive form of VarianceRecorder, for reductions."""
    y = np.asarray(y, dtype=float)
    return np.array([len(y), np.sum(y), np.sum(y * y)])


def std_from_moments(m):
    n, s, ss = m
    if n == 0:
        return 0.0
    mean = s / n
    return float(np.sqrt(max(ss / n - mean * mean, 0.0)))


def gram_device(x, y, device=None):
    """(X^T X, X^T y) of a row block on the GPU (fp64 MFMA)."""
    import ctypes as C
    ctx = _lib.get_context(device)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    n_rows, n_feat = x.shape
    gram = np.empty((n_feat, n_feat))
    ordinate = np.empty(n_feat)
    ctx.check(ctx.lib.uf3_gram(ctx.handle, _lib._p(x), _lib._p(y), n_rows, n_feat, n_feat, 0,
                               _lib._p(gram), _lib._p(ordinate)))
    return gram, ordinate


def moore_penrose_components(x, y):
    return gram_device(x, y)


def batched_moore_penrose(x, y, batch_size=2500):
    # one device launch covers any number of rows; batch_size kept for signature parity
    return gram_device(x, y)


def lu_factorization(a, b):
    return np.linalg.solve(a, b)


def get_freezing_mask(n_feats, col_idx):
    return np.setdiff1d(np.arange(n_feats), col_idx)


def freeze_columns(x, y, mask, frozen_c, col_idx):
    x = np.asarray(x)
    y = np.subtract(y, np.dot(x[:, col_idx], frozen_c))
    return x[:, mask], y


def freeze_regularizer(regularizer, mask):
    return regularizer[:, mask]


def revert_frozen_coefficients(solution, n_coeff, mask, frozen_c, frozen_idx):
    full = np.zeros(n_coeff)
    full[mask] = solution
    full[frozen_idx] = frozen_c
    return full


def calc_E_F_weights(n_e, n_f, std_e, std_f):
    if std_e == 0:
        return 1.0, 1 / np.sqrt(n_f)
    return 1 / np.sqrt(n_e) / std_e, 1 / np.sqrt(n_f) / std_f


def rmse_metric(predicted, actual):
    return np.sqrt(np.mean(np.subtract(predicted, actual) ** 2))


def mae_metric(predicted, actual):
    return np.mean(np.abs(np.subtract(predicted, actual)))


def arrange_coefficients(coefficients, bspline_config):
    parts = np.array_split(coefficients, np.cumsum(bspline_config.partition_sizes)[:-1])
    els = bspline_config.element_list
    out = {el: v[0] for el, v in zip(els, parts[:len(els)])}
    j = len(els)
    for d in range(2, bspline_config.degree + 1):
        for interaction in bspline_config.interactions_map[d]:
            out[interaction] = parts[j]
            j += 1
    return out


class BasicLinearModel:
    device = None      # HIP device of the Gram products (None: UF3_DEVICE / LOCAL_RANK / 0, like the featurizer)

    def __init__(self, regularizer=None):
        self.coefficients = None
        self.regularizer = regularizer

    def fit(self, x, y, ridge_penalty=1e-8):
        gram, ordinate = gram_device(x, y, self.device)
        reg = np.eye(len(gram)) * ridge_penalty if self.regularizer is None else self.regularizer
        self.coefficients = lu_factorization(gram + np.dot(reg.T, reg), ordinate)

    def predict(self, x):
        return np.dot(x, self.coefficients)

    def score(self, x, y, weights=None, normalize=True):
        score = -rmse_metric(y, self.predict(x))
        return score / np.std(y) if normalize else score


class WeightedLinearModel(BasicLinearModel):
    def __init__(self, bspline_config, regularizer=None, data_coverage=None, **params):
        super().__init__(regularizer)
        self.bspline_config = bspline_config
        n_basis = int(np.sum(bspline_config.get_feature_partition_sizes()))
        if data_coverage is not None:
            if len(data_coverage) != n_basis:
                raise ValueError(f"Incorrect data_coverage shape: {len(data_coverage)} != {n_basis}")
            self.data_coverage = data_coverage
        else:
            self.data_coverage = np.zeros(n_basis, dtype=bool)
        if self.regularizer is None:
            self.set_params(**params)

    def set_params(self, **params):
        if "bspline_config" in params:
            self.bspline_config = params["bspline_config"]
        if "regularizer" in params:
            self.regularizer = params["regularizer"]
        elif self.regularizer is None:
            reg = {k: v for k, v in params.items() if isinstance(v, (int, float, np.floating))}
            self.regularizer = self.bspline_config.get_regularization_matrix(**reg)

    @staticmethod
    def from_config(config):
        return WeightedLinearModel.from_dict(config)

    @staticmethod
    def from_dict(config):
        basis = bspline.BSplineBasis.from_dict(config)
        model = WeightedLinearModel(basis, regularizer=config.get("regularizer", None),
                                    data_coverage=config.get("data_coverage", None))
        model.load(solution=config)
        return model

    @staticmethod
    def from_json(filename):
        return WeightedLinearModel.from_dict(json_io.load_interaction_map(filename))

    def as_dict(self):
        solution = arrange_coefficients(self.coefficients, self.bspline_config)
        for trio in self.bspline_config.interactions_map.get(3, []) if self.bspline_config.degree > 2 else []:
            solution[trio] = self.bspline_config.decompress_3B(solution[trio], trio)
        return dict(coefficients=solution, knots=self.bspline_config.knots_map,
                    data_coverage=self.data_coverage, **self.bspline_config.as_dict())

    dump = as_dict

    def to_json(self, filename):
        json_io.dump_interaction_map(self.as_dict(), filename=filename, write=True)

    n_feats = property(lambda self: self.bspline_config.n_feats)
    frozen_c = property(lambda self: self.bspline_config.frozen_c)
    col_idx = property(lambda self: self.bspline_config.col_idx)
    mask = property(lambda self: get_freezing_mask(self.n_feats, self.col_idx))

    def __repr__(self):
        return "\n".join(["WeightedLinearModel:", f"    Fit: {self.coefficients is not None}",
                          repr(self.bspline_config)])

    # -- fitting ----------------------------------------------------------------------
    def fit_with_gram(self, gram, ordinate):
        """Solve (G + R^T R) c = o on the unfrozen columns (least_squares.py:248-272)."""
        coverage = revert_frozen_coefficients(np.sum(gram, axis=0) != 0, self.n_feats, self.mask,
                                              self.frozen_c, self.col_idx)
        self.data_coverage = np.logical_or(self.data_coverage, coverage)
        reg = freeze_regularizer(self.regularizer, self.mask)
        solution = lu_factorization(gram + np.dot(reg.T, reg), ordinate)
        self.coefficients = revert_frozen_coefficients(solution, self.n_feats, self.mask, self.frozen_c,
                                                       self.col_idx)

    def gram_pieces(self, x_e, y_e, x_f=None, y_f=None):
        """Additive pieces of one shard: Gram/ordinate of the frozen system + target moments."""
        x_e, y_e = np.asarray(x_e, dtype=float), np.asarray(y_e, dtype=float)
        xe, ye = freeze_columns(x_e, y_e, self.mask, self.frozen_c, self.col_idx)
        # the energy weight comes from the FROZEN targets, the force weight from the raw ones (least_squares.py:296-304)
        pieces = dict(m_e=moments(ye))
        pieces["gram_e"], pieces["ord_e"] = gram_device(xe, ye, self.device)
        if x_f is not None:
            x_f, y_f = np.asarray(x_f, dtype=float), np.asarray(y_f, dtype=float)
            pieces["m_f"] = moments(y_f)
            xf, yf = freeze_columns(x_f, y_f, self.mask, self.frozen_c, self.col_idx)
            pieces["gram_f"], pieces["ord_f"] = gram_device(xf, yf, self.device)
        return pieces

    def fit_from_pieces(self, pieces, weight=0.5):
        if "gram_f" in pieces:
            w_e, w_f = calc_E_F_weights(pieces["m_e"][0], pieces["m_f"][0], std_from_moments(pieces["m_e"]),
                                        std_from_moments(pieces["m_f"]))
            gram, ordinate = self.combine_weighted_gram(pieces["gram_e"], pieces["gram_f"], pieces["ord_e"],
                                                        pieces["ord_f"], w_e, w_f, weight)
        else:
            gram, ordinate = pieces["gram_e"], pieces["ord_e"]
        self.fit_with_gram(gram, ordinate)

    def fit(self, x_e, y_e, x_f=None, y_f=None, weight=0.5, batch_size=2500):
        """Energies (+ forces) -> coefficients (least_squares.py:274-321)."""
        x_e, y_e = np.asarray(x_e, dtype=float), np.asarray(y_e, dtype=float)
        xe, ye = freeze_columns(x_e, y_e, self.mask, self.frozen_c, self.col_idx)
        gram, ordinate = gram_device(xe, ye, self.device)
        if x_f is not None:
            y_f = np.asarray(y_f, dtype=float)
            # std of the frozen energies, of the raw forces (least_squares.py:296-304)
            w_e, w_f = calc_E_F_weights(len(ye), len(y_f), np.std(ye), np.std(y_f))
            xf, yf = freeze_columns(np.asarray(x_f, dtype=float), y_f, self.mask, self.frozen_c, self.col_idx)
            gram_f, ord_f = gram_device(xf, yf, self.device)
            gram, ordinate = self.combine_weighted_gram(gram, gram_f, ordinate, ord_f, w_e, w_f, weight)
        self.fit_with_gram(gram, ordinate)

    def combine_weighted_gram(self, gram_e, gram_f, ord_e, ord_f, energy_weight, force_weight, weight):
        gram = (weight * energy_weight ** 2 * gram_e) + ((1 - weight) * force_weight ** 2 * gram_f)
        ordinate = (weight * energy_weight ** 2 * ord_e) + ((1 - weight) * force_weight ** 2 * ord_f)
        return gram, ordinate

    def initialize_gram_ordinate(self):
        n = self.n_feats - len(self.col_idx)
        return np.zeros((n, n)), np.zeros((n, n)), np.zeros(n), np.zeros(n)

    # -- feature tables (the from-file workflow, least_squares.py:355-526) ------------------
    def gram_from_df(self, df, keys, e_variance=None, f_variance=None, sample_weights=None, energy_key="energy",
                     batch_size=2500):
        """(gram_e, gram_f, ordinate_e, ordinate_f) of the rows of ``keys`` in one feature table (:435-483); the
        products run on the GPU (``uf3_gram``), ``batch_size`` is accepted for signature parity only."""
        x_e, y_e, x_f, y_f = dataframe_to_tuples(df.loc[keys], n_elements=len(self.bspline_config.element_list),
                                                 energy_key=energy_key, sample_weights=sample_weights)
        x_e, y_e = freeze_columns(x_e, y_e, self.mask, self.frozen_c, self.col_idx)
        x_f, y_f = freeze_columns(x_f, y_f, self.mask, self.frozen_c, self.col_idx)
        if e_variance is not None and f_variance is not None:
            e_variance.update(y_e)
            f_variance.update(y_f)
        gram_e, ord_e = gram_device(x_e, y_e, self.device)
        gram_f, ord_f = gram_device(x_f, y_f, self.device)
        return gram_e, gram_f, ord_e, ord_f

    def fit_from_tables(self, tables, subset, weight=0.5, batch_size=2500, sample_weights=None, energy_key="energy",
                        drop_columns=None):
        """The accumulation loop of ``fit_from_file`` (:386-424) over any iterable of feature tables
        (DataFrames as ``BasisFeaturizer.evaluate`` returns them): Gram pieces summed table by table, energy /
        force weights from the streamed target statistics, one solve."""
        gram_e, gram_f, ord_e, ord_f = self.initialize_gram_ordinate()
        e_variance, f_variance = VarianceRecorder(), VarianceRecorder()
        for df in tables:
            keys = df.index.unique(level=0).intersection(subset)
            if len(keys) == 0:
                continue
            if drop_columns is not None:
                df = df.drop(columns=drop_columns)
            g_e, g_f, o_e, o_f = self.gram_from_df(df, keys, e_variance=e_variance, f_variance=f_variance,
                                                   sample_weights=sample_weights, energy_key=energy_key,
                                                   batch_size=batch_size)
            gram_e += g_e
            gram_f += g_f
            ord_e += o_e
            ord_f += o_f
        w_e, w_f = calc_E_F_weights(e_variance.n, f_variance.n, e_variance.std, f_variance.std)
        gram, ordinate = self.combine_weighted_gram(gram_e, gram_f, ord_e, ord_f, w_e, w_f, weight)
        self.fit_with_gram(gram, ordinate)

    def fit_from_file(self, filename, subset, weight=0.5, batch_size=2500, sample_weights=None, energy_key="energy",
                      progress="bar", drop_columns=None):
        """``fit_from_tables`` over the tables of an HDF5 feature file written by the reference's
        ``batched_to_hdf`` (:355-424).  Reading goes through ``pandas.read_hdf`` and therefore needs PyTables, which
        the build image lacks: this wrapper is not exercised by the tests (the loop it delegates to is)."""
        import os
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        names = hdf_table_names(filename)
        self.fit_from_tables((pd_read_hdf(filename, name) for name in names), subset, weight=weight,
                             batch_size=batch_size, sample_weights=sample_weights, energy_key=energy_key,
                             drop_columns=drop_columns)

    def batched_predict(self, filename=None, keys=None, table_names=None, score=True, drop_columns=None, tables=None):
        """Targets and predictions over the tables of a feature file (:485-526), or over ``tables`` (an iterable
        of DataFrames) when given."""
        y_e, p_e, y_f, p_f = batched_prediction(self, filename, table_names=table_names, subset_keys=keys,
                                                drop_columns=drop_columns, tables=tables,
                                                n_elements=len(self.bspline_config.element_list))
        if score:
            rmse_e, rmse_f = rmse_metric(y_e, p_e), rmse_metric(y_f, p_f)
            print(f"RMSE (energy): {rmse_e:.3F}")
            print(f"RMSE (forces): {rmse_f:.3F}")
            return y_e, p_e, y_f, p_f, rmse_e, rmse_f
        return y_e, p_e, y_f, p_f

    # -- model files ------------------------------------------------------------------
    def fix_repulsion_2b(self, pair, r_target=None, min_curvature=2.0):
        """
        Coefficients of the pair block that no training distance reached (``data_coverage`` False between the leading trim
        and the first covered function) replaced by a second-order Taylor continuation of the fitted pair potential around
        ``r_target`` (default: the centre of the first covered function), its curvature there raised to ``min_curvature``:
        a repulsive wall where the data said nothing (reference ``least_squares.py:623-647``).
        """
        sizes, offsets = self.bspline_config.get_interaction_partitions()
        first, n_basis = offsets[pair], sizes[pair]
        block = np.arange(first, first + n_basis)
        covered = np.asarray(self.data_coverage)[block]
        first_covered = int(np.argmax(covered == True))  # noqa: E712  (element-wise, like the reference)
        if first_covered == 0:
            print(f"Coverage is sufficient; no fix applied to {pair}.")
        to_fix = np.arange(self.bspline_config.leading_trim[2], first_covered)
        knots = self.bspline_config.knots_map[pair]
        centres = knots[2: n_basis + 2]
        if r_target is None:
            r_target = centres[first_covered]
        values = get_spline_taylor_expansion(r_target, centres[to_fix], self.coefficients[block], knots,
                                             min_curvature=min_curvature)
        print(f"{pair} Correction: adjusted {len(to_fix)} coefficients.")
        self.coefficients[block[to_fix]] = values

    def load(self, solution=None, filename=None):
        """Flatten a per-interaction coefficient dict (3-body given as full grids) into ``coefficients``."""
        if filename is not None:
            if solution is not None:
                warnings.warn("Provided solutions ignored; loading file.")
            solution = json_io.load_interaction_map(filename)
        elif solution is None:
            raise ValueError("Neither solution nor filename were provided.")
        if "coefficients" in solution:
            solution = solution["coefficients"]
        elif "solution" in solution:
            warnings.warn("'solution' should be renamed to 'coefficients'")
            solution = solution["solution"]
        solution = dict(solution)
        for key in list(solution):
            if isinstance(key, tuple):
                skey = composition.sort_interaction_symbols(key)
                if skey != key:
                    solution[skey] = solution[key]
        basis = self.bspline_config
        sizes = basis.get_interaction_partitions()[0]
        for pair in basis.interactions_map[2]:
            if pair not in solution:
                warnings.warn(f"{pair} not provided.")
                solution[pair] = np.zeros(sizes[pair])
            if len(solution[pair]) != sizes[pair]:
                raise ValueError(f"Incorrect shape: {pair}, {len(solution[pair])} != {sizes[pair]}")
        for trio in basis.interactions_map.get(3, []) if basis.degree > 2 else []:
            if trio not in solution:
                warnings.warn(f"{trio} not provided.")
            component = np.array(solution[trio])
            if component.ndim > 1:
                solution[trio] = basis.compress_3B(component, trio, fitting=False)
            if len(solution[trio]) != sizes[trio]:
                raise ValueError(f"Incorrect shape: {trio}, {len(solution[trio])} != {sizes[trio]}")
        flat = [[solution[el]] for el in basis.element_list]
        for degree in range(2, basis.degree + 1):
            flat.extend(solution[i] for i in basis.interactions_map[degree])
        if len(flat) != len(basis.partition_sizes):
            raise ValueError("Incorrect interactions: {} provided, {} expected.".format(
                len(flat), len(basis.partition_sizes)))
        flat = np.concatenate([np.atleast_1d(np.asarray(v, dtype=float)) for v in flat])
        if len(flat) != sum(basis.partition_sizes):
            raise ValueError("Incorrect coefficients: {} provided, {} expected.".format(
                len(flat), sum(basis.partition_sizes)))
        self.coefficients = np.array(flat)


def get_spline_taylor_expansion(r_target, r, coefficients, knot_sequence, min_curvature=0.0):
    """Second-order Taylor polynomial, around ``r_target``, of the cubic spline sum_b c_b B_b on ``knot_sequence``, evaluated
    at ``r``; the curvature is raised to ``min_curvature`` when that is not None (reference ``least_squares.py:650-663``,
    there through ``ndsplines``)."""
    from uf3_amd.representation import bspline
    knots = np.asarray(knot_sequence, dtype=float)
    c = np.asarray(coefficients, dtype=float)
    elements = bspline.generate_basis_functions([knots[i:i + 5] for i in range(len(knots) - 4)])
    at = np.atleast_1d(np.asarray(r_target, dtype=float))
    value, slope, curvature = (float(sum(c[b] * elements[b](at, nu=nu)[0] for b in range(len(c)))) for nu in (0, 1, 2))
    if min_curvature is not None:
        curvature = max(curvature, min_curvature)
    step = np.asarray(r, dtype=float) - r_target
    return value + slope * step + 0.5 * curvature * step ** 2


def find_pair_potential_well(coefficients, rounding_factor):
    """Index of the coefficient that looks like the pair potential's well: the minimum -- unless it lies left of the
    maximum and the coefficients before the maximum are flat to ``rounding_factor`` decimals (no well there: the index
    behind the maximum).  Reference ``least_squares.py:1123-1144``."""
    c = np.asarray(coefficients)
    peak, well = int(np.argmax(c)), int(np.argmin(c))
    if well < peak:
        before_peak = np.round(c[:peak], rounding_factor)
        if np.ptp(before_peak) < 10 ** -(rounding_factor - 1):
            well = peak + 1
    return well


def postprocess_coefficients_2b(coefficients, core_hardness=2.0, min_core=2.0, min_slope=0.1, rounding_factor=3,
                                smooth_cutoff=False, in_place=False):
    """
    A repulsive core for a vector of pair coefficients (reference ``least_squares.py:1075-1120``): left of the well, when the
    coefficients only rise towards a maximum (the lower bound lies far below the data and they are nearly zero), they are
    rebuilt from the maximum downwards in r as ``max(core_hardness * |next|, min_slope)``; the first coefficient is at least
    ``min_core``; ``smooth_cutoff`` zeroes the last two.  Works on a copy unless ``in_place``.
    """
    c = coefficients if in_place else np.array(coefficients)
    well = find_pair_potential_well(c, rounding_factor)
    if well > 1:
        left = np.round(c[:well], rounding_factor)
        left = left + np.arange(len(left)) * 10 ** (-2 * rounding_factor)      # (a plateau leans towards the well)
        slope = np.gradient(left)
        peak = int(np.argmax(left))
        if np.all(slope[:peak] >= 0):
            for i in range(peak - 1, -1, -1):
                c[i] = max(np.abs(c[i + 1]) * core_hardness, min_slope)
    if c[0] < min_core:
        c[0] = min_core
    if smooth_cutoff:
        c[-2:] = 0
    return c


def dataframe_to_tuples(df_features, n_elements=None, energy_key='energy', sample_weights=None):
    """
    Split a feature table (``BasisFeaturizer.evaluate`` layout: MultiIndex (name, 'energy' | 'fx_i' ...), target ``y``
    in the first column) into energy and force inputs / targets (reference: least_squares.py:666-713).

    ``n_elements``: the energy rows and targets are divided by the sum of the first ``n_elements`` feature columns
    (the per-element atom counts), i.e. normalised per atom.  ``sample_weights``: {name: weight}, default 1, applied
    to rows and targets.  Returns (x_e, y_e, x_f, y_f).
    """
    keys = np.asarray(df_features.index.get_level_values(-1))
    is_energy = keys == energy_key
    table = df_features.to_numpy()
    y, x = table[:, 0], table[:, 1:]
    x_e, y_e, x_f, y_f = x[is_energy], y[is_energy], x[~is_energy], y[~is_energy]
    if n_elements is not None:
        n_atoms = x_e[:, :n_elements].sum(axis=1)
        x_e, y_e = x_e / n_atoms[:, None], y_e / n_atoms
    if sample_weights is not None:
        w = np.array([sample_weights.get(name, 1.0) for name in df_features.index.get_level_values(0)])
        w_e, w_f = w[is_energy], w[~is_energy]
        x_e, y_e, x_f, y_f = x_e * w_e[:, None], y_e * w_e, x_f * w_f[:, None], y_f * w_f
    return x_e, y_e, x_f, y_f


def subset_prediction(df, model, subset_keys=None, **kwargs):
    """
    Targets and predictions of (a subset of) a feature table (reference: least_squares.py:933-962).
    Returns (y_e, p_e, y_f, p_f); four empty lists when none of ``subset_keys`` is in the table.
    """
    if subset_keys is not None:
        present = df.index.unique(level=0).intersection(subset_keys)
        if len(present) == 0:
            return list(), list(), list(), list()
        df = df.loc[present]
    x_e, y_e, x_f, y_f = dataframe_to_tuples(df, **kwargs)
    return y_e, model.predict(x_e), y_f, model.predict(x_f)



def hdf_table_names(filename):
    """Sorted names of the top-level tables of an HDF5 feature file (io.py:943-956); needs PyTables."""
    import pandas as pd
    with pd.HDFStore(filename, mode="r") as store:
        return sorted(key.lstrip("/").split("/")[0] for key in store.keys())


def pd_read_hdf(filename, table_name):
    import pandas as pd
    return pd.read_hdf(filename, table_name)


def batched_prediction(model, filename=None, table_names=None, subset_keys=None, drop_columns=None, tables=None,
                       **kwargs):
    """``subset_prediction`` table by table, concatenated (least_squares.py:965-1014); ``tables``: DataFrames
    instead of a file."""
    if tables is None:
        if table_names is None:
            table_names = hdf_table_names(filename)
        tables = (pd_read_hdf(filename, name) for name in table_names)
    parts = ([], [], [], [])
    for df in tables:
        if drop_columns is not None:
            df = df.drop(columns=drop_columns)
        for dst, piece in zip(parts, subset_prediction(df, model, subset_keys=subset_keys, **kwargs)):
            dst.append(piece)
    return tuple(np.concatenate(p) for p in parts)


def linear_least_squares(x, y):
    """Unregularised solve through the normal equations (least_squares.py:774-787); products on the GPU."""
    gram, ordinate = gram_device(x, y)
    return lu_factorization(gram, ordinate)


def apply_weights(x, y, weights):
    """Rows and targets scaled by sqrt(weight) (least_squares.py:899-913)."""
    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    if weights is None:
        return x, y
    weights = np.asarray(weights, dtype=float)
    if len(weights) != len(x) or np.any(weights < 0):
        raise ValueError("Weights must be non-negative, one per sample.")
    w = np.sqrt(weights)
    return x * w[:, None], y * w


def weighted_least_squares(x, y, weights=None, regularizer=None):
    """Weighted, optionally regularised solve (least_squares.py:790-814)."""
    x, y = apply_weights(x, y, weights)
    gram, ordinate = gram_device(x, y)
    if regularizer is not None:
        regularizer = np.asarray(regularizer, dtype=float)
        gram = gram + regularizer.T @ regularizer
    return lu_factorization(gram, ordinate)


def validate_regularizer(regularizer, n_feats):
    """Shape check of a user-supplied penalty matrix (least_squares.py:916-930)."""
    n_row, n_col = np.shape(regularizer)
    if n_col != n_feats:
        raise ValueError(f"Expected regularizer shape: N x {n_feats}. Provided: {n_row} x {n_col}")


def apply_weighted_gram(gram_matrix, weight):
    return gram_matrix * weight ** 2
