"""
Penalty matrices for the regularised normal equations (host side, NumPy).

Same public functions as the reference's ``uf3/regression/regularize.py``
(:16-201): ridge identity, 1-D second-difference operator with halved corner
entries, 2-D / 3-D nearest-neighbour Laplacians, block-diagonal combination.
Entries are small integers, so equality with the reference is exact.
"""
import numpy as np

DEFAULT_REGULARIZER_GRID = dict(ridge_1b=1e-16,
                                ridge_2b=0.0,
                                ridge_3b=1e-10,
                                curve_2b=1e-16,
                                curve_3b=1e-16)


def get_ridge_penalty_matrix(n_features):
    return np.eye(n_features)


def get_curvature_penalty_matrix_1D(n_features):
    n = n_features
    m = -2.0 * np.eye(n) + np.eye(n, k=1) + np.eye(n, k=-1)
    m[0, 0] *= 0.5
    m[n - 1, n - 1] *= 0.5
    return m


def _laplacian(shape):
    """Graph Laplacian (negated degree on the diagonal) of an n-D grid, rows in C order."""
    size = int(np.prod(shape))
    out = np.zeros((size,) + tuple(shape))
    idx = np.arange(size).reshape(shape)
    for axis, n in enumerate(shape):
        lo = [slice(None)] * len(shape)
        hi = [slice(None)] * len(shape)
        lo[axis], hi[axis] = slice(0, n - 1), slice(1, n)
        a, b = idx[tuple(lo)].ravel(), idx[tuple(hi)].ravel()
        flat = out.reshape(size, size)
        flat[a, b] = 1.0   # +1 neighbour along the axis
        flat[b, a] = 1.0   # -1 neighbour
    flat = out.reshape(size, size)
    flat[np.arange(size), np.arange(size)] = -flat.sum(axis=1)
    return out


def get_curvature_penalty_matrix_2D(L, M, flatten=True):
    m = _laplacian((L, M))
    return m.reshape(L * M, L * M) if flatten else m


def get_curvature_penalty_matrix_3D(L, M, N, flatten=True):
    m = _laplacian((L, M, N))
    return m.reshape(L * M * N, L * M * N) if flatten else m


def combine_regularizer_matrices(matrices):
    rows = [m.shape[0] for m in matrices]
    cols = [m.shape[1] for m in matrices]
    full = np.zeros((int(np.sum(rows)), int(np.sum(cols))))
    r0 = c0 = 0
    for m, nr, nc in zip(matrices, rows, cols):
        full[r0:r0 + nr, c0:c0 + nc] = m
        r0 += nr
        c0 += nc
    return full
