"""
Module-level pair-distance surfaces of the reference's ``uf3/representation/distances.py``
(``distances_by_interaction`` :19-75, ``derivatives_by_interaction`` :78-143,
``mask_supercell_with_radius`` :146-169, ``mask_matrix_by_pair_interaction`` :172-209,
``get_distance_matrix`` :212-235, ``get_distance_derivatives`` :238-277,
``distances_from_geometry`` :280-304, ``kronecker_*`` :307-328,
``compute_direction_cosines`` :331-364), same signatures and return layouts, computed by
``libuf3hip.so``.

Two routes, chosen from the ``supercell`` argument:

* **lists** -- ``supercell`` is ``None`` / the frame itself, or a tiling of the frame by whole
  lattice images in blocks of ``len(geom)`` atoms (what ``geometry.get_supercell`` builds, with any
  cut-off and either image order): the device's cell list finds the pairs (``uf3_pair_geometry``:
  indices in the reference's supercell numbering, distances, unit vectors), O(pairs) memory, any
  frame size.  The pairs are re-numbered to the blocks of the supercell that was passed, so the
  order of the returned distances is the reference's ``np.where`` order on ITS matrix;
* **dense** -- any other atom set: the n x m distance matrix from ``uf3_distance_matrix`` (scipy
  ``cdist``'s order of operations) and the reference's own masks; O(n m) memory like the reference.

There is no CPU arithmetic in either: distances, quotients and direction cosines come from the
kernels; the host selects, sorts and scatters.  Without the library / a gfx950 device every function
that computes raises ``uf3_amd._lib.HipUnavailable``.
"""
import ctypes as C

import numpy as np

from uf3_amd import _lib
from uf3_amd.data import composition
from uf3_amd.data.atoms import Atoms


# ------------------------------------------------------------------------------ plumbing
def _pbc_of(geom):
    return np.asarray(geom.get_pbc() if hasattr(geom, "get_pbc") else geom.pbc, dtype=bool).reshape(3)


def lattice_images(geom, supercell):
    """
    How ``supercell`` relates to ``geom``:

    * ``None``               -- no images: ``supercell`` is None, ``geom`` itself or a copy of it;
    * int array (n_img, 3)   -- block b of ``len(geom)`` atoms is the frame shifted by ``shifts[b] @ cell``
                                (distinct shifts, block 0 = the frame);
    * ``False``              -- an arbitrary atom set.
    """
    if supercell is None or supercell is geom:
        return None
    own = np.asarray(geom.get_positions(), dtype=float).reshape(-1, 3)
    got = np.asarray(supercell.get_positions(), dtype=float).reshape(-1, 3)
    z_own, z_got = np.asarray(geom.get_atomic_numbers()), np.asarray(supercell.get_atomic_numbers())
    n = len(own)
    if got.shape == own.shape and np.array_equal(z_own, z_got) and np.array_equal(own, got):
        return None
    if n == 0 or len(got) % n or not np.array_equal(z_got, np.tile(z_own, len(got) // n)):
        return False
    blocks = got.reshape(-1, n, 3)
    if not np.array_equal(blocks[0], own):
        return False
    delta = blocks[:, 0, :] - own[0]
    scale = max(1.0, float(np.abs(got).max()))
    if not np.allclose(blocks - own[None], delta[:, None, :], rtol=0, atol=1e-9 * scale):
        return False
    cell = np.array(geom.get_cell(), dtype=float).reshape(3, 3)
    if not np.any(cell):
        return False
    with np.errstate(all="ignore"):
        frac = np.linalg.lstsq(cell.T, delta.T, rcond=None)[0].T
    shifts = np.rint(frac).astype(int)
    if not np.allclose(shifts @ cell, delta, rtol=0, atol=1e-9 * scale):
        return False
    if len({tuple(v) for v in shifts}) != len(shifts) or np.any(shifts[0]):
        return False
    # the tiled positions must be the reference's own sums (positions + np.dot(shift, cell), geometry.py:131-149):
    # only then are the distances the device forms from (atom, shift) the distances of the supercell that was passed
    for b, s in enumerate(shifts):
        if not np.array_equal(blocks[b], own + np.dot(s, cell)):
            return False
    return shifts


def _device_frame(geom, shifts):
    """(frame for the device, r_cut whose image range equals the tiling's, {device image rank: block}) or None when
    no cut-off reproduces the tiling's range (a lopsided hand-made tiling: the dense route takes it)."""
    from uf3_amd.data import geometry
    pos = np.asarray(geom.get_positions(), dtype=float).reshape(-1, 3)
    z = np.asarray(geom.get_atomic_numbers())
    cell = np.array(geom.get_cell(), dtype=float).reshape(3, 3)
    if shifts is None:
        return Atoms(numbers=z, positions=pos, cell=cell, pbc=False), 1.0, {0: 0}
    fac = np.abs(shifts).max(axis=0)
    pbc = fac > 0
    normals = [np.cross(cell[1], cell[2]), np.cross(cell[0], cell[2]), np.cross(cell[0], cell[1])]
    lo, hi = 0.0, np.inf
    for k in np.flatnonzero(pbc):
        # the reference's projected height (geometry.py:74-82) -- the device evaluates the same expression
        p = normals[k] * np.dot(cell[k], normals[k]) / np.dot(normals[k], normals[k])
        h = float(np.linalg.norm(p))
        lo, hi = max(lo, (fac[k] - 1) * h), min(hi, fac[k] * h)
    if not lo < hi:
        return None
    r_cut = 0.5 * (lo + hi) if np.isfinite(hi) else 1.0
    frame = Atoms(numbers=z, positions=pos, cell=cell, pbc=pbc)
    dev_shifts = geometry.image_shifts(cell, pbc, r_cut)
    block_of = {tuple(int(x) for x in s): b for b, s in enumerate(shifts)}
    rank_to_block = {r: block_of[tuple(int(x) for x in s)] for r, s in enumerate(dev_shifts)
                     if tuple(int(x) for x in s) in block_of}
    return frame, r_cut, rank_to_block


def _pair_lists(geom, shifts, pair_numbers, r_min, r_max):
    """Per pair of ``pair_numbers``: (i, j in the numbering of the supercell that was passed, distance, unit vector
    (R_j - R_i) / d), rows sorted by (i, j); i runs over the frame's atoms.  None: use the dense route."""
    dev = _device_frame(geom, shifts)
    if dev is None:
        return None
    frame, r_cut, rank_to_block = dev
    n = len(frame)
    zs = sorted({int(z) for z in frame.get_atomic_numbers()} | {int(z) for p in pair_numbers for z in p})
    ranges = {tuple(sorted(p)): (lo, hi) for p, lo, hi in zip(pair_numbers, r_min, r_max)}
    ctx = _lib.get_context()
    basis = _lib.RawDeviceBasis(zs, pairs=ranges, r_cut=r_cut, ctx=ctx)
    batch = _lib.FrameBatch([frame])
    P = len(basis.pairs)
    cnt = np.zeros(P, dtype=np.int64)
    args = (basis.handle, C.byref(batch.struct), _lib._p(batch.pos), _lib._p(batch.z))
    ctx.check(ctx.lib.uf3_pair_geometry(*args, _lib._p(cnt), None, None, 0))
    cap = max(1, int(cnt.max()) if P else 1)
    pij = np.zeros((P, cap, 2), dtype=np.int64)
    geo = np.zeros((P, cap, 4))
    ctx.check(ctx.lib.uf3_pair_geometry(*args, _lib._p(cnt), _lib._p(pij), _lib._p(geo), cap))
    rank_map = np.full(max(rank_to_block) + 2 if rank_to_block else 1, -1, dtype=np.int64)
    for r, b in rank_to_block.items():
        rank_map[r] = b
    out = []
    for p in pair_numbers:
        k = basis.pairs.index(tuple(sorted(int(z) for z in p)))
        i, j = pij[k, :cnt[k], 0], pij[k, :cnt[k], 1]
        g = geo[k, :cnt[k]]
        block = rank_map[np.minimum(j // max(n, 1), len(rank_map) - 1)] if n else j
        have = block >= 0                               # (images the supercell that was passed does not hold)
        i, jg, g = i[have], block[have] * n + j[have] % max(n, 1), g[have]
        order = np.lexsort((jg, i))
        out.append((i[order], jg[order], g[order, 0], g[order, 1:4]))
    return out


# ------------------------------------------------------------------------------ dense helpers
def _cdist(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 3)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1, 3)
    out = np.empty((len(a), len(b)))
    ctx = _lib.get_context()
    ctx.check(ctx.lib.uf3_distance_matrix(ctx.handle, _lib._p(a), len(a), _lib._p(b), len(b), _lib._p(out)))
    return out


def get_distance_matrix(geom, supercell=None):
    """(n x n) or (n x m) matrix of pair distances between ``geom`` and ``supercell`` (distances.py:212-235)."""
    if supercell is None:
        supercell = geom
    return _cdist(geom.get_positions(), supercell.get_positions())


def mask_matrix_by_pair_interaction(pair, geo_composition, sup_composition=None):
    """Boolean (n x m) mask of the entries whose two species are ``pair`` (distances.py:172-209)."""
    geo_composition = np.asarray(geo_composition)
    sup_composition = geo_composition if sup_composition is None else np.asarray(sup_composition)
    a, b = pair
    g, s = geo_composition[:, None], sup_composition[None, :]
    return ((g == a) & (s == b)) | ((g == b) & (s == a))


def mask_supercell_with_radius(geom, supercell, r_max):
    """Copy of ``supercell`` without the atoms further than ``r_max`` from every atom of ``geom`` (:146-169)."""
    keep = np.any(get_distance_matrix(geom, supercell) <= r_max, axis=0)
    return Atoms(numbers=np.asarray(supercell.get_atomic_numbers())[keep],
                 positions=np.asarray(supercell.get_positions())[keep],
                 cell=supercell.get_cell() if hasattr(supercell, "get_cell") else None,
                 pbc=_pbc_of(supercell) if hasattr(supercell, "get_pbc") or hasattr(supercell, "pbc") else False)


def kronecker_delta(m_range, i_where, j_where):
    m = np.asarray(m_range)[:, None]
    return ((m == np.asarray(j_where)[None, :]).astype(np.float64) - (m == np.asarray(i_where)[None, :]))


def kronecker_vectorized(n_atoms, i_where, j_where):
    m = np.arange(n_atoms)[:, None]
    return (m == np.asarray(j_where)[None, :]).astype(int) - (m == np.asarray(i_where)[None, :]).astype(int)


def compute_direction_cosines(sup_positions, distance_matrix, i_where, j_where, n_atoms):
    """drij_dR [n_atoms, 3, n_distances] = (delta(m, j) - delta(m, i)) (R_j - R_i) / r_ij (distances.py:331-364)."""
    pos = np.ascontiguousarray(sup_positions, dtype=np.float64).reshape(-1, 3)
    i_where = np.ascontiguousarray(i_where, dtype=np.int64)
    j_where = np.ascontiguousarray(j_where, dtype=np.int64)
    rij = np.ascontiguousarray(np.asarray(distance_matrix)[i_where, j_where], dtype=np.float64)
    out = np.empty((int(n_atoms), 3, len(i_where)))
    ctx = _lib.get_context()
    ctx.check(ctx.lib.uf3_direction_cosines(ctx.handle, _lib._p(pos), len(pos), _lib._p(i_where), _lib._p(j_where),
                                            _lib._p(rij), len(i_where), int(n_atoms), _lib._p(out)))
    return out


# ------------------------------------------------------------------------------ reference surfaces
def distances_by_interaction(geom, pair_tuples, r_min_map, r_max_map, supercell=None, atomic=False):
    """
    {pair: distances within (max(r_min, 0), r_max)} between ``geom`` and its ``supercell``, in the row-major order of
    the reference's masked distance matrix; ``atomic=True``: a list of per-atom arrays instead (distances.py:19-75).
    """
    pair_numbers = [tuple(composition.symbols2numbers(p)) for p in pair_tuples]
    r_min = [max(r_min_map[p], 0) for p in pair_tuples]
    r_max = [r_max_map[p] for p in pair_tuples]
    n = len(geom)
    shifts = lattice_images(geom, supercell)
    lists = None if shifts is False else _pair_lists(geom, shifts, pair_numbers, r_min, r_max)
    out = {}
    if lists is not None:
        for pair, (i, _, d, _) in zip(pair_tuples, lists):
            out[pair] = ([d[i == a] for a in range(n)] if atomic else d)
        return out
    dm = get_distance_matrix(geom, supercell)
    gz = np.asarray(geom.get_atomic_numbers())
    sz = np.asarray((supercell if supercell is not None else geom).get_atomic_numbers())
    for pair, pn, lo, hi in zip(pair_tuples, pair_numbers, r_min, r_max):
        mask = mask_matrix_by_pair_interaction(pn, gz, sz) & (dm > lo) & (dm < hi)
        out[pair] = [dm[a][mask[a]] for a in range(n)] if atomic else dm[mask]
    return out


def derivatives_by_interaction(geom, pair_tuples, r_cut, r_min_map, r_max_map, supercell=None):
    """
    ({pair: distances}, {pair: drij_dR [n_atoms, 3, n_distances]}) over the supercell atoms within ``r_cut`` of the
    frame, pairs with at least one real atom (distances.py:78-143).
    """
    pair_numbers = [tuple(composition.symbols2numbers(p)) for p in pair_tuples]
    r_min = [max(r_min_map[p], 0) for p in pair_tuples]
    r_max = [r_max_map[p] for p in pair_tuples]
    n = len(geom)
    shifts = lattice_images(geom, supercell)
    lists = None
    if shifts is not False and all(hi <= r_cut for hi in r_max):
        # (every listed pair lies within r_cut, so both its atoms survive the reference's radius mask, whose deletions
        # keep the order of the survivors: np.where order on the masked supercell = (i, j) order on the full one)
        lists = _pair_lists(geom, shifts, pair_numbers, r_min, r_max)
    distance_map, derivative_map = {}, {}
    if lists is not None:
        for pair, (i, j, d, u) in zip(pair_tuples, lists):
            ghost = j >= n
            # rows of a ghost atom: its pairs with real atoms, the transposes of the real atoms' pairs with it
            gi, gj, gd, gu = j[ghost], i[ghost], d[ghost], -u[ghost]
            order = np.lexsort((gj, gi))
            ii = np.concatenate([i, gi[order]])
            jj = np.concatenate([j, gj[order]])
            dd = np.concatenate([d, gd[order]])
            uu = np.concatenate([u, gu[order]])
            drij = np.zeros((n, 3, len(dd)))
            idx = np.arange(len(dd))
            real_j = jj < n
            drij[jj[real_j], :, idx[real_j]] = uu[real_j]
            real_i = ii < n
            drij[ii[real_i], :, idx[real_i]] = -uu[real_i]
            distance_map[pair], derivative_map[pair] = dd, drij
        return distance_map, derivative_map
    if supercell is None:
        supercell = geom
    sup = mask_supercell_with_radius(geom, supercell, r_cut)
    dm = get_distance_matrix(sup, sup)
    m = len(sup)
    idx = np.arange(m)
    real = (idx[:, None] < n) | (idx[None, :] < n)
    sz = np.asarray(sup.get_atomic_numbers())
    pos = sup.get_positions()
    for pair, pn, lo, hi in zip(pair_tuples, pair_numbers, r_min, r_max):
        mask = real & mask_matrix_by_pair_interaction(pn, sz, sz) & (dm > lo) & (dm < hi)
        distance_map[pair] = dm[mask]
        x, y = np.where(mask)
        derivative_map[pair] = compute_direction_cosines(pos, dm, x, y, n)
    return distance_map, derivative_map


def get_distance_derivatives(geom, supercell, r_min=0.0, r_max=10.0):
    """Legacy unary form: distances in (r_min, r_max] over the masked supercell and their drij_dR (:238-277)."""
    sup_pos = np.asarray(supercell.get_positions(), dtype=float)
    geo_pos = np.asarray(geom.get_positions(), dtype=float)
    keep = np.any(_cdist(geo_pos, sup_pos) <= r_max, axis=0)
    sup_pos = sup_pos[keep]
    dm = _cdist(sup_pos, sup_pos)
    mask = (dm > max(r_min, 0)) & (dm <= r_max)
    i, j = np.where(mask)
    return dm[mask], compute_direction_cosines(sup_pos, dm, i, j, len(geo_pos))


def distances_from_geometry(geom, supercell=None, r_min=0.0, r_max=10.0):
    """Legacy unary form: flattened distances in (r_min, r_max) (distances.py:280-304)."""
    dm = get_distance_matrix(geom, supercell)
    return dm[(dm > r_min) & (dm < r_max)]
