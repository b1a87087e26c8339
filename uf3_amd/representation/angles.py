"""
Module-level three-body surfaces of the reference's ``uf3/representation/angles.py``
(``featurize_energy_3b`` :17-78, ``coefficient_counts_from_knots`` :81-102,
``featurize_force_3b`` :142-232, ``identify_ij`` :289-346, ``group_idx_by_center`` :399-421,
``symmetrize_3B`` :645-674, ``get_symmetry_weights`` :677-735), same signatures and return
layouts, computed by ``libuf3hip.so``.

``featurize_energy_3b`` / ``featurize_force_3b`` return the RAW L x M x N grids per interaction
(no symmetry fold, no template mask; basis functions below ``n_lead`` / above ``-n_trail`` switched
off per dimension, as ``evaluate_triplet_distances`` :517-574 does).  They run the same featurizer
kernels as ``BasisFeaturizer`` on device tables whose output columns ARE the raw bins (an identity
look-up table instead of the compressed one), so the grids come out of ``uf3_featurize`` directly;
the ``basis_functions`` argument is accepted for signature compatibility and not used -- the device
evaluates the splines from the knot sequences.  The periodic images come from the frame's own cell
(see ``distances.lattice_images``): ``supercell`` must be ``None`` / the frame (a cluster) or a tiling
of the frame by lattice images in the reference's image order, any cut-off; other atom sets are
refused, as ``BasisFeaturizer`` does.

``identify_ij`` returns the dense supercell distance matrix by contract, so it takes the dense route
(``uf3_distance_matrix``) and applies the reference's mask on the host.
"""
import numpy as np

from uf3_amd import _lib
from uf3_amd.data import composition
from uf3_amd.representation import distances
from uf3_amd.representation.bspline import get_symmetry_weights  # noqa: F401  (angles.py:677-735 lives there)


def coefficient_counts_from_knots(knot_sets):
    """Basis functions per dimension and interaction: three lists L, M, N (angles.py:81-102)."""
    L, M, N = [], [], []
    for l_space, m_space, n_space in knot_sets:
        L.append(len(l_space) - 4)
        M.append(len(m_space) - 4)
        N.append(len(n_space) - 4)
    return L, M, N


def _centre_leg_range(knot_sets):
    """(r_min, r_max) of the 3-body neighbour search: lowest knot of any leg, highest knot of the legs that start at
    the centre atom (angles.py:309-325)."""
    flat = np.concatenate([np.asarray(seq, dtype=float) for set_ in knot_sets for seq in set_])
    r_min = max(np.min(flat), 0)
    flat = np.concatenate([np.asarray(seq, dtype=float) for set_ in knot_sets
                           for seq in set_[:int((1 + np.sqrt(1 + 8 * len(set_))) / 2 - 1)]])
    return r_min, np.max(flat)


def identify_ij(geom, knot_sets, supercell=None, square=False):
    """
    Neighbour pairs of the three-body terms: distances in (r_min, r_max] (angles.py:289-346).

    square=False: (supercell distance matrix, i_where, j_where) with i over the frame's atoms;
    square=True:  (supercell positions, distance matrix, i_where, j_where) with i over the whole supercell.
    """
    if supercell is None:
        supercell = geom
    r_min, r_max = _centre_leg_range(knot_sets)
    sup_positions = np.asarray(supercell.get_positions(), dtype=float)
    n_geo = len(geom)
    dist_matrix = distances._cdist(sup_positions, sup_positions)
    if square is False:
        cut = dist_matrix[:n_geo, :]
        i_where, j_where = np.where((cut > r_min) & (cut <= r_max))
        return dist_matrix, i_where, j_where
    i_where, j_where = np.where((dist_matrix > r_min) & (dist_matrix <= r_max))
    return sup_positions, dist_matrix, i_where, j_where


def group_idx_by_center(i_where, j_where):
    """(unique centre indices, list of their neighbour-index arrays) (angles.py:399-421)."""
    i_values, group_sizes = np.unique(i_where, return_counts=True)
    return i_values, np.array_split(j_where, np.cumsum(group_sizes)[:-1])


def symmetrize_3B(grid_3b, symmetry=2):
    """Mirror a grid over its symmetry planes with the diagonal weights 1/2, 1/6 (angles.py:645-674)."""
    grid_3b = np.asarray(grid_3b, dtype=float)
    i, j, k = np.indices(grid_3b.shape)
    template = np.ones_like(grid_3b)
    if symmetry == 2:
        template[i == j] = 0.5
        perms = [(0, 1, 2), (1, 0, 2)]
    elif symmetry == 3:
        template[(i == k) | (i == j) | (j == k)] = 0.5
        template[(i == j) & (i == k)] = 1 / 6
        perms = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
    else:
        perms = [(0, 1, 2)]
    weighted = grid_3b * template
    return np.sum([weighted.transpose(p) for p in perms], axis=0)


# ------------------------------------------------------------------------------ raw grids through the featurizer
def _raw_tables(geom, knot_sets, hashes, n_lead, n_trail, r_cut):
    """Device tables whose 3-body columns are the raw (l, m, n) bins kept by n_lead / n_trail, + the bins' indices."""
    trios, kept = [], []
    for ks, h in zip(knot_sets, hashes):
        zt = tuple(int(z) for z in composition.unpack_szudzik_hash([int(h)], 3)[0])
        if composition.get_szudzik_hash(np.array([zt]))[0] != int(h) or zt[1] > zt[2] or min(zt) < 1:
            raise ValueError(f"hash {h} is not the Szudzik hash of a (centre, Z1 <= Z2) triplet of atomic numbers")
        dims = [len(k) - 4 for k in ks]
        on = [np.zeros(d, dtype=bool) for d in dims]
        for d, flags in zip(dims, on):
            flags[n_lead:max(n_lead, d - n_trail)] = True
        keep = on[0][:, None, None] & on[1][None, :, None] & on[2][None, None, :]
        lut = np.full(keep.shape, -1, dtype=np.int32)
        lut[keep] = np.arange(int(keep.sum()), dtype=np.int32)
        trios.append((zt, ks, lut.ravel()))
        kept.append(keep)
    zs = sorted({int(z) for z in geom.get_atomic_numbers()} | {z for zt, _, _ in trios for z in zt})
    return _lib.RawDeviceBasis(zs, pairs={}, trios=trios, r_cut=r_cut), kept


def _device_frame_3b(geom, supercell, what):
    shifts = distances.lattice_images(geom, supercell)
    dev = None if shifts is False else distances._device_frame(geom, shifts)
    if dev is None:
        raise ValueError(f"{what}: supercell is neither the frame itself nor a tiling of it by whole lattice images "
                         "(geometry.get_supercell); the GPU featurizer takes its periodic images from the frame's own "
                         "cell and cannot honour an arbitrary atom set")
    frame, r_cut, rank_to_block = dev
    if shifts is not None:
        # ghost-centre numbering (angles.py:451-460, 474) follows the supercell's image order: the device reproduces the
        # reference's own order only
        if any(rank_to_block.get(r) != r for r in range(len(shifts))):
            raise ValueError(f"{what}: supercell images are not in geometry.get_supercell's order (sort_indices=True?)")
    return frame, r_cut


def _rows(geom, knot_sets, hashes, supercell, n_lead, n_trail, energy, forces, what):
    import ctypes as C
    frame, r_cut = _device_frame_3b(geom, supercell, what)
    basis, kept = _raw_tables(frame, knot_sets, hashes, n_lead, n_trail, r_cut)
    ctx = basis.ctx
    batch = _lib.FrameBatch([frame])
    F, n = basis.n_feat, len(frame)
    x_e = np.empty((1, F)) if energy else None
    x_f = np.empty((n, 3, F)) if forces else None
    if n:
        ctx.check(ctx.lib.uf3_featurize(basis.handle, C.byref(batch.struct), _lib._p(batch.pos), _lib._p(batch.z),
                                        _lib._p(x_e), _lib._p(x_f)))
    elif energy:
        x_e[:] = 0.0
    return basis, kept, x_e, x_f


def featurize_energy_3b(geom, knot_sets, basis_functions, hashes, supercell=None, n_lead=0, n_trail=0):
    """Energy feature grids, one L x M x N array per interaction of ``hashes`` (angles.py:17-78)."""
    basis, kept, x_e, _ = _rows(geom, knot_sets, hashes, supercell, n_lead, n_trail, True, False, "featurize_energy_3b")
    grids = []
    for keep, col, ncol in zip(kept, basis.trio_col, basis.trio_ncol):
        grid = np.zeros(keep.shape)
        grid[keep] = x_e[0, col:col + int(keep.sum())]
        grids.append(grid)
    return grids


def featurize_force_3b(geom, knot_sets, basis_functions, trio_hashes, supercell=None, n_lead=0, n_trail=0):
    """Force feature grids ``[interaction][atom][x | y | z]`` -> L x M x N array (angles.py:142-232)."""
    basis, kept, _, x_f = _rows(geom, knot_sets, trio_hashes, supercell, n_lead, n_trail, False, True,
                                "featurize_force_3b")
    out = []
    for keep, col, ncol in zip(kept, basis.trio_col, basis.trio_ncol):
        w = int(keep.sum())
        per_atom = []
        for a in range(len(geom)):
            comps = []
            for c in range(3):
                grid = np.zeros(keep.shape)
                grid[keep] = x_f[a, c, col:col + w]
                comps.append(grid)
            per_atom.append(comps)
        out.append(per_atom)
    return out
