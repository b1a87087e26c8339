"""
Explicit periodic-image tiling in the reference's order (host, NumPy).

The device kernels never build a supercell -- they use periodic neighbour lists
with image shifts -- but the *numbering* of the reference's ghost atoms
(``uf3/data/geometry.py:14-149``: real atoms first, images ordered b-slowest /
a-middle / c-fastest with per-axis order 0, +1, -1, +2, -2, ...) defines the
neighbour indices parity is checked against, so the same tiling is offered here
for users and tests.
"""
import numpy as np

from uf3_amd.data.atoms import Atoms


def get_supercell_factors(cell, r_cut=10):
    cell = np.asarray(cell, dtype=float).reshape(3, 3)
    if np.all(cell == 0) or np.any(np.linalg.norm(cell, axis=1) == 0):
        return np.array([1.0, 1.0, 1.0])
    a, b, c = cell
    normals = [np.cross(b, c), np.cross(a, c), np.cross(a, b)]
    heights = [abs(np.dot(v, n)) / np.linalg.norm(n) for v, n in zip(cell, normals)]
    return np.ceil([r_cut / h for h in heights])


def generate_periodic_image_indices(cell, r_cut):
    out = []
    for f in get_supercell_factors(cell, r_cut):
        idx = [0]
        for k in range(1, int(f) + 1):
            idx += [k, -k]
        out.append(np.array(idx))
    return out


def image_shifts(cell, pbc, r_cut):
    """(n_images, 3) integer shifts in the reference's enumeration order."""
    axes = generate_periodic_image_indices(cell, r_cut)
    for d in range(3):
        if not pbc[d]:
            axes[d] = axes[d][:1]
    return np.array([[a, b, c] for b in axes[1] for a in axes[0] for c in axes[2]], dtype=int)


def get_supercell(geometry, r_cut=10, sort_indices=False):
    cell = np.array(geometry.get_cell(), dtype=float).reshape(3, 3)
    pbc = np.asarray(geometry.get_pbc() if hasattr(geometry, "get_pbc") else geometry.pbc)
    shifts = image_shifts(cell, pbc, r_cut)
    if sort_indices:
        order = np.argsort(np.linalg.norm(shifts @ cell, axis=1), kind="stable")
        shifts = shifts[order]
    pos = np.asarray(geometry.get_positions(), dtype=float)
    z = np.asarray(geometry.get_atomic_numbers())
    sup_pos = np.concatenate([pos + s @ cell for s in shifts])
    return Atoms(numbers=np.tile(z, len(shifts)), positions=sup_pos)
