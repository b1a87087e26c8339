"""
``BasisFeaturizer``: energy / force feature rows of atomic configurations,
computed by the MI355X kernels in ``uf3_amd/csrc`` through ``libuf3hip.so``.

Surface follows the reference's ``uf3/representation/process.py`` (class :20-536:
``featurize_{energy,force}_{2B,3B}`` :369-506, ``evaluate_configuration`` :293-367,
``evaluate`` :121-174, ``evaluate_parallel`` :196-254, ``flatten_by_interactions``
:619-630).  Differences in mechanism, not in results:

* no supercell is materialised: periodic images come from a cell list with image
  shifts on the device.  The optional ``supercell`` argument of ``featurize_*`` selects the
  boundary conditions like in the reference: ``None`` (or ``geom`` itself) -> ``geom`` is an
  isolated cluster; the supercell the reference builds for ``geom``
  (``geometry.get_supercell(geom, r_cut=basis.r_cut)``) -> the periodic frame.  Any other
  atom set (a hand-made supercell, another cut-off) is refused rather than silently
  replaced by the frame's own periodicity;
* frames are processed in batches (``featurize_frames``); ``evaluate_parallel`` needs
  no worker pool -- the whole batch goes to the GPU of this process.

There is no CPU fallback: without ``libuf3hip.so`` and a gfx950 device every method
that computes features raises ``uf3_amd._lib.HipUnavailable``.
"""
import warnings

import numpy as np

from uf3_amd import _lib


class BasisFeaturizer:
    def __init__(self, bspline_config, fit_forces=True, prefix='x', device=None):
        self.bspline_config = bspline_config
        self.fit_forces = fit_forces
        self.prefix = prefix
        self.device = device
        self.columns = self.bspline_config.get_column_names()

    chemical_system = property(lambda self: self.bspline_config.chemical_system)
    degree = property(lambda self: self.chemical_system.degree)
    element_list = property(lambda self: self.chemical_system.element_list)
    interactions_map = property(lambda self: self.chemical_system.interactions_map)
    r_min_map = property(lambda self: self.bspline_config.r_min_map)
    r_max_map = property(lambda self: self.bspline_config.r_max_map)
    resolution_map = property(lambda self: self.bspline_config.resolution_map)
    r_cut = property(lambda self: self.bspline_config.r_cut)
    knots_map = property(lambda self: self.bspline_config.knots_map)
    knot_subintervals = property(lambda self: self.bspline_config.knot_subintervals)
    basis_functions = property(lambda self: self.bspline_config.basis_functions)
    partition_sizes = property(lambda self: self.bspline_config.partition_sizes)
    interaction_hashes = property(lambda self: self.chemical_system.interaction_hashes)
    leading_trim = property(lambda self: self.bspline_config.leading_trim)
    trailing_trim = property(lambda self: self.bspline_config.trailing_trim)

    @staticmethod
    def from_config(bspline_config, config):
        keys = ['prefix', 'fit_forces']
        return BasisFeaturizer(bspline_config, **{k: v for k, v in config.items() if k in keys})

    def __repr__(self):
        return "\n".join(["BasisFeaturizer:", f"    Fit forces: {self.fit_forces}",
                          f"    Column prefix: {self.prefix}", repr(self.bspline_config)])

    # ------------------------------------------------------------------ device plumbing
    def _dev(self):
        ctx = _lib.get_context(self.device)
        return ctx, _lib.device_basis(self.bspline_config, ctx)

    def featurize_frames(self, atoms_list, energy=True, forces=True, periodic=None, max_bytes=2 << 30, out=None):
        """
        Feature rows of a batch of frames (host arrays in, host arrays out).

        Returns (x_e [n_frames, F] | None, x_f [sum N, 3, F] | None, offsets [n_frames+1]).
        Column order = ``get_column_names()[1:]`` (no ``y``).  Long lists go to the device in chunks of at
        most ``max_bytes`` of force rows, so the staging buffers stay bounded.

        ``out=(x_e, x_f)``: C-contiguous float64 arrays of those shapes to fill instead of fresh ones.  Worth it
        for repeated calls: 104 MB of rows per 10k-atom frame come back at ~46 GB/s into memory that has been
        touched before, at ~15 GB/s into a fresh ``np.empty`` (first-touch page faults): 440 vs 140 frames/s.
        """
        import ctypes as C
        ctx, db = self._dev()
        F = db.n_feat
        counts = [len(a) for a in atoms_list]
        offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        x_e = np.empty((len(atoms_list), F)) if energy else None
        x_f = np.empty((int(offsets[-1]), 3, F)) if forces else None
        if out is not None:
            for given, fresh, name in ((out[0], x_e, "x_e"), (out[1], x_f, "x_f")):
                if fresh is not None and (given is None or given.shape != fresh.shape or given.dtype != np.float64
                                          or not given.flags.c_contiguous):
                    raise ValueError(f"out: {name} must be a C-contiguous float64 array of shape {fresh.shape}")
            x_e = out[0] if energy else None
            x_f = out[1] if forces else None
        per_atom = 24 * F if forces else 0
        start = 0
        while start < len(atoms_list):
            stop, nbytes = start, 0
            while stop < len(atoms_list) and (stop == start or nbytes + counts[stop] * per_atom <= max_bytes):
                nbytes += counts[stop] * per_atom
                stop += 1
            batch = _lib.FrameBatch(atoms_list[start:stop], periodic=periodic)
            xe_c = x_e[start:stop] if energy else None
            xf_c = x_f[offsets[start]:offsets[stop]] if forces else None
            ctx.check(ctx.lib.uf3_featurize(db.handle, C.byref(batch.struct), _lib._p(batch.pos), _lib._p(batch.z),
                                            _lib._p(xe_c), _lib._p(xf_c)))
            start = stop
        return x_e, x_f, offsets

    def featurize_device(self, frames_struct, d_pos, d_z, d_x_e=None, d_x_f=None, ld=None):
        """Device-resident entry: raw HBM pointers (ints), asynchronous on the context stream.  ``ld``: doubles between
        consecutive force rows (default: the number of features, dense rows; a multiple of 16 puts every row on a cache
        line of its own, ``aligned_ld``)."""
        import ctypes as C
        ctx, db = self._dev()
        if ld is None or int(ld) == db.n_feat:
            ctx.check(ctx.lib.uf3_featurize_dev(db.handle, C.byref(frames_struct), C.c_void_p(d_pos), C.c_void_p(d_z),
                                                C.c_void_p(d_x_e or 0), C.c_void_p(d_x_f or 0)))
        else:
            ctx.check(ctx.lib.uf3_featurize_ld_dev(db.handle, C.byref(frames_struct), C.c_void_p(d_pos), C.c_void_p(d_z),
                                                   C.c_void_p(d_x_e or 0), C.c_void_p(d_x_f or 0), int(ld)))

    @staticmethod
    def aligned_ld(n_feat):
        """the row stride (doubles) that starts every force row on a 128-byte line"""
        return (int(n_feat) + 15) // 16 * 16

    def get_training_tuples(self, df_features, kappa, data_coordinator):
        """(x, y, row weights) of a feature table -- deprecated in the reference (``process.py:508-535``), kept for scripts that
        still call it; see ``dataframe_to_training_tuples``."""
        import warnings
        warnings.warn("get_training_tuples() is deprecated.", DeprecationWarning)
        return dataframe_to_training_tuples(df_features, kappa=kappa, energy_key=data_coordinator.energy_key)

    def neighbor_indices(self, geom):
        """
        Neighbour indices in the reference's supercell numbering:
        ({pair: (n, 2) int64 (i, j)} in row-major order, (n, 2) int64 3-body pairs).
        """
        import ctypes as C
        ctx, db = self._dev()
        batch = _lib.FrameBatch([geom])
        P = len(db.pairs)
        cnt = np.zeros(max(P, 1), dtype=np.int64)
        n3 = np.zeros(1, dtype=np.int64)
        args = (db.handle, C.byref(batch.struct), _lib._p(batch.pos), _lib._p(batch.z))
        ctx.check(ctx.lib.uf3_neighbors_debug(*args, _lib._p(cnt), None, 0, _lib._p(n3), None, 0))
        cap2, cap3 = max(1, int(cnt.max())), max(1, int(n3[0]))
        pij = np.zeros((max(P, 1), cap2, 2), dtype=np.int64)
        nij = np.zeros((cap3, 2), dtype=np.int64)
        ctx.check(ctx.lib.uf3_neighbors_debug(*args, _lib._p(cnt), _lib._p(pij), cap2, _lib._p(n3), _lib._p(nij), cap3))
        return ({p: pij[k, :cnt[k]].copy() for k, p in enumerate(db.pairs)}, nij[:int(n3[0])].copy())

    def product_n3_indices(self, geom):
        """
        The 3-body neighbour pairs (i, j) -- reference supercell numbering, row-major order as ``identify_ij`` returns them
        (angles.py:289-346) -- read back from the lists the featurizer's OWN launches build and consume for this frame (one
        energy-row call, then ``uf3_n3_lists_debug``), not from the separate walk behind ``neighbor_indices``.  Test surface.
        """
        import ctypes as C
        ctx, db = self._dev()
        self.featurize_frames([geom], energy=True, forces=False)
        n = len(geom)
        cap = C.c_int64(0)
        ctx.check(ctx.lib.uf3_n3_lists_debug(db.handle, n, C.byref(cap), None, None, 0))
        cnt = np.zeros(n, dtype=np.int32)
        sidx = np.zeros((n, cap.value), dtype=np.int32)
        ctx.check(ctx.lib.uf3_n3_lists_debug(db.handle, n, C.byref(cap), _lib._p(cnt), _lib._p(sidx), cap.value))
        keep = np.arange(cap.value)[None, :] < cnt[:, None]
        i = np.broadcast_to(np.arange(n, dtype=np.int64)[:, None], sidx.shape)[keep]
        j = sidx[keep].astype(np.int64)
        order = np.lexsort((j, i))
        return np.stack([i[order], j[order]], axis=1)

    # ------------------------------------------------------------------ reference surface
    def _block(self, degree):
        sizes, offsets = self.bspline_config.get_interaction_partitions()
        inter = self.interactions_map[degree]
        lo = int(offsets[inter[0]])
        hi = int(offsets[inter[-1]] + sizes[inter[-1]])
        return lo, hi

    def _rows(self, geom, supercell, energy, forces):
        periodic = self._supercell_means_periodic(geom, supercell)
        return self.featurize_frames([geom], energy=energy, forces=forces, periodic=periodic)

    def _supercell_means_periodic(self, geom, supercell):
        """None (= periodic images from the frame's own cell) or False (cluster); see the module docstring."""
        if supercell is None or supercell is geom:
            return False
        from uf3_amd.data import geometry
        got = np.asarray(supercell.get_positions(), dtype=float).reshape(-1, 3)
        own = np.asarray(geom.get_positions(), dtype=float).reshape(-1, 3)
        z_got, z_own = np.asarray(supercell.get_atomic_numbers()), np.asarray(geom.get_atomic_numbers())
        if got.shape == own.shape and np.array_equal(z_got, z_own) and np.allclose(got, own, rtol=0, atol=1e-9):
            return False                                             # a copy of the frame: no images
        # Any tiling of the frame by whole lattice images (blocks of N atoms, each the frame shifted by an integer
        # combination of the cell vectors) that covers at least the reference's image range gives the reference's
        # features: a supercell built with a larger r_cut (get_supercell's default is 10) or with sort_indices=True is as
        # good as get_supercell(geom, basis.r_cut).  The images themselves come from the frame's own cell on the device.
        n = len(own)
        cell = np.array(geom.get_cell(), dtype=float).reshape(3, 3)
        pbc = np.asarray(geom.get_pbc() if hasattr(geom, "get_pbc") else geom.pbc, dtype=bool)
        problem = None
        if n == 0 or len(got) % n or not np.array_equal(z_got, np.tile(z_own, len(got) // max(n, 1))):
            problem = f"{len(got)} atoms given: not whole images of the frame's {n} atoms"
        else:
            blocks = got.reshape(-1, n, 3)
            delta = blocks[:, 0, :] - own[0]
            if not np.allclose(blocks - own[None], delta[:, None, :], rtol=0, atol=1e-8):
                problem = "its blocks are not rigid translations of the frame"
            else:
                with np.errstate(all="ignore"):
                    frac = np.linalg.lstsq(cell.T, delta.T, rcond=None)[0].T if np.any(cell) else np.zeros_like(delta)
                shifts = np.rint(frac).astype(int)
                if not np.allclose(shifts @ cell, delta, rtol=0, atol=1e-8) or np.any(shifts[:, ~pbc] != 0):
                    problem = "its blocks are not shifted by whole lattice vectors along the periodic axes"
                else:
                    have = {tuple(v) for v in shifts}
                    need = {tuple(v) for v in geometry.image_shifts(cell, pbc, self.r_cut)}
                    if len(have) != len(shifts) or not need <= have:
                        problem = (f"it holds {len(have)} distinct images, the cut-off {self.r_cut} needs the "
                                   f"{len(need)} images of get_supercell(geom, r_cut=basis.r_cut)")
                    elif have != need:
                        # more images than the reference's range: the same features as long as no atom sits FAR outside
                        # its cell (for those the reference's rows depend on the image range, geometry.py:131-149).  The
                        # window is the device's (make_geom in uf3_hip.hip): inside a fractional width fac + 1 - r_cut / h
                        # around the cell, atoms within r_cut of each other are at most fac images apart.
                        fr_own = own @ np.linalg.inv(cell) if abs(np.linalg.det(cell)) > 0 else np.zeros_like(own)
                        normals = [np.cross(cell[1], cell[2]), np.cross(cell[2], cell[0]), np.cross(cell[0], cell[1])]
                        for k in np.flatnonzero(pbc):
                            h = abs(np.dot(cell[k], normals[k])) / max(np.linalg.norm(normals[k]), 1e-300)
                            w = np.ceil(self.r_cut / h) + 1.0 - self.r_cut / h - 1e-9
                            if np.any((fr_own[:, k] < 0.5 - 0.5 * w) | (fr_own[:, k] > 0.5 + 0.5 * w)):
                                problem = ("it holds more images than get_supercell(geom, r_cut=basis.r_cut) and the frame has "
                                           "atoms far outside its cell, whose reference rows depend on the image range")
        if problem is None:
            return None
        raise ValueError(
            "supercell is neither the frame itself nor a tiling of it by lattice images that covers "
            f"get_supercell(geom, r_cut=basis.r_cut): {problem}; the GPU featurizer takes its periodic images from "
            "the frame's own cell and cannot honour an arbitrary atom set")

    def featurize_energy_2B(self, geom, supercell=None):
        lo, hi = self._block(2)
        return self._rows(geom, supercell, True, False)[0][0, lo:hi]

    def featurize_force_2B(self, geom, supercell=None):
        lo, hi = self._block(2)
        return self._rows(geom, supercell, False, True)[1][:, :, lo:hi]

    def featurize_energy_3B(self, geom, supercell=None):
        lo, hi = self._block(3)
        return self._rows(geom, supercell, True, False)[0][0, lo:hi]

    def featurize_force_3B(self, geom, supercell=None):
        lo, hi = self._block(3)
        return self._rows(geom, supercell, False, True)[1][:, :, lo:hi]

    def evaluate_configuration(self, geom, name=None, energy=None, forces=None, energy_key="energy"):
        """dict of (1 + F)-vectors: the energy row and the 3N force rows, ``y`` first."""
        eval_map = {}
        invalid = set(geom.get_chemical_symbols()).difference(self.element_list)
        if invalid:
            msg = "Invalid elements: {}".format(', '.join(invalid))
            if name is not None:
                msg += " in configuration " + name
            warnings.warn(msg, RuntimeWarning)
            return dict()
        any_pbc = bool(np.any(geom.get_pbc() if hasattr(geom, "get_pbc") else geom.pbc))
        x_e, x_f, _ = self.featurize_frames([geom], energy=energy is not None, forces=forces is not None,
                                            periodic=None if any_pbc else False)
        if energy is not None:
            key = (name, energy_key) if name is not None else energy_key
            eval_map[key] = np.insert(x_e[0], 0, energy)
        if forces is not None:
            n_atoms = len(geom)
            for j, component in enumerate(['fx', 'fy', 'fz']):
                for i in range(n_atoms):
                    atom_index = component + '_' + str(i)
                    key = (name, atom_index) if name is not None else atom_index
                    eval_map[key] = np.insert(x_f[i, j, :], 0, forces[j][i])
        return eval_map

    def evaluate(self, df_data, atoms_key="geometry", energy_key="energy", progress="bar"):
        """
        DataFrame in -> feature DataFrame out (MultiIndex (name, 'energy'|'fx_i'|...), ``y`` first; process.py:121-194).
        Same rows in the same order as calling ``evaluate_configuration`` frame by frame, as the reference does, but
        the frames go to the GPU in batches (grouped by boundary-condition class and by whether force rows are
        wanted) and the table is assembled as one array instead of 3N ``np.insert`` calls per frame.
        """
        import pandas as pd
        header = list(df_data.columns)
        jobs = []                                                   # (name, geom, energy, forces)
        for name, row in zip(df_data.index, df_data.itertuples(index=False, name=None)):
            rec = dict(zip(header, row))
            geom = rec[atoms_key]
            energy = rec.get(energy_key)
            forces = None
            if 'fx' in rec and self.fit_forces:
                forces = [rec[c] for c in ('fx', 'fy', 'fz')]
                if np.any(np.isnan(np.asarray(forces, dtype=float))):
                    forces = None
            invalid = set(geom.get_chemical_symbols()).difference(self.element_list)
            if invalid:                                             # process.py:321-330: warn, contribute nothing
                msg = "Invalid elements: {}".format(', '.join(invalid))
                if name is not None:
                    msg += " in configuration " + str(name)
                warnings.warn(msg, RuntimeWarning)
                continue
            if energy is None and forces is None:
                continue
            jobs.append((name, geom, energy, forces))
        F = len(self.columns) - 1
        rows_of = [(1 if e is not None else 0) + (3 * len(g) if f is not None else 0) for _, g, e, f in jobs]
        starts = np.concatenate([[0], np.cumsum(rows_of)]).astype(np.int64)
        table = np.empty((int(starts[-1]), F + 1))
        groups = {}
        for j, (_, geom, _, forces) in enumerate(jobs):
            any_pbc = bool(np.any(geom.get_pbc() if hasattr(geom, "get_pbc") else geom.pbc))
            groups.setdefault((any_pbc, forces is not None), []).append(j)
        for (any_pbc, with_forces), members in groups.items():
            frames = [jobs[j][1] for j in members]
            x_e, x_f, offsets = self.featurize_frames(frames, energy=True, forces=with_forces,
                                                      periodic=None if any_pbc else False)
            for k, j in enumerate(members):
                _, geom, energy, forces = jobs[j]
                r = int(starts[j])
                if energy is not None:
                    table[r, 0] = energy
                    table[r, 1:] = x_e[k]
                    r += 1
                if forces is not None:
                    n = len(geom)
                    table[r:r + 3 * n, 0] = np.asarray(forces, dtype=float).reshape(3, n).ravel()
                    # rows in the reference's order: fx_0 .. fx_{n-1}, fy_0 .., fz_0 ..
                    table[r:r + 3 * n, 1:] = x_f[offsets[k]:offsets[k + 1]].transpose(1, 0, 2).reshape(3 * n, F)
        index = []
        for name, geom, energy, forces in jobs:
            if energy is not None:
                index.append((name, energy_key))
            if forces is not None:
                n = len(geom)
                index.extend((name, f"{c}_{i}") for c in ('fx', 'fy', 'fz') for i in range(n))
        if not index:
            return self.arrange_features_dataframe({})
        return pd.DataFrame(table, index=pd.MultiIndex.from_tuples(index), columns=self.columns)

    def arrange_features_dataframe(self, eval_map):
        import pandas as pd
        if not eval_map:                     # nothing usable in the input: an empty table with the right columns
            return pd.DataFrame(np.empty((0, len(self.columns))), columns=self.columns,
                                index=pd.MultiIndex.from_arrays([[], []]))
        df = pd.DataFrame.from_dict(eval_map, orient='index', columns=self.columns)
        return df.set_index(pd.MultiIndex.from_tuples(df.index))

    def evaluate_parallel(self, df_data, client=None, atoms_key="geometry", energy_key="energy", n_jobs=2,
                          shuffle=True, progress="bar"):
        """Kept for drop-in use: the batch runs on this process's GPU, ``client`` is ignored."""
        return self.evaluate(df_data, atoms_key=atoms_key, energy_key=energy_key, progress=progress)


    def batched_to_hdf(self, filename, df_data, client=None, n_jobs=16, batch_size=50, progress="bar",
                       table_template="features_{}", **kwargs):
        """
        Feature tables of ``df_data`` in chunks of ``batch_size`` frames, each written as table
        ``features_000``, ``features_001`` ... of one HDF5 file -- the on-disk cache ``fit_from_file`` /
        ``batched_predict`` stream from (process.py:256-291).  Chunk boundaries, table names (zero-padded to at
        least three digits) and the skipping of chunks an existing file already holds follow the reference;
        every chunk is one GPU batch here (``client`` / ``n_jobs`` are accepted and ignored).  Writing goes
        through ``DataFrame.to_hdf`` (``save_feature_db``) and therefore needs PyTables.
        """
        import os
        idx_all = np.arange(len(df_data))
        idx_batches = np.array_split(idx_all, idx_all[batch_size::batch_size])
        idx_magnitude = max(int(np.ceil(np.log10(len(idx_batches)) + 0.1)), 3)
        chunk_names = []
        if os.path.isfile(filename):
            chunk_names = existing_feature_tables(filename)
            warnings.warn(f"File already exists: contains {len(chunk_names)} chunks.", RuntimeWarning)
        kwargs.pop("progress", None)
        kwargs.pop("n_jobs", None)
        kwargs.pop("shuffle", None)
        for j, idx_batch in enumerate(idx_batches):
            table_name = table_template.format(str(j).rjust(idx_magnitude, "0"))
            if table_name in chunk_names:
                continue
            df_features = self.evaluate(df_data.iloc[idx_batch], progress=False, **kwargs)
            save_feature_db(df_features, filename, table_name=table_name)


def save_feature_db(dataframe, filename, table_name='features'):
    """One feature table into an HDF5 file (process.py:538-547); needs PyTables."""
    dataframe.to_hdf(filename, key=table_name, mode="a", format='fixed')


def load_feature_db(filename, table_name='features'):
    """... and back (process.py:550-562)."""
    import pandas as pd
    return pd.read_hdf(filename, table_name)


def existing_feature_tables(filename):
    """Names of the tables an HDF5 feature file already holds (what io.analyze_hdf_tables reports, io.py:943-970)."""
    from uf3_amd.regression.least_squares import hdf_table_names
    return hdf_table_names(filename)


def flatten_by_interactions(vector_map, pair_tuples):
    return np.concatenate([vector_map[pair] for pair in pair_tuples], axis=-1)


def dataframe_to_training_tuples(df_features, kappa=0.5, energy_key='energy'):
    """
    The weights-per-row form of the training data (reference ``process.py:574-616``, deprecated there in favour of the
    Gram-level weights): x = all columns but the first, y = the first, and a weight per row -- energy rows
    ``kappa / (std_e n_e)``, force rows ``(1 - kappa) / (std_f n_f)`` with the population standard deviations of the two
    target groups.  Rows are energies where the last index level equals ``energy_key``.
    """
    if kappa < 0 or kappa > 1:
        raise ValueError("Invalid domain for kappa weighting parameter.")
    if len(df_features) <= 1:
        raise ValueError(f"Not enough samples ({len(df_features)} provided)")
    is_energy = np.asarray(df_features.index.get_level_values(-1) == energy_key)
    table = df_features.to_numpy()
    y, x = table[:, 0], table[:, 1:]
    n_e, n_f = int(is_energy.sum()), int((~is_energy).sum())
    w = np.zeros(n_e + n_f)
    w[is_energy] = kappa / np.std(y[is_energy]) / n_e
    w[~is_energy] = (1 - kappa) / np.std(y[~is_energy]) / n_f
    return x, y, w
