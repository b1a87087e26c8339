"""
Device-resident fit pipeline: frames -> feature rows -> normal-equation pieces, without the rows
ever leaving the GPU (BASELINE config 4: per-GPU X^T X accumulate, one reduce, host solve).

Replaces the reference's ``batched_to_hdf`` + ``fit_from_file`` round trip
(``uf3/representation/process.py:256-291``, ``uf3/regression/least_squares.py:355-483``): per-atom
normalisation of the energy rows and targets (``dataframe_to_tuples``, :697-700), Gram pieces of
energy and force rows, target moments for the E/F weights (``VarianceRecorder``, :19-67; the energy
moments are those of the FROZEN targets, the force moments of the raw ones, as ``fit`` does, :296-304).

Everything additive lives in ONE flat fp64 device buffer
    [ G_e (F x F) | G_f (F x F) | o_e (F) | o_f (F) | m_e (3) | m_f (3) ]
over all F columns; the frozen columns are folded out on the device when the pieces are asked for, and the
packed buffer of the unfrozen columns (2 F'^2 + 2 F' + 6 doubles) is what ``parallel.allreduce_packed`` sums
over the ranks -- no host copy before the collective.

PyTorch is used only as the owner of the device buffers and of the stream: the arithmetic around the Gram kernels
(normalisation, target moments, the fold of the frozen columns) is the library's (``uf3_fit_rows_dev``,
``uf3_fit_pack_dev``).
"""
import ctypes as C

import numpy as np

from uf3_amd import _lib


class DeviceFitAccumulator:
    def __init__(self, model, featurizer, device=None, max_atoms_per_chunk=320000, with_forces=True, first_chunk_fraction=0.25):
        """with_forces: whether force rows take part in the fit.  It is a property of the FIT, not of the frames a
        rank happens to hold: a rank with an empty shard still contributes (zero) force pieces."""
        import torch
        self.torch = torch
        self.model, self.fz = model, featurizer
        self.ctx, self.db = featurizer._dev()
        self.dev = torch.device("cuda", self.ctx.device if device is None else device)
        F = self.n_feat = self.db.n_feat
        self.with_forces = bool(with_forces)
        self.flat = torch.zeros(2 * F * F + 2 * F + 6, dtype=torch.float64, device=self.dev)
        o = 0
        self.gram_e = self.flat[o:o + F * F].view(F, F); o += F * F
        self.gram_f = self.flat[o:o + F * F].view(F, F); o += F * F
        self.ord_e = self.flat[o:o + F]; o += F
        self.ord_f = self.flat[o:o + F]; o += F
        self.m_e = self.flat[o:o + 3]; o += 3
        self.m_f = self.flat[o:o + 3]
        self.max_atoms = int(max_atoms_per_chunk)
        self.first_fraction = float(first_chunk_fraction)     # (a call's first chunk is smaller: the GPU starts sooner)
        self.n_chunks = 0
        self._counts = [0.0, 0.0]
        mask = np.asarray(model.mask)
        self._keep = torch.from_numpy(np.flatnonzero(mask) if mask.dtype == bool else mask.astype(np.int64)).to(self.dev)
        self._frozen = torch.from_numpy(np.asarray(model.col_idx, dtype=np.int64)).to(self.dev)
        self._frozen_c = torch.from_numpy(np.asarray(model.frozen_c, dtype=np.float64).reshape(-1)).to(self.dev)

    def reset(self):
        self.flat.zero_()
        self.n_chunks = 0
        self._counts = [0.0, 0.0]

    def _gram(self, x, y, gram, ordn):
        self.ctx.check(self.ctx.lib.uf3_gram_dev(self.ctx.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                                 x.shape[0], self.n_feat, self.n_feat, 1, C.c_void_p(gram.data_ptr()),
                                                 C.c_void_p(ordn.data_ptr())))

    # ---- host staging: two sets of (pinned host block, device block), used alternately ---------------------------------
    # A copy out of pageable memory is staged by the runtime and blocks the host until the stream has reached it -- behind
    # the previous chunk's kernels: packing and the GPU then take turns (round 4: 2897 against 3828 frames/s on the W/Mo
    # workload).  Here the host packs chunk k + 1 straight into a pinned block (no intermediate concatenations) while the GPU
    # works on chunk k, and ONE transfer per chunk runs on a copy stream of its own, beside the previous chunk's kernels.
    # Three events per set: `copied` (the transfer has run: the host may pack the pinned block again, the compute stream may
    # read the device block), `consumed` (the chunk's kernels have run: the copy stream may overwrite the device block).
    def _staging_set(self, which, n_atoms, n_frames):
        """Block layout, host and device alike: positions [3 A] | force targets [3 A] | per-atom energies [Fm] | atom counts
        [Fm] | species [A] (int32)."""
        torch = self.torch
        sets = self.__dict__.setdefault("_staging", [None, None])
        st = sets[which]
        if st is None or st["atoms"] < n_atoms or st["frames"] < n_frames:
            if st is not None:
                torch.cuda.synchronize(self.dev)   # (the old blocks may still be in use)
            ca, cf = max(n_atoms, self.max_atoms if st is None else st["atoms"]), max(n_frames, 64 if st is None else st["frames"])
            n = 6 * ca + 2 * cf + (ca + 1) // 2
            block = torch.empty((n,), dtype=torch.float64).pin_memory()
            st = dict(atoms=ca, frames=cf, copied=None, consumed=None, block=block, np=block.numpy(),
                      dev=torch.empty((n,), dtype=torch.float64, device=self.dev))
            sets[which] = st
        elif st["copied"] is not None:
            st["copied"].synchronize()             # (the transfer of the chunk packed into this set two chunks ago)
        return st

    def add_frames(self, frames, energies, forces=None):
        """frames: list of Atoms; energies [n]; forces: list of (N_i, 3) arrays (required when with_forces).

        Asynchronous: the host packs the next chunk into pinned staging while the GPU works on the current one (nothing
        is read back here; ``uf3_featurize_dev`` does not synchronise once the context knows its neighbour capacities).
        A capacity overflow in an earlier chunk surfaces as ``_lib.RetryError`` from a later call or from ``pieces``:
        the accumulated sums are then invalid (``fit_frames`` starts over once)."""
        torch = self.torch
        if self.with_forces and forces is None and len(frames):
            raise ValueError("this accumulator was set up with forces: pass them")
        stream = torch.cuda.current_stream(self.dev)
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(self.dev)
        prev = self.ctx.set_stream(stream.cuda_stream)
        try:
            start = 0
            while start < len(frames):           # chunks bounded by the row buffer 3*atoms*F*8 bytes
                # (the first chunk of a call a quarter of the size: the GPU starts sooner, nothing overlaps its packing)
                limit = self.max_atoms if start else max(1, int(self.max_atoms * self.first_fraction))
                stop, atoms = start, 0
                while stop < len(frames) and (stop == start or atoms + len(frames[stop]) <= limit):
                    atoms += len(frames[stop])
                    stop += 1
                nf = stop - start
                st = self._staging_set(self.n_chunks & 1, atoms, nf)
                A3 = 3 * atoms
                n_block = 2 * A3 + 2 * nf + (atoms + 1) // 2
                h = st["np"][:n_block]
                h_pos, h_yf = h[:A3].reshape(atoms, 3), h[A3:2 * A3]
                h_ye, h_cnt = h[2 * A3:2 * A3 + nf], h[2 * A3 + nf:2 * A3 + 2 * nf]
                h_z = h[2 * A3 + 2 * nf:].view(np.int32)[:atoms]
                offsets = np.zeros(nf + 1, dtype=np.int64)
                cells = np.empty((nf, 3, 3), dtype=np.float64)
                pbc = np.zeros((nf, 3), dtype=np.uint8)
                k = 0
                for i in range(nf):
                    a = frames[start + i]
                    n = len(a)
                    # (the frame's own arrays where it exposes them -- ase.Atoms and data.atoms.Atoms do --: one copy, not two)
                    pos = getattr(a, "positions", None)
                    np.copyto(h_pos[k:k + n], pos if pos is not None else a.get_positions())
                    num = getattr(a, "numbers", None)
                    np.copyto(h_z[k:k + n], num if num is not None else a.get_atomic_numbers(), casting="unsafe")
                    if self.with_forces:
                        np.copyto(h_yf[3 * k:3 * (k + n)].reshape(n, 3), np.asarray(forces[start + i]).reshape(n, 3), casting="same_kind")
                    cells[i] = a.get_cell()
                    pbc[i, :] = a.get_pbc() if hasattr(a, "get_pbc") else a.pbc
                    k += n
                    offsets[i + 1] = k
                # per-atom normalisation of the energy rows and targets (least_squares.py:697-700); the atom counts
                # are known on the host (= the sum of the composition columns)
                h_cnt[:] = np.diff(offsets)
                np.divide(np.asarray(energies[start:stop], dtype=np.float64), h_cnt, out=h_ye)
                d = st["dev"][:n_block]
                cs = self._copy_stream
                if st["consumed"] is not None:
                    cs.wait_event(st["consumed"])
                with torch.cuda.stream(cs):
                    d.copy_(st["block"][:n_block], non_blocking=True)
                st["copied"] = torch.cuda.Event()
                st["copied"].record(cs)
                stream.wait_event(st["copied"])
                d_pos, y_f = d[:A3].view(atoms, 3), (d[A3:2 * A3] if self.with_forces else None)
                y_e, counts = d[2 * A3:2 * A3 + nf], d[2 * A3 + nf:2 * A3 + 2 * nf]
                d_z = d[2 * A3 + 2 * nf:].view(torch.int32)[:atoms]
                self.add_device_batch(_lib.make_frames(offsets, cells, pbc), nf, atoms, d_pos, d_z, counts, y_e, y_f)
                st["consumed"] = torch.cuda.Event()
                st["consumed"].record(stream)
                self.n_chunks += 1
                start = stop
        finally:
            self.ctx.restore_stream(prev)

    def add_device_batch(self, frames_struct, n_frames, n_atoms, d_pos, d_z, d_counts, d_ye, d_yf=None, x_e=None, x_f=None):
        """One batch whose inputs already live in HBM (torch tensors; ``d_ye`` per-atom normalised, ``d_yf`` flat):
        featurize -> rows (in ``x_e`` / ``x_f`` if given, else fresh buffers) -> Gram pieces and moments.  Nothing
        synchronises; the context must be on the caller's stream."""
        torch = self.torch
        if x_e is None:
            x_e = torch.empty((n_frames, self.n_feat), dtype=torch.float64, device=self.dev)
        if x_f is None and self.with_forces:
            x_f = torch.empty((n_atoms * 3, self.n_feat), dtype=torch.float64, device=self.dev)
        self.fz.featurize_device(frames_struct, d_pos.data_ptr(), d_z.data_ptr(), x_e.data_ptr(),
                                 x_f.data_ptr() if self.with_forces else None)
        # per-atom normalisation of the energy rows (least_squares.py:697-700), moments of the FROZEN energies and of the
        # force targets (:296-304): one kernel of the library (uf3_fit_rows_dev), sums on the device, counts on the host
        n_yf = int(d_yf.numel()) if self.with_forces else 0
        self.ctx.check(self.ctx.lib.uf3_fit_rows_dev(
            self.ctx.handle, n_frames, self.n_feat, x_e.data_ptr(), d_counts.data_ptr(), d_ye.data_ptr(),
            d_yf.data_ptr() if self.with_forces else None, n_yf, self._frozen.data_ptr() if self._frozen.numel() else None,
            self._frozen_c.data_ptr() if self._frozen.numel() else None, int(self._frozen.numel()), self.m_e.data_ptr()))
        self._counts[0] += float(n_frames)
        self._counts[1] += float(n_yf)
        self._gram(x_e, d_ye, self.gram_e, self.ord_e)
        if self.with_forces:
            # (rows listed by species, each list multiplied on the columns of its species' blocks only)
            self.ctx.check(self.ctx.lib.uf3_gram_force_rows_dev(
                self.db.handle, x_f.data_ptr(), d_yf.data_ptr(), d_z.data_ptr(), n_atoms, self.n_feat, 1,
                self.gram_f.data_ptr(), self.ord_f.data_ptr()))

    def packed(self):
        """Device tensor [G_e | G_f | o_e | o_f | m_e | m_f] on the UNFROZEN columns (F' of them): the additive
        pieces of this rank, ready for ``parallel.allreduce_packed``.  Frozen columns are folded out on the
        Gram level: X_m^T (y - X_f c_f) = o_m - G[m, f] c_f."""
        torch = self.torch
        self.ctx.synchronize()                  # verdicts on the asynchronous featurizer calls (RetryError)
        n_keep, n_fro = int(self._keep.numel()), int(self._frozen.numel())
        out = torch.empty(2 * n_keep * n_keep + 2 * n_keep + 6, dtype=torch.float64, device=self.dev)
        prev = self.ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)     # (ordered with the caller's use of `out`)
        try:
            self.ctx.check(self.ctx.lib.uf3_fit_pack_dev(
                self.ctx.handle, self.n_feat, self.flat.data_ptr(), self._keep.data_ptr(), n_keep,
                self._frozen.data_ptr() if n_fro else None, self._frozen_c.data_ptr() if n_fro else None, n_fro,
                float(self._counts[0]), float(self._counts[1]), out.data_ptr()))
        finally:
            self.ctx.restore_stream(prev)
        return out

    def pieces(self):
        """Additive pieces of this rank as host arrays (what ``WeightedLinearModel.fit_from_pieces`` takes)."""
        from uf3_amd import parallel
        return parallel.unpack_pieces(self.packed().cpu().numpy(), int(self._keep.numel()), with_forces=self.with_forces)


def fit_frames(model, featurizer, frames, energies, forces=None, weight=0.5, reduce=True, with_forces=None,
               max_atoms_per_chunk=320000):
    """
    Featurize + accumulate on this rank's GPU, sum-reduce the packed pieces across ranks on the device (if a
    process group is initialised), solve on every rank.  ``frames`` is THIS rank's shard; ``with_forces`` must be
    the same on every rank (default: whether forces were passed) -- a rank with an empty shard passes it explicitly.
    """
    from uf3_amd import parallel
    if with_forces is None:
        with_forces = forces is not None
    acc = DeviceFitAccumulator(model, featurizer, with_forces=with_forces, max_atoms_per_chunk=max_atoms_per_chunk)
    attempt = 0
    while True:
        try:
            acc.add_frames(frames, energies, forces)
            flat = acc.packed()
            break
        except _lib.UF3Error as exc:
            # Chunks queued before the verdict arrived are still in flight with the old capacities: wait for them and
            # drop their verdicts (they would otherwise fail the next attempt, or the next user of the shared context),
            # then start over.  A capacity overflow (or the switch to the image-range launches for atoms far outside
            # their cell) is repeated until the context has converged -- capacities only grow, geometrically --; any
            # other error is the caller's.
            try:
                acc.ctx.synchronize()
            except _lib.UF3Error:
                pass
            acc.reset()
            attempt += 1
            if not isinstance(exc, _lib.RetryError) or attempt >= 16:
                raise
    n_cols = int(acc._keep.numel())
    if reduce:
        flat = parallel.allreduce_packed(flat, ctx=acc.ctx)          # (the library's own communicator when the context has one)
    pieces = parallel.unpack_pieces(flat.cpu().numpy(), n_cols, with_forces=with_forces)
    model.fit_from_pieces(pieces, weight=weight)
    return pieces
