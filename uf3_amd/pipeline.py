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

import os

import numpy as np

from uf3_amd import _lib


class DeviceFitAccumulator:
    def __init__(self, model, featurizer, device=None, max_atoms_per_chunk=320000, with_forces=True, first_chunk_fraction=0.125):
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
        self.first_fraction = float(first_chunk_fraction)     # (a call's chunks grow from this fraction of the limit, doubling: the GPU starts sooner)
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
        if getattr(self, "_fit", None):
            prev = self.ctx.set_stream(self.torch.cuda.current_stream(self.dev).cuda_stream)      # (one order with torch's fill)
            try:
                self.ctx.check(self.ctx.lib.uf3_fit_reset(self._fit))
            finally:
                self.ctx.restore_stream(prev)

    def _gram(self, x, y, gram, ordn):
        self.ctx.check(self.ctx.lib.uf3_gram_dev(self.ctx.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                                 x.shape[0], self.n_feat, self.n_feat, 1, C.c_void_p(gram.data_ptr()),
                                                 C.c_void_p(ordn.data_ptr())))

    # ---- frames given as host arrays: the library's own accumulation (uf3_fit_add), into this accumulator's flat buffer -------
    # Round 5 first staged the chunks here, through pinned torch tensors and a torch side stream; that version returned wrong
    # rows on small chunks once another accumulator had tuned the context (species flags from stale device blocks; the same
    # staging inside the library, on HIP streams of its own, does not) -- see DESIGN 3.4.  The host path is therefore ONE
    # implementation, the library's: one pointer per frame in, pinned double-buffered staging, one transfer per chunk on a copy
    # stream beside the previous chunk's kernels.
    def _native(self):
        if getattr(self, "_fit", None) is None:
            h = C.c_void_p()
            frozen = self._frozen.cpu().numpy() if self._frozen.numel() else np.zeros(0, dtype=np.int64)
            frozen_c = self._frozen_c.cpu().numpy() if self._frozen.numel() else np.zeros(0)
            self.ctx.check(self.ctx.lib.uf3_fit_create(self.db.handle, int(self.with_forces), int(self.max_atoms),
                                                       _lib._p(frozen) if len(frozen) else None, _lib._p(frozen_c) if len(frozen) else None,
                                                       len(frozen), C.byref(h)))
            self.ctx.check(self.ctx.lib.uf3_fit_use_flat(h, C.c_void_p(self.flat.data_ptr())))
            self._fit = h
        return self._fit

    def __del__(self):
        try:
            if getattr(self, "_fit", None):
                self.ctx.lib.uf3_fit_destroy(self._fit)
                self._fit = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def add_frames(self, frames, energies, forces=None):
        """frames: list of Atoms; energies [n]; forces: list of (N_i, 3) arrays (required when with_forces).

        Asynchronous: the host packs the next chunk into pinned staging while the GPU works on the current one (nothing
        is read back here; ``uf3_featurize_dev`` does not synchronise once the context knows its neighbour capacities).
        A capacity overflow in an earlier chunk surfaces as ``_lib.RetryError`` from a later call or from ``pieces``:
        the accumulated sums are then invalid (``fit_frames`` starts over once)."""
        torch = self.torch
        if self.with_forces and forces is None and len(frames):
            raise ValueError("this accumulator was set up with forces: pass them")
        fit = self._native()
        prev = self.ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        try:
            before = _fit_counts(self.ctx, fit)
            _fit_add(self.ctx, fit, frames, energies, forces, self.with_forces, self.first_fraction)
            after = _fit_counts(self.ctx, fit)
        finally:
            self.ctx.restore_stream(prev)
        self.n_chunks += after[0] - before[0]
        self._counts[0] += after[1] - before[1]
        self._counts[1] += after[2] - before[2]

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


def _fit_counts(ctx, fit):
    n, e, f = C.c_int64(), C.c_double(), C.c_double()
    ctx.check(ctx.lib.uf3_fit_info(fit, C.byref(n), C.byref(e), C.byref(f)))
    return n.value, e.value, f.value


def _fit_add(ctx, fit, frames, energies, forces, with_forces, first_fraction=None):
    """frames -> the pointer tables uf3_fit_add takes (one pointer per frame: positions, atomic numbers, force targets; the
    frames' own arrays where they expose them)"""
    n = len(frames)
    if not n:
        return
    keep = []                                   # (the arrays the pointer tables refer to, alive until the call returns)
    counts = np.array([len(a) for a in frames], dtype=np.int64)
    P, Z, Fo = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    # (per frame: the frame's own arrays where they already are C-contiguous float64 / int64 -- no copy, the address through
    # __array_interface__: 128 frames of np.ascontiguousarray + .ctypes.data were 0.6 ms in front of the first chunk, GPU idle)
    def as_is(x, dtype):
        if type(x) is np.ndarray and x.dtype == dtype and x.flags.c_contiguous:
            return x
        return np.ascontiguousarray(x, dtype=dtype)

    addr = _lib._addr
    for i, a in enumerate(frames):
        pos = getattr(a, "positions", None)
        pos = as_is(pos if pos is not None else a.get_positions(), np.float64)
        num = getattr(a, "numbers", None)
        num = as_is(num if num is not None else a.get_atomic_numbers(), np.int64)
        keep.append(pos)
        keep.append(num)
        P[i], Z[i] = addr(pos), addr(num)
        if with_forces:
            fo = as_is(forces[i], np.float64)
            if fo.size != 3 * len(a):
                raise ValueError("forces[%d] does not hold 3 components per atom" % i)
            keep.append(fo)
            Fo[i] = addr(fo)
    cells = np.ascontiguousarray([np.asarray(a.get_cell(), dtype=np.float64).reshape(3, 3) for a in frames])
    pbc = np.ascontiguousarray([np.asarray(a.get_pbc() if hasattr(a, "get_pbc") else a.pbc, dtype=np.uint8) for a in frames])
    e = np.ascontiguousarray(energies, dtype=np.float64)
    if first_fraction is not None:
        ctx.check(ctx.lib.uf3_fit_first_chunk(fit, float(first_fraction)))
    ctx.check(ctx.lib.uf3_fit_add(fit, n, _lib._p(counts), P, Z, 1, _lib._p(cells), _lib._p(pbc), _lib._p(e), Fo if with_forces else None))


class NativeFitAccumulator:
    """The same accumulation through the library alone (``uf3_fit_*``): frames go in as one pointer per frame, staging, copy
    stream, row buffers and the flat piece buffer are the library's own -- no PyTorch anywhere on this path.  ``pieces()`` /
    ``packed_host()`` return what ``WeightedLinearModel.fit_from_pieces`` takes; ``reduce=True`` sums the packed pieces over
    the ranks of the context's communicator (``parallel.native_comm``) before they come back."""

    def __init__(self, model, featurizer, max_atoms_per_chunk=320000, with_forces=True):
        self.model, self.fz = model, featurizer
        self.ctx, self.db = featurizer._dev()
        self.with_forces = bool(with_forces)
        mask = np.asarray(model.mask)
        self._keep = np.ascontiguousarray(np.flatnonzero(mask) if mask.dtype == bool else mask.astype(np.int64), dtype=np.int64)
        self._frozen = np.ascontiguousarray(model.col_idx, dtype=np.int64).reshape(-1)
        self._frozen_c = np.ascontiguousarray(model.frozen_c, dtype=np.float64).reshape(-1)
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.uf3_fit_create(self.db.handle, int(self.with_forces), int(max_atoms_per_chunk),
                                                   _lib._p(self._frozen) if len(self._frozen) else None,
                                                   _lib._p(self._frozen_c) if len(self._frozen) else None, len(self._frozen), C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.ctx.lib.uf3_fit_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def reset(self):
        self.ctx.check(self.ctx.lib.uf3_fit_reset(self.handle))

    @property
    def n_chunks(self):
        n = C.c_int64()
        self.ctx.check(self.ctx.lib.uf3_fit_info(self.handle, C.byref(n), None, None))
        return n.value

    def add_frames(self, frames, energies, forces=None):
        """frames: list of Atoms; energies [n]; forces: list of (N_i, 3) arrays (required when with_forces).  Returns once the
        last chunk is queued; a capacity overflow surfaces as ``_lib.RetryError`` later (``fit_frames`` starts over)."""
        if self.with_forces and forces is None and len(frames):
            raise ValueError("this accumulator was set up with forces: pass them")
        _fit_add(self.ctx, self.handle, frames, energies, forces, self.with_forces)

    def packed_host(self, reduce=False):
        n_keep = len(self._keep)
        out = np.empty(2 * n_keep * n_keep + 2 * n_keep + 6)
        self.ctx.check(self.ctx.lib.uf3_fit_pack(self.handle, _lib._p(self._keep) if n_keep else None, n_keep, int(bool(reduce)), _lib._p(out)))
        return out

    def pieces(self, reduce=False):
        from uf3_amd import parallel
        return parallel.unpack_pieces(self.packed_host(reduce), len(self._keep), with_forces=self.with_forces)


def fit_frames_native(model, featurizer, frames, energies, forces=None, weight=0.5, reduce=True, with_forces=None,
                      max_atoms_per_chunk=320000):
    """``fit_frames`` without PyTorch: accumulation in the library (``NativeFitAccumulator``), the ranks' pieces summed by the
    library's own RCCL communicator when the context has one (``parallel.native_comm``), solved on every rank."""
    if with_forces is None:
        with_forces = forces is not None
    acc = NativeFitAccumulator(model, featurizer, max_atoms_per_chunk=max_atoms_per_chunk, with_forces=with_forces)
    attempt = 0
    while True:
        try:
            acc.add_frames(frames, energies, forces)
            pieces = acc.pieces(reduce=reduce and acc.ctx.comm_info()[0] > 1)
            break
        except _lib.UF3Error as exc:
            try:
                acc.ctx.synchronize()
            except _lib.UF3Error:
                pass
            acc.reset()
            attempt += 1
            if not isinstance(exc, _lib.RetryError) or attempt >= 16:
                raise
    model.fit_from_pieces(pieces, weight=weight)
    return pieces


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
