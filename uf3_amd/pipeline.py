"""
Device-resident fit pipeline: frames -> feature rows -> normal-equation pieces, without the rows
ever leaving HBM (BASELINE config 4: per-GPU X^T X accumulate, one reduce, host solve).

Replaces the reference's ``batched_to_hdf`` + ``fit_from_file`` round trip
(``uf3/representation/process.py:256-291``, ``uf3/regression/least_squares.py:355-483``): per-atom
normalisation of the energy rows and targets (``dataframe_to_tuples``, :697-700), Gram pieces of
energy and force rows, target moments for the E/F weights (``VarianceRecorder``, :19-67).

PyTorch is used only as the owner of the device buffers and of the stream.
"""
import ctypes as C

import numpy as np

from uf3_amd import _lib
from uf3_amd.regression import least_squares as ls


class DeviceFitAccumulator:
    def __init__(self, model, featurizer, device=None, max_atoms_per_chunk=320000):
        import torch
        self.torch = torch
        self.model, self.fz = model, featurizer
        self.ctx, self.db = featurizer._dev()
        self.dev = torch.device("cuda", self.ctx.device if device is None else device)
        self.ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        F = self.db.n_feat
        z = lambda *s: torch.zeros(s, dtype=torch.float64, device=self.dev)  # noqa: E731
        self.gram_e, self.gram_f, self.ord_e, self.ord_f = z(F, F), z(F, F), z(F), z(F)
        self.m_e, self.m_f = np.zeros(3), np.zeros(3)
        self.max_atoms = int(max_atoms_per_chunk)
        self.n_feat = F
        self.n_el = len(model.bspline_config.element_list)
        self.with_forces = False

    def _gram(self, x, y, gram, ordn):
        rows = x.shape[0]
        self.ctx.check(self.ctx.lib.uf3_gram_dev(self.ctx.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                                 rows, self.n_feat, self.n_feat, 1, C.c_void_p(gram.data_ptr()),
                                                 C.c_void_p(ordn.data_ptr())))

    def add_frames(self, frames, energies, forces=None):
        """frames: list of Atoms; energies [n]; forces: list of (N_i, 3) arrays or None."""
        torch = self.torch
        start = 0
        while start < len(frames):           # chunks bounded by the row buffer 3*atoms*F*8 bytes
            stop, atoms = start, 0
            while stop < len(frames) and (stop == start or atoms + len(frames[stop]) <= self.max_atoms):
                atoms += len(frames[stop])
                stop += 1
            chunk = frames[start:stop]
            batch = _lib.FrameBatch(chunk)
            d_pos = torch.from_numpy(batch.pos).to(self.dev)
            d_z = torch.from_numpy(batch.z).to(self.dev)
            x_e = torch.empty((batch.n_frames, self.n_feat), dtype=torch.float64, device=self.dev)
            x_f = (torch.empty((batch.n_atoms * 3, self.n_feat), dtype=torch.float64, device=self.dev)
                   if forces is not None else None)
            self.fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), x_e.data_ptr(),
                                     x_f.data_ptr() if x_f is not None else None)
            # per-atom normalisation of the energy rows and targets (least_squares.py:697-700); the atom counts are
            # known on the host (= the sum of the composition columns), so nothing is read back inside the loop and
            # the host packs the next chunk while this one is still on the GPU (one stream: buffers handed back to
            # torch's allocator here are not reused before the kernels queued above have run)
            counts = np.diff(batch.offsets).astype(np.float64)
            y_e_host = np.asarray(energies[start:stop], dtype=np.float64) / counts
            x_e = (x_e / torch.from_numpy(counts).to(self.dev)[:, None]).contiguous()
            y_e = torch.from_numpy(y_e_host).to(self.dev)
            self._gram(x_e, y_e, self.gram_e, self.ord_e)
            self.m_e += ls.moments(y_e_host)
            if forces is not None:
                y_host = np.concatenate([np.asarray(f, dtype=np.float64).reshape(-1, 3) for f in forces[start:stop]]).reshape(-1)
                y_f = torch.from_numpy(y_host).to(self.dev)
                self._gram(x_f, y_f, self.gram_f, self.ord_f)
                self.m_f += ls.moments(y_host)
                self.with_forces = True
            start = stop

    def pieces(self):
        """Additive pieces on the unfrozen columns (what ``parallel.allreduce_pieces`` sums)."""
        model = self.model
        mask, col_idx, frozen_c = model.mask, model.col_idx, np.asarray(model.frozen_c, dtype=float)

        def reduce(gram, ordn):
            g, o = gram.cpu().numpy(), ordn.cpu().numpy()
            # freeze_columns on the Gram level: X_m^T (y - X_f c_f) = o_m - G[m, f] c_f
            return g[np.ix_(mask, mask)], o[mask] - g[np.ix_(mask, col_idx)] @ frozen_c

        out = dict(m_e=self.m_e.copy())
        out["gram_e"], out["ord_e"] = reduce(self.gram_e, self.ord_e)
        if self.with_forces:
            out["m_f"] = self.m_f.copy()
            out["gram_f"], out["ord_f"] = reduce(self.gram_f, self.ord_f)
        return out


def fit_frames(model, featurizer, frames, energies, forces=None, weight=0.5, reduce=True):
    """
    Featurize + accumulate on this rank's GPU, sum-reduce the pieces across ranks (if a process group
    is initialised), solve on every rank.  ``frames`` is THIS rank's shard.
    """
    from uf3_amd import parallel
    acc = DeviceFitAccumulator(model, featurizer)
    acc.add_frames(frames, energies, forces)
    pieces = acc.pieces()
    if reduce:
        pieces = parallel.allreduce_pieces(pieces, model.n_feats - len(model.col_idx))
    model.fit_from_pieces(pieces, weight=weight)
    return pieces
