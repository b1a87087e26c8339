"""
Frame sharding across the GPUs of a node and the single sum-reduce of the normal
equations.

The reference fans ``BasisFeaturizer.evaluate`` out over DataFrame chunks with a
process pool / dask (``uf3/representation/process.py:196-254``,
``uf3/util/parallel.py:167-251``) and adds the per-table Gram pieces serially
(``least_squares.py:391-412``).  Here frames are sharded one contiguous block per
rank (one process per GPU), every rank featurizes and accumulates its own
``{G_e, G_f, o_e, o_f}`` + target moments, and ONE ``all_reduce(SUM)`` over a packed
fp64 buffer (2F'^2 + 2F' + 6 doubles; RCCL over xGMI when the backend is "nccl",
gloo in the CPU tests) yields the global pieces.  Featurize-only work needs no
collective at all.
"""
import numpy as np


def shard_range(n_items, rank, world_size):
    """Contiguous block [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def frame_work(atoms, r_max3=None):
    """Relative featurizer work of a frame ~ atoms x triplets per atom (shards balanced by sum N*T).
    T grows with the square of the 3-body neighbour count q = 4/3 pi r^3 rho; without a volume (clusters) or a
    3-body range the atom count is all there is."""
    n = len(atoms)
    if not n:
        return 0.0
    vol = 0.0
    try:
        cell = np.asarray(atoms.get_cell(), dtype=float).reshape(3, 3)
        if np.all(np.asarray(atoms.get_pbc() if hasattr(atoms, "get_pbc") else atoms.pbc)):
            vol = abs(float(np.linalg.det(cell)))
    except Exception:  # noqa: BLE001 - an estimate only
        vol = 0.0
    if not r_max3 or vol <= 0.0:
        return float(n)
    q = 4.18879 * r_max3 ** 3 * n / vol
    return float(n) * (1.0 + 0.5 * q * q)


def shard_balanced(weights, rank, world_size):
    """Contiguous block [lo, hi) of rank such that the blocks' summed weights are as even as contiguity allows:
    block r ends at the first item where the running weight reaches (r + 1) / world of the total."""
    w = np.asarray(weights, dtype=float)
    n = len(w)
    if n == 0 or w.sum() <= 0:
        return shard_range(n, rank, world_size)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world_size):
        target = cum[-1] * r / world_size
        k = int(np.searchsorted(cum, target, side="left"))
        if k > 0 and (target - cum[k - 1]) < (cum[k] - target):      # nearest boundary
            k -= 1
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return cuts[rank], cuts[rank + 1]


PIECE_KEYS = ("gram_e", "gram_f", "ord_e", "ord_f", "m_e", "m_f")


def pack_pieces(pieces, n_cols):
    """dict of additive pieces -> one flat fp64 vector (missing force pieces = zeros)."""
    shapes = dict(gram_e=(n_cols, n_cols), gram_f=(n_cols, n_cols), ord_e=(n_cols,), ord_f=(n_cols,),
                  m_e=(3,), m_f=(3,))
    return np.concatenate([np.asarray(pieces.get(k, np.zeros(shapes[k])), dtype=np.float64).reshape(-1)
                           for k in PIECE_KEYS])


def unpack_pieces(buf, n_cols, with_forces=True):
    buf = np.asarray(buf, dtype=np.float64)
    n2 = n_cols * n_cols
    out, o = {}, 0
    for k, size, shape in (("gram_e", n2, (n_cols, n_cols)), ("gram_f", n2, (n_cols, n_cols)),
                           ("ord_e", n_cols, (n_cols,)), ("ord_f", n_cols, (n_cols,)),
                           ("m_e", 3, (3,)), ("m_f", 3, (3,))):
        out[k] = buf[o:o + size].reshape(shape).copy()
        o += size
    if not with_forces:
        for k in ("gram_f", "ord_f", "m_f"):
            out.pop(k)
    return out


def native_comm(ctx, rank=None, world_size=None, id_path=None):
    """
    Bring up the library's own RCCL communicator on ``ctx`` (``uf3_comm_init`` -- librccl bound behind the C ABI, no
    torch.distributed in the data path).  The 128-byte id rank 0 draws reaches the other ranks through torch.distributed's
    default process group when one is up (any backend: it is a host-side broadcast), else through the file ``id_path``
    (rank 0 writes it, the others wait for it) -- the launcher's RANK / WORLD_SIZE tell who is who.  Returns ``ctx``.
    """
    import os
    import time
    rank = int(os.environ.get("RANK", 0)) if rank is None else int(rank)
    world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else int(world_size)
    uid = None
    dist = None
    try:
        import torch.distributed as dist_mod
        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
    except ImportError:
        pass
    if world_size == 1:
        uid = ctx.comm_unique_id()
    elif dist is not None and id_path is None:
        box = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    else:
        if id_path is None:
            raise ValueError("native_comm: more than one rank needs a process group or an id_path to pass the id")
        if rank == 0:
            uid = ctx.comm_unique_id()
            with open(id_path + ".tmp", "wb") as f:
                f.write(uid)
            os.replace(id_path + ".tmp", id_path)
        else:
            t_end = time.time() + 120
            while not os.path.exists(id_path):
                if time.time() > t_end:
                    raise TimeoutError(f"native_comm: {id_path} did not appear")
                time.sleep(0.01)
            with open(id_path, "rb") as f:
                uid = f.read()
    ctx.comm_init(world_size, rank, uid)
    return ctx


def allreduce_packed(flat, force=None, ctx=None):
    """
    Sum a packed piece buffer (torch tensor, device or host) over all ranks of the default process group, in place
    where the backend allows: with "nccl" (= RCCL over xGMI) the DEVICE buffer goes straight into the collective;
    with "gloo" (CPU tests) a host copy is reduced.  No-op without a process group, and -- unless ``force`` (default:
    the environment variable UF3_FORCE_COLLECTIVE) -- in a group of one rank: forcing it runs the collective itself on
    a one-GPU box (a sum over one rank: the buffer is unchanged).  ``ctx`` with a communicator of its own
    (``native_comm``): the collective is the library's (``uf3_allreduce_sum_f64``), not torch.distributed's.
    """
    import os
    import torch.distributed as dist
    if force is None:
        force = bool(os.environ.get("UF3_FORCE_COLLECTIVE"))
    if ctx is not None and ctx.comm_info()[0] > 0:
        # the library's own communicator (native_comm): the device buffer through uf3_allreduce_sum_f64, ordered on the
        # caller's stream like the kernels that filled it
        import torch
        if not flat.is_cuda:
            raise ValueError("the library's communicator reduces device buffers")
        if ctx.comm_info()[0] == 1 and not force:
            return flat
        prev = ctx.set_stream(torch.cuda.current_stream(flat.device).cuda_stream)
        try:
            ctx.allreduce_sum(flat.data_ptr(), flat.numel())
        finally:
            ctx.restore_stream(prev)
        return flat
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return flat
    if dist.get_backend() == "nccl":
        if not flat.is_cuda:
            raise ValueError("the nccl backend reduces device buffers")
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat
    host = flat.cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM)
    return host


def allreduce_pieces(pieces, n_cols, device=None, with_forces=None):
    """
    Host-dict flavour of ``allreduce_packed``.  ``with_forces`` is a property of the fit and must be the same on
    every rank (default: decided AFTER the reduction, from the global number of force targets m_f[0], so that a
    rank whose shard had no forces -- or no frames -- ends up with the same system as the others).
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return pieces
    flat = torch.from_numpy(pack_pieces(pieces, n_cols))
    if dist.get_backend() == "nccl":
        flat = flat.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))
    flat = allreduce_packed(flat).cpu().numpy()
    if with_forces is None:
        with_forces = unpack_pieces(flat, n_cols)["m_f"][0] > 0
    return unpack_pieces(flat, n_cols, with_forces=with_forces)


def sharded_fit(model, featurizer, frames, energies, forces=None, weight=0.5, balance=True):
    """
    Data-parallel ``WeightedLinearModel`` fit over ALL frames: this rank takes a contiguous block (balanced by the
    estimated work sum N*T when ``balance``), featurizes it and accumulates the Gram pieces on its GPU without the
    rows leaving it (``pipeline.DeviceFitAccumulator``), all ranks sum-reduce the packed device buffer once and every
    rank solves the same small system.  Energy rows / targets are per-atom normalised as in the reference's
    from-file path (least_squares.py:697-700).
    """
    import torch.distributed as dist
    from uf3_amd import pipeline
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    if balance and world > 1:
        basis = model.bspline_config
        r3 = 0.0
        for trio in basis.interactions_map.get(3, []) if basis.degree > 2 else []:
            r3 = max(r3, float(np.max(np.asarray(basis.r_max_map[trio])[:2])))
        lo, hi = shard_balanced([frame_work(a, r3) for a in frames], rank, world)
    else:
        lo, hi = shard_range(len(frames), rank, world)
    return pipeline.fit_frames(model, featurizer, frames[lo:hi], energies[lo:hi],
                               None if forces is None else forces[lo:hi], weight=weight,
                               with_forces=forces is not None)


def sharded_evaluate(calculator, atoms, forces=True, virial=False, device=None):
    """
    ONE large frame decomposed over the ranks (the reference's calculator is a
    single process, calculator.py:124-153).  Each rank holds the whole frame's positions (28 B per atom) and
    evaluates the triplets CENTRED in its contiguous index block once each (``uf3_eval_centres``: the block's pair
    terms, its triplets, and what they put on the block's halo -- no halo bookkeeping on the host) and one
    ``all_reduce(SUM)`` over [energy | dE/d(strain) (6) | forces (3N)] -- 1.2 MB at 50 k atoms, RCCL over
    xGMI under "nccl" -- gives every rank the full result.  Returns (energy, forces or None, virial or None).
    """
    import torch
    import torch.distributed as dist
    n = len(atoms)
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    lo, hi = shard_range(n, rank, world)
    # (a block of centres: every triplet once, what it puts on atoms of other blocks rides in the same reduce; calculators
    # without that entry -- the stand-ins of the CPU tests -- give the gather shares, which add up the same way)
    share = getattr(calculator, "evaluate_centre_range", None) or calculator.evaluate_atom_range
    e, f, v = share(atoms, lo, hi, forces=forces, virial=virial)
    if not on:
        return e, f, v
    flat = np.concatenate([[e], np.zeros(6) if v is None else v, np.zeros(0) if f is None else f.reshape(-1)])
    t = torch.from_numpy(flat)
    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        flat = t.cpu().numpy()
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        flat = t.numpy()
    return (float(flat[0]), flat[7:].reshape(n, 3) if forces else None, flat[1:7].copy() if virial else None)


class ShardedEvaluator:
    """
    The MD-loop form of ``sharded_evaluate``: ONE large frame decomposed over the ranks, everything a step touches resident on
    the device (what ``bench.py --mode eval`` times at N > 1 is this class, not a private copy).

    Per rank: the whole frame's positions and species in HBM (28 B per atom), one flat fp64 device buffer
    ``[forces (3N) | energy | dE/d(strain) (6)]``; a step evaluates the triplets CENTRED in the rank's contiguous block of atoms
    (``uf3_eval_centres_dev``, on the evaluator's MD route -- neighbour lists kept with a skin -- when ``md_skin`` > 0) straight
    into that buffer and the ranks sum it with ONE ``all_reduce(SUM)`` (RCCL over xGMI under "nccl", or the library's own
    communicator with ``native=True``); one rank runs the whole-frame evaluator (``uf3_eval_virial_dev``), no collective.
    Positions move on the device (``positions`` is the device tensor: update it in place) or come from the host
    (``set_positions``); results come to the host only when asked for (``result``).  The reference's calculator is a single
    process (uf3/forcefield/calculator.py:124-153).

    A calculator without device coefficient tables -- the table stand-ins of the CPU tests -- takes the host route: its
    ``evaluate_centre_range`` / ``evaluate_atom_range`` shares fill a host buffer of the same layout, reduced over gloo.
    """

    def __init__(self, calculator, atoms, device=None, md_skin=0.0, native=False, force_collective=None):
        import os
        import torch
        import torch.distributed as dist
        self.calculator = calculator
        self.atoms = atoms
        self.n = len(atoms)
        on = dist.is_available() and dist.is_initialized()
        self.dist = dist if on else None
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
        if force_collective is None:
            force_collective = bool(os.environ.get("UF3_FORCE_COLLECTIVE"))
        # (forced: the decomposed route + its collective in a group of one rank -- what a one-GPU box can exercise of the N > 1 path)
        self.decomposed = self.world > 1 or (on and force_collective)
        self.lo, self.hi = shard_range(self.n, self.rank, self.world)
        self.native = bool(native)
        self.device_route = hasattr(calculator, "_pc")
        self.steps = 0
        if not self.device_route:
            self.flat = torch.zeros(3 * self.n + 7, dtype=torch.float64)
            self._pos = np.array(atoms.get_positions(), dtype=np.float64)
            return
        from uf3_amd import _lib
        self._lib = _lib
        dev_index = calculator.device if device is None else device
        self.ctx = _lib.get_context(dev_index)
        self.db = _lib.device_basis(calculator.bspline_config, self.ctx)
        self.batch = _lib.FrameBatch([atoms])
        self.dev = torch.device("cuda", self.ctx.device if hasattr(self.ctx, "device") else (dev_index or 0))
        self.positions = torch.from_numpy(self.batch.pos).to(self.dev)          # [N, 3] (device; update in place)
        self._z = torch.from_numpy(self.batch.z).to(self.dev)
        self.flat = torch.zeros(3 * self.n + 7, dtype=torch.float64, device=self.dev)
        self.md_skin = float(md_skin)
        self._args = self._out = None
        if self.md_skin > 0:
            self.ctx.md_skin(self.md_skin)
        if self.native and self.ctx.comm_info()[0] <= 0:
            native_comm(self.ctx, self.rank, self.world)

    def set_positions(self, positions):
        """New positions from the host ([N, 3]); device callers write ``self.positions`` in place instead."""
        import torch
        positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(self.n, 3)
        if self.device_route:
            self.positions.copy_(torch.from_numpy(positions))
        else:
            self._pos = positions

    def step(self):
        """One evaluation of the current positions into ``self.flat`` (summed over the ranks); asynchronous on the device route."""
        import ctypes as C
        import torch
        self.steps += 1
        if not self.device_route:
            atoms = self.atoms.copy() if hasattr(self.atoms, "copy") else self.atoms
            if hasattr(atoms, "set_positions"):
                atoms.set_positions(self._pos)
            share = getattr(self.calculator, "evaluate_centre_range", None) or self.calculator.evaluate_atom_range
            e, f, v = share(atoms, self.lo, self.hi, forces=True, virial=True)
            host = np.concatenate([np.asarray(f, dtype=np.float64).reshape(-1), [e], np.zeros(6) if v is None else v])
            self.flat = torch.from_numpy(host)
            if self.dist is not None and self.decomposed:
                self.dist.all_reduce(self.flat, op=self.dist.ReduceOp.SUM)
            return self
        ctx = self.ctx
        cur = torch.cuda.current_stream(self.dev).cuda_stream
        if getattr(ctx, "_stream", None) != cur:               # (the _dev entries run on the stream the context is bound to)
            ctx.set_stream(cur)
        if self._args is None:                                  # (every pointer of a step is fixed: positions move in place)
            n, base = self.n, self.flat.data_ptr()
            self._args = (self.db.handle, C.byref(self.batch.struct), C.c_void_p(self.positions.data_ptr()),
                          C.c_void_p(self._z.data_ptr())) + tuple(self.calculator._pc)
            self._out = (C.c_void_p(base + 8 * 3 * n), C.c_void_p(base), C.c_void_p(base + 8 * (3 * n + 1)))    # energy | forces | strain
        if not self.decomposed:
            ctx.check(ctx.lib.uf3_eval_virial_dev(*self._args, *self._out))
        else:
            # (uf3_eval_centres_dev zeroes every force row itself and overwrites energy / strain derivative: nothing to clear)
            ctx.check(ctx.lib.uf3_eval_centres_dev(*self._args, self.lo, self.hi, *self._out))
            if self.native:
                ctx.allreduce_sum(self.flat.data_ptr(), self.flat.numel())
            else:
                self.dist.all_reduce(self.flat, op=self.dist.ReduceOp.SUM)
        return self

    def result(self):
        """(energy, forces [N, 3], dE/d(strain) [6]) of the last step, on the host (synchronises)."""
        host = self.flat.cpu().numpy() if hasattr(self.flat, "cpu") else np.asarray(self.flat)
        n = self.n
        return float(host[3 * n]), host[:3 * n].reshape(n, 3).copy(), host[3 * n + 1:].copy()

    def host_positions(self):
        return self.positions.cpu().numpy() if self.device_route else self._pos.copy()

    def close(self):
        """Leave the evaluator's MD route (the context is shared with other callers)."""
        if self.device_route and self.md_skin > 0:
            self.ctx.md_skin(0.0)


class DeviceFeatureBatch:
    """A rank's block of frames with everything the featurizer touches resident in HBM: positions | species in, energy rows
    ``x_e [frames][F]`` and force rows ``x_f [atoms][3][ld]`` out (``run()`` = one ``uf3_featurize_ld_dev`` on torch's current stream)."""

    def __init__(self, featurizer, frames, device=None, ld=0):
        import torch
        from uf3_amd import _lib
        self.featurizer = featurizer
        self.frames = list(frames)
        self.ctx, self.db = featurizer._dev()
        self.batch = _lib.FrameBatch(self.frames)
        self.F = self.db.n_feat
        dev_index = self.ctx.device if device is None else device
        self.dev = torch.device("cuda", dev_index or 0)
        self.ld = self.F if ld == 0 else (featurizer.aligned_ld(self.F) if ld < 0 else max(self.F, int(ld)))
        self.pos = torch.from_numpy(self.batch.pos).to(self.dev)
        self.z = torch.from_numpy(self.batch.z).to(self.dev)
        self.x_e = torch.empty((len(self.frames), self.F), dtype=torch.float64, device=self.dev)
        self.x_f_full = torch.empty((self.batch.n_atoms, 3, self.ld), dtype=torch.float64, device=self.dev)
        self.x_f = self.x_f_full[:, :, :self.F]
        self.offsets = self.batch.offsets

    def run(self):
        import torch
        cur = torch.cuda.current_stream(self.dev).cuda_stream
        if getattr(self.ctx, "_stream", None) != cur:          # (the _dev entries run on the stream the context is bound to)
            self.ctx.set_stream(cur)
        self.featurizer.featurize_device(self.batch.struct, self.pos.data_ptr(), self.z.data_ptr(), self.x_e.data_ptr(),
                                         self.x_f_full.data_ptr(), ld=self.ld)
        return self


def featurize_sharded(featurizer, frames, n_frames=None, device=None, ld=0, balance=False, rank=None, world_size=None):
    """
    The featurize-only fan-out (reference: ``BasisFeaturizer.evaluate_parallel`` over DataFrame chunks,
    uf3/representation/process.py:196-254): this rank's contiguous block of the frames -- ``frames`` a list, or a callable
    ``frames(i)`` over ``n_frames`` global indices so that a rank materialises only its own block -- as a ``DeviceFeatureBatch``
    (rows resident in HBM); the data path has no collective.  ``balance``: blocks of even estimated work sum N*T instead of even
    frame counts (needs the list).  Returns (batch, (lo, hi)): call ``batch.run()`` per pass.
    """
    if rank is None or world_size is None:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        rank, world_size = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    if callable(frames):
        if n_frames is None:
            raise ValueError("featurize_sharded: a frame generator needs n_frames")
        lo, hi = shard_range(n_frames, rank, world_size)
        mine = [frames(i) for i in range(lo, hi)]
    else:
        frames = list(frames)
        if balance and world_size > 1:
            basis = featurizer.bspline_config
            r3 = 0.0
            for trio in basis.interactions_map.get(3, []) if basis.degree > 2 else []:
                r3 = max(r3, float(np.max(np.asarray(basis.r_max_map[trio])[:2])))
            lo, hi = shard_balanced([frame_work(a, r3) for a in frames], rank, world_size)
        else:
            lo, hi = shard_range(len(frames), rank, world_size)
        mine = frames[lo:hi]
    return DeviceFeatureBatch(featurizer, mine, device=device, ld=ld), (lo, hi)
