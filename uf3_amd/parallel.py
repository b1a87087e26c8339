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
