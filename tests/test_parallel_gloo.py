"""world_size-2 gloo run of the sharded normal-equation reduce (the N>1 path, on CPU)."""
import json
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from _util import GOLDEN, basis_from_meta


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from uf3_amd import parallel
    from uf3_amd.regression import least_squares as ls
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(GOLDEN, "fit_case.npz"))
    basis = basis_from_meta(json.loads(str(d["meta"])))
    model = ls.WeightedLinearModel(basis, regularizer=d["regularizer"])
    mask, fc, ci = model.mask, model.frozen_c, model.col_idx
    lo_e, hi_e = parallel.shard_range(len(d["y_e"]), rank, world)
    lo_f, hi_f = parallel.shard_range(len(d["y_f"]), rank, world)
    # the per-shard Gram pieces come from the GPU in production; here NumPy stands in for the
    # device so that the packing / reduce / solve path is exercised on CPU
    xe, ye = ls.freeze_columns(d["x_e"][lo_e:hi_e], d["y_e"][lo_e:hi_e], mask, fc, ci)
    xf, yf = ls.freeze_columns(d["x_f"][lo_f:hi_f], d["y_f"][lo_f:hi_f], mask, fc, ci)
    pieces = dict(gram_e=xe.T @ xe, ord_e=xe.T @ ye, gram_f=xf.T @ xf, ord_f=xf.T @ yf,
                  m_e=ls.moments(d["y_e"][lo_e:hi_e]), m_f=ls.moments(d["y_f"][lo_f:hi_f]))
    n_cols = model.n_feats - len(ci)
    total = parallel.allreduce_pieces(pieces, n_cols)
    model.fit_from_pieces(total, weight=float(d["kappa"][0]))
    np.save(os.path.join(out_dir, f"coeff_{rank}.npy"), model.coefficients)
    np.save(os.path.join(out_dir, f"gram_{rank}.npy"), total["gram_f"])
    dist.destroy_process_group()


def test_two_rank_reduce_reproduces_single_process_fit(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d = np.load(os.path.join(GOLDEN, "fit_case.npz"))
    c0, c1 = np.load(tmp_path / "coeff_0.npy"), np.load(tmp_path / "coeff_1.npy")
    assert np.array_equal(c0, c1)
    assert np.allclose(c0, d["coefficients"], rtol=1e-6, atol=1e-8)
    assert np.allclose(np.load(tmp_path / "gram_0.npy"), d["gram_f"], rtol=1e-9, atol=1e-9)


class _TableCalculator:
    """per-atom shares from a table (the GPU computes them in production: tests/test_gpu_parity.py checks that
    uf3_eval_atoms' shares add up); exercises the block assignment, packing and the reduce"""

    def __init__(self, e_atom, f_atom, v_atom):
        self.e, self.f, self.v = e_atom, f_atom, v_atom

    def evaluate_atom_range(self, atoms, lo, hi, forces=True, virial=False):
        f = np.zeros_like(self.f)
        f[lo:hi] = self.f[lo:hi]
        return float(self.e[lo:hi].sum()), (f if forces else None), (self.v[lo:hi].sum(0) if virial else None)


def _eval_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from uf3_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(3)
    n = 101
    calc = _TableCalculator(rng.normal(size=n), rng.normal(size=(n, 3)), rng.normal(size=(n, 6)))
    e, f, v = parallel.sharded_evaluate(calc, [None] * n, forces=True, virial=True)
    e2, f2, v2 = parallel.sharded_evaluate(calc, [None] * n, forces=False, virial=False)
    assert f2 is None and v2 is None and abs(e2 - e) < 1e-12
    np.savez(os.path.join(out_dir, f"eval_{rank}.npz"), e=e, f=f, v=v, e_ref=calc.e.sum(), f_ref=calc.f,
             v_ref=calc.v.sum(0))
    dist.destroy_process_group()


class _MovingTableCalculator(_TableCalculator):
    """shares that depend on the positions (a harmonic well per atom), so that the stand-in sees what ShardedEvaluator hands it"""

    def evaluate_atom_range(self, atoms, lo, hi, forces=True, virial=False):
        x = atoms.get_positions()
        f = np.zeros_like(x)
        f[lo:hi] = -x[lo:hi] * self.f[lo:hi]
        e = 0.5 * float((x[lo:hi] ** 2 * self.f[lo:hi]).sum())
        return e, f, self.v[lo:hi].sum(0)


class _Frame:
    def __init__(self, x):
        self.x = np.array(x)

    def __len__(self):
        return len(self.x)

    def get_positions(self):
        return self.x

    def set_positions(self, x):
        self.x = np.array(x)

    def copy(self):
        return _Frame(self.x)


def _sharded_evaluator_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from uf3_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(3)
    n = 101
    k = np.abs(rng.normal(size=(n, 3))) + 0.5
    calc = _MovingTableCalculator(None, k, rng.normal(size=(n, 6)))
    x = rng.normal(size=(n, 3))
    ev = parallel.ShardedEvaluator(calc, _Frame(x))
    assert (ev.rank, ev.world, ev.decomposed) == (rank, world, True) and not ev.device_route
    out = []
    for step in range(3):                                    # an MD loop: positions from the host, one reduce per step
        ev.set_positions(x)
        e, f, v = ev.step().result()
        out.append((e, f, v))
        x = x + 0.01 * f
    np.savez(os.path.join(out_dir, f"sev_{rank}.npz"), e=[o[0] for o in out], f=np.stack([o[1] for o in out]),
             v=np.stack([o[2] for o in out]), k=k, v_ref=calc.v.sum(0), x_last=ev.host_positions())
    dist.destroy_process_group()


def test_two_rank_sharded_evaluator_md_loop(tmp_path):
    """parallel.ShardedEvaluator (what bench.py --mode eval runs on the GPUs) with the CPU stand-in calculator: block of centres
    per rank, flat buffer [forces | energy | strain derivative], one all_reduce per step, positions updated between steps"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_evaluator_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d0, d1 = np.load(tmp_path / "sev_0.npz"), np.load(tmp_path / "sev_1.npz")
    assert np.array_equal(d0["f"], d1["f"]) and np.array_equal(d0["e"], d1["e"])       # every rank holds the full result
    rng = np.random.default_rng(3)
    n = 101
    k = np.abs(rng.normal(size=(n, 3))) + 0.5
    rng.normal(size=(n, 6))
    x = rng.normal(size=(n, 3))
    for step in range(3):
        f = -x * k
        assert np.allclose(d0["f"][step], f, rtol=0, atol=1e-14) and abs(d0["e"][step] - 0.5 * (x * x * k).sum()) < 1e-10
        assert np.allclose(d0["v"][step], d0["v_ref"], rtol=1e-12, atol=1e-12)
        x = x + 0.01 * f


def test_two_rank_decomposed_frame_evaluation(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_eval_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        d = np.load(tmp_path / f"eval_{rank}.npz")
        assert abs(d["e"] - d["e_ref"]) < 1e-12
        assert np.array_equal(d["f"], d["f_ref"])                # each row comes from exactly one rank (+ zeros)
        assert np.allclose(d["v"], d["v_ref"], rtol=1e-12, atol=1e-12)


def _uneven_worker(rank, world, port, out_dir):
    """rank 1 holds no force rows (as a rank with an empty shard would): every rank must still solve the E + F system"""
    import torch
    import torch.distributed as dist
    from uf3_amd import parallel
    from uf3_amd.regression import least_squares as ls
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(GOLDEN, "fit_case.npz"))
    basis = basis_from_meta(json.loads(str(d["meta"])))
    model = ls.WeightedLinearModel(basis, regularizer=d["regularizer"])
    mask, fc, ci = model.mask, model.frozen_c, model.col_idx
    n_cols = model.n_feats - len(ci)
    pieces = {}
    if rank == 0:
        pieces = model_pieces_numpy(ls, d, mask, fc, ci, slice(None), slice(None))
    else:
        z = np.zeros
        pieces = dict(gram_e=z((n_cols, n_cols)), ord_e=z(n_cols), m_e=z(3))        # no force keys at all
    total = parallel.allreduce_pieces(pieces, n_cols)
    assert "gram_f" in total
    model.fit_from_pieces(total, weight=float(d["kappa"][0]))
    # the packed flavour (what pipeline.fit_frames reduces): a host tensor under gloo
    flat = parallel.allreduce_packed(torch.from_numpy(parallel.pack_pieces(pieces, n_cols)))
    assert np.allclose(parallel.unpack_pieces(flat.numpy(), n_cols)["gram_f"], total["gram_f"])
    np.save(os.path.join(out_dir, f"uneven_{rank}.npy"), model.coefficients)
    dist.destroy_process_group()


def model_pieces_numpy(ls, d, mask, fc, ci, rows_e, rows_f):
    xe, ye = ls.freeze_columns(d["x_e"][rows_e], d["y_e"][rows_e], mask, fc, ci)
    xf, yf = ls.freeze_columns(d["x_f"][rows_f], d["y_f"][rows_f], mask, fc, ci)
    return dict(gram_e=xe.T @ xe, ord_e=xe.T @ ye, gram_f=xf.T @ xf, ord_f=xf.T @ yf,
                m_e=ls.moments(ye), m_f=ls.moments(d["y_f"][rows_f]))


def test_a_rank_without_force_rows_solves_the_same_system(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_uneven_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d = np.load(os.path.join(GOLDEN, "fit_case.npz"))
    c0, c1 = np.load(tmp_path / "uneven_0.npy"), np.load(tmp_path / "uneven_1.npy")
    assert np.array_equal(c0, c1)
    assert np.allclose(c0, d["coefficients"], rtol=1e-6, atol=1e-8)


def test_balanced_shards():
    from uf3_amd import parallel
    from uf3_amd.data.atoms import Atoms
    w = [1, 1, 1, 1, 8, 1, 1, 1, 1]
    parts = [parallel.shard_balanced(w, r, 2) for r in range(2)]
    assert parts[0][0] == 0 and parts[0][1] == parts[1][0] and parts[1][1] == len(w)
    assert abs(sum(w[parts[0][0]:parts[0][1]]) - sum(w[parts[1][0]:parts[1][1]])) <= 8
    # every item goes to exactly one rank, blocks are contiguous and in rank order, empty weights fall back to counts
    for world in (1, 3, 4, 8):
        cuts = [parallel.shard_balanced(w, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(w) and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    assert parallel.shard_balanced([0, 0, 0, 0], 1, 2) == parallel.shard_range(4, 1, 2)
    big = Atoms(numbers=[74] * 2000, positions=np.zeros((2000, 3)), cell=np.eye(3) * 31.65, pbc=True)
    small = Atoms(numbers=[74] * 250, positions=np.zeros((250, 3)), cell=np.eye(3) * 15.825, pbc=True)
    assert parallel.frame_work(big, 3.5) == pytest.approx(8 * parallel.frame_work(small, 3.5))     # same density
    assert parallel.frame_work(small) == 250.0


def test_shard_range_and_packing():
    from uf3_amd import parallel
    parts = [parallel.shard_range(10, r, 4) for r in range(4)]
    assert parts == [(0, 3), (3, 6), (6, 8), (8, 10)]
    rng = np.random.default_rng(0)
    p = dict(gram_e=rng.random((4, 4)), gram_f=rng.random((4, 4)), ord_e=rng.random(4), ord_f=rng.random(4),
             m_e=rng.random(3), m_f=rng.random(3))
    q = parallel.unpack_pieces(parallel.pack_pieces(p, 4), 4)
    assert all(np.array_equal(p[k], q[k]) for k in p)


def test_bench_launcher_starts_n_ranks_or_refuses(capsys):
    """`python bench.py --gpus N` without torch.distributed.run around it is its own launcher (the shape of the driver's N = 1
    command): N ranks through torch.distributed.run on 127.0.0.1, or a loud refusal -- never a one-rank run labelled N."""
    import subprocess
    import sys
    import bench
    seen = {}

    def runner(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--mode", "fit"]
    assert bench.spawn_ranks(4, argv, device_count=8, runner=runner, port=29517) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-len(argv):] == argv and cmd[-len(argv) - 1].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.spawn_ranks(4, argv, device_count=2, runner=runner) == 2          # too few GPUs: refused, nothing started
    assert "refusing" in capsys.readouterr().err
    # end to end on this box (no GPU): a non-zero exit and the message, not a JSON line with n_gpus = 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, bench.__file__, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "refusing" in r.stderr and "n_gpus" not in r.stdout
    # a launcher whose world differs from --gpus is refused as well
    env2 = dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_PORT="29518", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, bench.__file__, "--gpus", "4"], env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


class _FakeCommCtx:
    """what parallel.native_comm needs of a context: an id to draw and a place to join with it"""

    def __init__(self):
        self.joined = None

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, n_ranks, rank, uid):
        self.joined = (n_ranks, rank, bytes(uid))


def _native_comm_rank(rank, world, port, out_dir, use_group):
    """one rank of the id exchange: through torch.distributed's group (gloo here) or through a file"""
    import torch.distributed as dist
    from uf3_amd import parallel
    ctx = _FakeCommCtx()
    if use_group:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        parallel.native_comm(ctx, rank, world)
        dist.destroy_process_group()
    else:
        parallel.native_comm(ctx, rank, world, id_path=os.path.join(out_dir, "comm_id"))
    np.save(os.path.join(out_dir, f"joined_{int(use_group)}_{rank}.npy"), np.frombuffer(ctx.joined[2], dtype=np.uint8))
    assert ctx.joined[:2] == (world, rank)


@pytest.mark.parametrize("use_group", [True, False])
def test_native_comm_carries_rank_zeros_id_to_every_rank(tmp_path, use_group):
    """parallel.native_comm (the host side of RCCL behind the C ABI, uf3_comm_init): the 128-byte id rank 0 draws reaches the
    other rank through the process group when one is up, else through a file; every rank joins with (world, rank, id)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_native_comm_rank, args=(2, port, str(tmp_path), use_group), nprocs=2, join=True)
    ids = [np.load(tmp_path / f"joined_{int(use_group)}_{r}.npy") for r in range(2)]
    assert np.array_equal(ids[0], np.arange(128, dtype=np.uint8)) and np.array_equal(ids[0], ids[1])
    from uf3_amd import parallel
    lone = _FakeCommCtx()
    parallel.native_comm(lone, 0, 1)
    assert lone.joined == (1, 0, bytes(range(128)))
    with pytest.raises(ValueError):
        parallel.native_comm(_FakeCommCtx(), 1, 2)              # two ranks, no group, no file: no way to pass the id
