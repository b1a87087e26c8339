"""world_size-2 gloo run of the sharded normal-equation reduce (the N>1 path, on CPU)."""
import json
import os
import socket

import numpy as np
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


def test_shard_range_and_packing():
    from uf3_amd import parallel
    parts = [parallel.shard_range(10, r, 4) for r in range(4)]
    assert parts == [(0, 3), (3, 6), (6, 8), (8, 10)]
    rng = np.random.default_rng(0)
    p = dict(gram_e=rng.random((4, 4)), gram_f=rng.random((4, 4)), ord_e=rng.random(4), ord_f=rng.random(4),
             m_e=rng.random(3), m_f=rng.random(3))
    q = parallel.unpack_pieces(parallel.pack_pieces(p, 4), 4)
    assert all(np.array_equal(p[k], q[k]) for k in p)
