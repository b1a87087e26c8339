#!/usr/bin/env python3
"""
bench.py -- featurized frames/s of the UF3 hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the featurizer (cell list -> 3-body neighbour lists -> energy row +
3N force rows) over one batch of --frames-per-step synthetic frames of the north-star
workload: 10 000-atom rattled bcc cells, 2 elements, 2+3-body notebook basis, F = 434
(SURVEY 8d, config C4 metric variant).  Inputs (positions, species) and outputs (X_e, X_f)
are resident in HBM; the timed region holds exactly K steps between barrier +
torch.cuda.synchronize() pairs; value = all ranks' frames / max-over-ranks time.

N > 1: launched by torch.distributed.run, one rank per GPU.  Frames are independent, so
the data path has no collective (weak scaling: every rank featurizes its own batch); the
barrier and the MAX-reduce of the elapsed time go over RCCL.

Extra objects on the JSON line:
  roofline      achieved = algorithmic bytes per k_featurize launch (28N+75+8F(3N+1)
                per frame x frames per launch) / mean launch duration measured with HIP events
                on the launch stream, against the 8 TB/s HBM3E peak.  The path is FP64-ALU/LDS
                bound at this F, so the fp64 fraction is reported beside it.
  cpu_baseline  the oracle (oracle/uf3_oracle.c, "port", 1 thread) timed on rank 0 on one frame
                of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-workers", type=int, default=0, help="processes of the cpu_baseline pool (0: all cores, <= 32)")
    ap.add_argument("--frames-per-step", type=int, default=32,
                    help="frames per step and rank (32 x 10k atoms: 3.3 GB of rows in HBM); smaller batches leave a "
                         "few per cent on the table to workgroup tail effects")
    ap.add_argument("--atoms", type=int, default=10000, help="10000 = north-star; smaller = debug only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from uf3_amd import _lib, synthetic
    from uf3_amd.representation import process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if distributed else 0)
    os.environ["UF3_DEVICE"] = str(dev.index)

    # ---- workload ---------------------------------------------------------------------
    if args.atoms == 10000:
        reps, workload = (10, 20, 25), "C4-metric: 10k-atom bcc, W/Mo Bernoulli(0.5), 2+3-body notebook basis"
    else:
        n = max(2, round((args.atoms / 2) ** (1 / 3)))
        reps, workload = (n, n, n), f"debug: {2 * n ** 3}-atom bcc, W/Mo"
    basis = synthetic.notebook_basis(['Mo', 'W'])
    B = args.frames_per_step
    frames = [synthetic.lattice_frame("bcc", reps, 3.165, [42, 74], 3000 + rank * 1000 + k) for k in range(B)]
    batch = _lib.FrameBatch(frames)
    n_atoms = len(frames[0])
    fz = process.BasisFeaturizer(basis, device=dev.index)
    ctx, db = fz._dev()
    F = db.n_feat
    d_pos = torch.from_numpy(batch.pos).to(dev)
    d_z = torch.from_numpy(batch.z).to(dev)
    d_xe = torch.empty((B, F), dtype=torch.float64, device=dev)
    d_xf = torch.empty((batch.n_atoms, 3, F), dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def step():
        fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), d_xe.data_ptr(), d_xf.data_ptr())

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    ctx.timing_reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    timing = ctx.timing_read()
    ctx.timing_reset(False)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * B * args.steps / elapsed

    # ---- sanity of what was timed: rows are finite and obey translation invariance -------------
    xf_sum = d_xf[:n_atoms].sum(dim=0).abs().max().item()
    assert os.environ.get("UF3_BENCH_NOCHECK") or np.isfinite(xf_sum) and xf_sum < 1e-6 * d_xf[:n_atoms].abs().max().item(), xf_sum

    out = None
    if rank == 0:
        launches = max(1, timing["featurize_launches"])
        launch_ms = timing["featurize_ms"] / launches
        bytes_per_launch = synthetic.algorithmic_bytes(n_atoms, F) * B
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        # algorithmic fp64 work: ~100 flop per directed pair, ~3.1 kflop per triplet
        pairs_per_atom, trip_per_atom = 58.0, 91.0
        flops_frame = n_atoms * (100.0 * pairs_per_atom + 3100.0 * trip_per_atom)
        # HBM bytes per launch from the PMC passes of this same command (separate rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE runs, summary committed under profiles/); null when the workload differs
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "round1_hbm_counters.json")))
            if pmc["workload"] == dict(atoms_per_frame=n_atoms, n_feat=F, frames_per_step=B):
                traffic = pmc["hbm_bytes_per_launch_raw"]
        except (OSError, KeyError, ValueError):
            pass
        roofline = dict(bound="hbm", achieved=round(achieved, 2), peak=8000.0, unit="GB/s",
                        frac=round(achieved / 8000.0, 5), traffic=traffic,
                        kernel="k_featurize<E,F,R,MODE> launch group of one step: MODE 0 (one-body + pairs + 3-body list "
                               "build) + MODE 7 (3-body windows on the fp64 matrix cores, 3 waves/SIMD)",
                        launch_ms=round(launch_ms, 4), launches=launches,
                        algorithmic_bytes_per_launch=bytes_per_launch,
                        fp64_tflops=round(flops_frame * B / (launch_ms * 1e-3) / 1e12, 3), fp64_peak_tflops=78.6,
                        neighbor_ms_per_step=round(timing["neighbor_ms"] / args.steps, 4))
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N=1 only
            from oracle import oracle as O
            ob = O.OracleBasis(basis)
            t1 = time.perf_counter()
            ref = O.featurize(ob, frames[0])
            dt = time.perf_counter() - t1
            got_e = d_xe[0].cpu().numpy()
            got_f = d_xf[:n_atoms].cpu().numpy()
            err = max(np.abs(got_e - ref["xe"]).max() / np.abs(ref["xe"]).max(),
                      np.abs(got_f - ref["xf"]).max() / np.abs(ref["xf"]).max())
            assert err < 1e-9, err
            # the reference fans frames out over a process pool (process.py:196-254): one frame per process on all
            # host cores (at most 32, one frame each: ~2.5 s per frame), wall clock from the first start to the last end
            import multiprocessing as mp
            workers = max(1, min(args.cpu_workers or (os.cpu_count() or 1), 32, B))
            value_cpu, cores, how = 1.0 / dt, 1, "single thread"
            if workers > 1:
                try:
                    with mp.get_context("spawn").Pool(workers) as pool:
                        spans = pool.map_async(_cpu_worker, [(reps, 3000 + k) for k in range(workers)]).get(timeout=240)
                    wall = max(t[1] for t in spans) - min(t[0] for t in spans)
                    value_cpu, cores = workers / wall, workers
                    how = f"{workers} processes x 1 frame each in {wall:.1f} s (one process alone: {dt:.1f} s per frame)"
                except Exception as exc:  # noqa: BLE001 - the pool is a convenience; the single-thread figure stands
                    how = f"single thread (process pool failed: {type(exc).__name__})"
            cpu = dict(value=round(value_cpu, 5), unit="frames/s", cores=cores, kind="port",
                       sample=f"frames of the same workload ({n_atoms} atoms, F={F}), energy + force rows, "
                              f"oracle/uf3_oracle.c, {how}; GPU rows matched frame 0 to {err:.1e}")
        out = dict(metric="featurized frames/sec (10k-atom, 2-elem, 2+3-body)", value=round(value, 3),
                   unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(1e3 * elapsed / args.steps, 4), higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=workload, atoms_per_frame=n_atoms, n_feat=F, frames_per_step=B,
                               outputs="energy row + 3N force rows per frame, resident in HBM",
                               sharding=f"frames x{world}, no data-path collective"),
                   roofline=roofline, cpu_baseline=cpu)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return out


def _cpu_worker(job):
    """one frame of the workload through the CPU restatement (a process of the cpu_baseline pool)"""
    reps, seed = job
    from oracle import oracle as O
    from uf3_amd import synthetic
    ob = O.OracleBasis(synthetic.notebook_basis(['Mo', 'W']))
    frame = synthetic.lattice_frame("bcc", reps, 3.165, [42, 74], seed)
    t0 = time.time()
    O.featurize(ob, frame)
    return t0, time.time()


if __name__ == "__main__":
    main()
