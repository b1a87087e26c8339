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

--mode fit (BASELINE config 4: W 2+3-body, 10k-atom frames, F = 73 unless --workload says otherwise): a step =
featurize one batch of frames into HBM rows + accumulate X^T X / X^T y of the energy and force rows on the fp64
matrix cores (the rows never leave the GPU); for N > 1 the timed region ends with the ONE all_reduce of the packed
pieces over RCCL.  --workload lead0 (2-element, no leading trim, F = 1798) is the bandwidth-heavier featurize case.

Extra objects on the JSON line:
  roofline      the dominant kernel launch group of a step, timed with HIP events on the launch stream.  Both
                ceilings are reported: `hbm` = algorithmic bytes (28N+75+8F(3N+1) per frame x frames per
                launch) against 8 TB/s, and `fp64` = algorithmic flops against 78.6 TF.  `bound` names the measured
                limiter: on MI355X an fp64 MFMA holds the SIMD's vector issue while it runs
                (tools/experiments/dp_pipe_bench.hip), so matrix and vector fp64 work add up -- the launch is bound by
                fp64 issue ("mfma" in the contract's vocabulary), not by HBM, at these F.  `traffic` is measured in the run
                (N = 1, featurize mode): two short child runs of this command under `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE,
                separate passes), after the timed region; --no-traffic, or a box without the profiler, quotes the committed
                PMC passes under profiles/ instead, labelled as such.
  cpu_baseline  the oracle (oracle/uf3_oracle.c, "port") timed on rank 0 on frames of the same workload.
  extra         (N = 1, default mode only; measured AFTER the headline's timed region) one sub-line per other BASELINE
                configuration, each with ms_per_step and its own roofline: fit_w (config 4: W, F = 73, featurize + X^T X /
                X^T y), fit_c4 (the same at F = 434), lead0 (F = 1798: the bandwidth-heavier featurize case), eval_50k
                (config 5: ternary 50k-atom frame, uf3_eval_dev loop, with the oracle's evaluator as cpu_baseline) and
                eval_128 (latency of one 128-atom UFCalculator call).  --no-extra skips them.
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


def rank_command(n_gpus, argv, port, python=None):
    """The command line that starts ``n_gpus`` ranks of this script on one node -- the same one the driver uses for N > 1
    (one process per GPU, rendezvous on 127.0.0.1)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n_gpus, argv, device_count=None, runner=None, port=None):
    """``python bench.py --gpus N`` without a launcher around it: start the N ranks here, or refuse LOUDLY when this node
    does not have N GPUs -- never a silent one-rank run that reports n_gpus = 1.  Returns the exit code of the launcher.
    ``device_count`` / ``runner`` / ``port`` are injection points of the CPU-side test."""
    if device_count is None:
        import torch
        device_count = torch.cuda.device_count()
    if device_count < n_gpus:
        sys.stderr.write(f"bench.py: --gpus {n_gpus} asked for, {device_count} visible on this node: refusing to run "
                         f"(no silent fallback to fewer ranks)\n")
        return 2
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    if runner is None:
        import subprocess
        runner = subprocess.call
    return int(runner(rank_command(n_gpus, argv, port), env=env))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE under torch.distributed.run, else 1); without a "
                         "launcher around it, --gpus N > 1 starts the N ranks itself")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-workers", type=int, default=0, help="processes of the cpu_baseline pool (0: all cores, <= 32)")
    ap.add_argument("--frames-per-step", type=int, default=128,
                    help="frames per step and rank (128 x 10k atoms: 13.4 GB of rows in HBM, sized for 288 GB); smaller "
                         "batches leave a few per cent on the table to launch tails (32: -2.5 %%)")
    ap.add_argument("--atoms", type=int, default=10000, help="10000 = north-star; smaller = debug only")
    ap.add_argument("--mode", choices=["featurize", "fit", "eval"], default="featurize",
                    help="featurize = BASELINE metric (rows into HBM); fit = config 4 (rows -> X^T X / X^T y on the device); "
                         "eval = config 5 (one 50k-atom ternary frame per step, decomposed over the ranks: a block of centres "
                         "per rank + one all_reduce of [forces | energy | strain derivative])")
    ap.add_argument("--workload", choices=["c4", "w", "lead0"], default=None,
                    help="c4 = W/Mo notebook basis F=434 (featurize default); w = W only, F=73 (fit default); "
                         "lead0 = W/Mo without leading trim, F=1798 (bandwidth-heavier)")
    ap.add_argument("--row-ld", type=int, default=0,
                    help="doubles between consecutive force rows in HBM (featurize mode): 0 = F (dense rows, the default), -1 = F "
                         "rounded up to a multiple of 16 (every row on 128-byte lines of its own), else the stride itself")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the sub-lines of the other BASELINE configurations")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two short child runs under rocprofv3 --pmc); the figure "
                         "of the committed PMC passes under profiles/ is quoted instead")
    args = ap.parse_args(argv)

    distributed = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    world = int(os.environ.get("WORLD_SIZE", "1")) if distributed else 1
    if args.gpus is None:
        args.gpus = world
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if not distributed and args.gpus > 1:
        # (the driver's N = 1 command shape with a larger N: be the launcher)
        sys.exit(spawn_ranks(args.gpus, argv))
    if distributed and args.gpus != world:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to run\n")
        sys.exit(2)

    import torch
    import torch.distributed as dist
    from uf3_amd import _lib, synthetic
    from uf3_amd.representation import process

    rank = int(os.environ.get("RANK", "0")) if distributed else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if distributed else 0
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible: no GPU, no number "
                         f"(there is no CPU path)\n")
        sys.exit(2)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        # (RCCL prints its version banner on stdout when the first communicator comes up: file descriptor 1 points at stderr
        # until then, so that this program's stdout stays the ONE JSON line)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        assert dist.get_world_size() == world == args.gpus
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if distributed else 0)
    os.environ["UF3_DEVICE"] = str(dev.index)

    if args.mode == "eval":
        out = eval_mode(args, torch, dist, dev, distributed, world, rank)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return out

    # ---- workload ---------------------------------------------------------------------
    if args.atoms == 10000:
        reps, workload = (10, 20, 25), "C4-metric: 10k-atom bcc, W/Mo Bernoulli(0.5), 2+3-body notebook basis"
    else:
        n = max(2, round((args.atoms / 2) ** (1 / 3)))
        reps, workload = (n, n, n), f"debug: {2 * n ** 3}-atom bcc, W/Mo"
    wl = args.workload or ("w" if args.mode == "fit" else "c4")
    elements, numbers, lead3 = (['W'], [74], 3) if wl == "w" else (['Mo', 'W'], [42, 74], 0 if wl == "lead0" else 3)
    if wl == "w":
        workload = workload.replace("C4-metric", "C4-fit").replace("W/Mo Bernoulli(0.5)", "W")
    elif wl == "lead0":
        workload = workload.replace("C4-metric", "C4-cells").replace("notebook basis", "notebook basis without leading trim")
    basis = synthetic.notebook_basis(elements, lead3=lead3)
    B = args.frames_per_step
    if wl == "lead0" and B > 24:
        B = 24                    # 432 MB of rows per frame
    # this rank's block of the world x B frames of a step (weak scaling: B per rank), resident in HBM with its rows:
    # uf3_amd.parallel.featurize_sharded -- the package's driver, not a copy of it
    from uf3_amd import parallel
    fz = process.BasisFeaturizer(basis, device=dev.index)
    fit = args.mode == "fit"
    fbatch, _ = parallel.featurize_sharded(
        fz, lambda i: synthetic.lattice_frame("bcc", reps, 3.165, numbers, 3000 + (i // B) * 1000 + (i % B)), n_frames=world * B,
        device=dev.index, ld=0 if (args.row_ld == 0 or fit) else args.row_ld, rank=rank, world_size=world)
    frames, batch = fbatch.frames, fbatch.batch
    n_atoms = len(frames[0])
    ctx, db = fz._dev()
    F = db.n_feat
    d_pos, d_z, d_xe, d_xf_full, d_xf, ld = fbatch.pos, fbatch.z, fbatch.x_e, fbatch.x_f_full, fbatch.x_f, fbatch.ld
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    acc = flat = None
    if fit:
        # targets of the batch (synthetic: any numbers do for the arithmetic; the parity of the fit is a test), already
        # in HBM like the positions; energies per-atom normalised as the pipeline does
        from uf3_amd import pipeline
        from uf3_amd.regression import least_squares as ls
        model = ls.WeightedLinearModel(basis, regularizer=basis.get_regularization_matrix(
            ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0))
        acc = pipeline.DeviceFitAccumulator(model, fz, with_forces=True)
        g = torch.Generator(device=dev).manual_seed(7 + rank)
        d_counts = torch.from_numpy(np.diff(batch.offsets).astype(np.float64)).to(dev)
        d_ye = torch.randn((B,), dtype=torch.float64, device=dev, generator=g)
        d_yf = torch.randn((3 * batch.n_atoms,), dtype=torch.float64, device=dev, generator=g)

    def step():
        if fit:
            acc.add_device_batch(batch.struct, B, batch.n_atoms, d_pos, d_z, d_counts, d_ye, d_yf, x_e=d_xe, x_f=d_xf)
        else:
            fbatch.run()

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if fit and distributed and os.environ.get("UF3_NATIVE_RCCL"):
        from uf3_amd import parallel
        parallel.native_comm(ctx, rank, world)           # (the communicator comes up outside the timed region, like torch's)
    for _ in range(args.warmup):
        step()
    fence()
    ctx.synchronize()
    if fit:
        acc.reset()
    ctx.timing_reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if fit:
        # the pieces of this rank, frozen columns folded out on the device, and the ONE collective of the fit
        from uf3_amd import parallel
        # (UF3_NATIVE_RCCL: the collective through the library's own communicator, uf3_allreduce_sum_f64, instead of torch.distributed's)
        flat = parallel.allreduce_packed(acc.packed(), ctx=ctx if os.environ.get("UF3_NATIVE_RCCL") else None)
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
    if fit:
        assert bool(torch.isfinite(flat).all())

    out = None
    if rank == 0:
        launches = max(1, timing["featurize_launches"])
        launch_ms = timing["featurize_ms"] / launches
        bytes_per_launch = synthetic.algorithmic_bytes(n_atoms, F) * B
        hbm_gbs = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        # algorithmic fp64 work of the featurizer: ~100 flop per directed pair, ~3.1 kflop per triplet
        # (p, T as realised on frame 0 of this batch, from the device's own lists -- not the ideal lattice's 58 / 91)
        pairs_per_atom, trip_per_atom = realised_counts(fz, frames[0])
        flops_frame = n_atoms * (100.0 * pairs_per_atom + 3100.0 * trip_per_atom)
        feat_tf = flops_frame * B / (launch_ms * 1e-3) / 1e12
        # HBM bytes per launch from the PMC passes of this same command (separate rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE runs, summary committed under profiles/); null when the workload differs
        traffic, traffic_source = None, None
        if world == 1 and not fit and not args.no_traffic:
            traffic, traffic_source = measure_traffic(B, args.workload, args.atoms, args.row_ld)
        for name in (() if traffic is not None else ("round6_hbm_counters.json", "round5_hbm_counters.json", "round4_hbm_counters.json", "round3_hbm_counters.json")):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                if pmc["workload"] == dict(atoms_per_frame=n_atoms, n_feat=F, frames_per_step=B):
                    traffic, traffic_source = pmc["hbm_bytes_per_launch_raw"], "profiles/" + name + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)"
                    break
            except (OSError, KeyError, ValueError):
                pass
        feat_kernel = ("k_featurize launch group of one step: k_featurize<..., MODE 0> (one-body + pair columns + 3-body list build) + "
                       "k_featurize3 (3-body energy / force rows by bond factorisation, fp64 vector units, no matrix cores; "
                       + ("6 x 12 windows in three rounds" if wl == "lead0" else "3 x 9 windows of the default trims, four waves per SIMD") + ")")
        hbm = dict(achieved=round(hbm_gbs, 2), peak=8000.0, unit="GB/s", frac=round(hbm_gbs / 8000.0, 5),
                   algorithmic_bytes_per_launch=bytes_per_launch)
        if not fit:
            roofline = dict(bound="mfma", achieved=round(feat_tf, 3), peak=78.6, unit="TFLOP/s", frac=round(feat_tf / 78.6, 5),
                            traffic=traffic, traffic_source=traffic_source, hbm=hbm,
                            bound_note="compute roofline = the fp64 peak, 78.6 TF (vector and matrix fp64 are one pipe on MI355X, "
                                       "tools/experiments/dp_pipe_bench.hip); since round 4 the 3-body rows come from the vector units "
                                       "alone (k_featurize3: ~5400 vector instructions and ~6100 LDS cycles per atom, LDS ~80 % busy); "
                                       "achieved = ALGORITHMIC flops (SURVEY 8d), the HBM view is in `hbm`",
                            kernel=feat_kernel, launch_ms=round(launch_ms, 4), launches=launches,
                            algorithmic_flops_per_launch=flops_frame * B,
                            realised_per_atom=dict(pairs=round(pairs_per_atom, 2), triplet_candidates=round(trip_per_atom, 2),
                                                   note="frame 0 of the batch, from the device's lists (uf3_neighbors_debug); the ideal "
                                                        "bcc lattice has 58 / 91"),
                            executed_fp64=_executed_fp64(n_atoms, F, B, launch_ms),
                            neighbor_ms_per_step=round(timing["neighbor_ms"] / args.steps, 4))
        else:
            n_keep = int(acc._keep.numel())
            # Gram flops: the 16 x 16 tiles of the UPPER triangle of the dense product over all F columns (the kernels skip the
            # lower one; with two or more species the force rows are multiplied on their species' columns only -- fewer still);
            # SURVEY 8d's full-matrix count 2 (3N+1) F'^2 is kept beside it for reference only
            tiles = (F + 15) // 16
            gram_flops = 2.0 * (3 * n_atoms + 1) * 256.0 * (tiles * (tiles + 1) / 2) * B
            gram_flops_full = 2.0 * (3 * n_atoms + 1) * n_keep * n_keep * B
            gram_ms = timing["gram_ms"] / args.steps
            gram_tf = gram_flops / (gram_ms * 1e-3) / 1e12
            step_tf = (flops_frame * B + gram_flops) / (1e-3 * 1e3 * elapsed / args.steps) / 1e12
            roofline = dict(bound="mfma", achieved=round(step_tf, 3), peak=78.6, unit="TFLOP/s", frac=round(step_tf / 78.6, 5),
                            traffic=None, hbm=hbm,
                            bound_note="whole step (featurizer + X^T X): fp64 flops (featurizer: SURVEY 8d per pair / triplet; Gram: "
                                       "the upper-triangle tiles of the dense product; the species-wise launches of a multi-element batch execute fewer) / step time; "
                                       "the rows stay in HBM / Infinity Cache",
                            kernel="k_featurize launch group + k_gram_tiled / k_gram_mfma (energy and force rows, X^T y fused)",
                            featurize_ms_per_step=round(launch_ms, 4), gram_ms_per_step=round(gram_ms, 4),
                            gram_tflops_executed_triangle=round(gram_tf, 3), gram_flops_per_step=gram_flops,
                            gram_flops_full_matrix=gram_flops_full, n_unfrozen_columns=n_keep,
                            neighbor_ms_per_step=round(timing["neighbor_ms"] / args.steps, 4))
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N=1 only
            from oracle import oracle as O
            ob = O.OracleBasis(basis)
            step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            ref = O.featurize(ob, frames[0])
            dt = time.perf_counter() - t1
            got_e = d_xe[0].cpu().numpy() * (n_atoms if fit else 1.0)        # (the fit step leaves per-atom rows)
            got_f = d_xf[:n_atoms].cpu().numpy()
            err = max(np.abs(got_e - ref["xe"]).max() / np.abs(ref["xe"]).max(),
                      np.abs(got_f - ref["xf"]).max() / np.abs(ref["xf"]).max())
            assert err < 1e-9, err
            dt_fit = 0.0
            if fit:       # the oracle's normal equations on this frame's rows (NumPy, least_squares.py:716-760)
                t2 = time.perf_counter()
                xf = ref["xf"].reshape(-1, F)
                _ = xf.T @ xf, xf.T @ np.ones(len(xf)), np.outer(ref["xe"], ref["xe"])
                dt_fit = time.perf_counter() - t2
            # the reference fans frames out over a process pool (process.py:196-254): one frame per process on all
            # host cores (at most 32, one frame each), wall clock from the first start to the last end
            import multiprocessing as mp
            workers = max(1, min(args.cpu_workers or (os.cpu_count() or 1), 32, B))
            value_cpu, cores, how = 1.0 / (dt + dt_fit), 1, "single thread"
            if workers > 1:
                try:
                    with mp.get_context("spawn").Pool(workers) as pool:
                        spans = pool.map_async(_cpu_worker, [(reps, 3000 + k, elements, numbers, lead3, fit) for k in range(workers)]).get(timeout=400)
                    wall = max(t[1] for t in spans) - min(t[0] for t in spans)
                    value_cpu, cores = workers / wall, workers
                    how = f"{workers} processes x 1 frame each in {wall:.1f} s (one process alone: {dt + dt_fit:.1f} s per frame)"
                except Exception as exc:  # noqa: BLE001 - the pool is a convenience; the single-thread figure stands
                    how = f"single thread (process pool failed: {type(exc).__name__})"
            what = "energy + force rows" + (" + X^T X / X^T y of the rows (NumPy)" if fit else "")
            cpu = dict(value=round(value_cpu, 5), unit="frames/s", cores=cores, kind="port",
                       sample=f"frames of the same workload ({n_atoms} atoms, F={F}), {what}, "
                              f"oracle/uf3_oracle.c, {how}; GPU rows matched frame 0 to {err:.1e}")
            cpu["reference_estimate"] = reference_estimate(1.0 / (dt + dt_fit), value_cpu, cores)
        metric = ("featurized frames/sec (10k-atom, 2-elem, 2+3-body)" if not fit else
                  "fitted frames/sec (featurize + X^T X / X^T y accumulate, 10k-atom frames)")
        out = dict(metric=metric, value=round(value, 3),
                   unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(1e3 * elapsed / args.steps, 4), higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=workload, mode=args.mode, atoms_per_frame=n_atoms, n_feat=F, frames_per_step=B,
                               outputs=(("energy row + 3N force rows per frame, resident in HBM" + (f", rows {ld} doubles apart" if ld != F else ""))
                                        if not fit else
                                        "packed {G_e, G_f, o_e, o_f, moments} of the unfrozen columns"),
                               sharding=(f"frames x{world}, no data-path collective" if not fit else
                                         f"frames x{world}, one all_reduce(SUM) of 2F'^2+2F'+6 doubles at the end")),
                   roofline=roofline, cpu_baseline=cpu)
        out["config"]["rccl_world_size"] = dist.get_world_size() if distributed else 1
        out["config"]["forced_collective"] = bool(distributed and os.environ.get("UF3_FORCE_COLLECTIVE"))
        out["config"]["collective"] = ("uf3_allreduce_sum_f64 (librccl behind the C ABI)" if distributed and os.environ.get("UF3_NATIVE_RCCL")
                                       else "torch.distributed all_reduce (RCCL)")
        if world == 1 and not fit and wl == "c4" and args.atoms == 10000 and not args.no_extra:
            out["extra"] = extra_lines(dev, ctx, fz, frames, batch, d_pos, d_z, d_xe, d_xf_full,
                                       cpu=not args.no_cpu_baseline)
            # the other BASELINE configurations again, compact, inside an object the driver's parser keeps
            out["roofline"]["configs"] = compact_configs(out["extra"])
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return out


# ---------------------------------------------------------------------------------------------------------------
# --mode eval: BASELINE config 5, one large frame decomposed over the ranks (strong scaling)
# ---------------------------------------------------------------------------------------------------------------
def eval_mode(args, torch, dist, dev, distributed, world, rank):
    """A step = energy + forces (+ strain derivative) of ONE 50 000-atom ternary frame.  N = 1: the whole-frame evaluator
    (uf3_eval_virial_dev).  N > 1: every rank holds the frame's positions, evaluates the triplets centred in its contiguous block
    of atoms (uf3_eval_centres_dev) into one flat device buffer [forces (3N) | energy | dE/d(strain) (6)] and the ranks sum
    that buffer with ONE all_reduce over RCCL (parallel.sharded_evaluate's arithmetic, device-resident).  The reference's
    calculator is a single process (uf3/forcefield/calculator.py:124-153)."""
    from uf3_amd import _lib, parallel, synthetic
    basis = synthetic.notebook_basis(['V', 'Mo', 'W'])
    reps = (25, 25, 40) if args.atoms == 10000 else (max(2, round((args.atoms / 2) ** (1 / 3))),) * 3
    atoms = synthetic.lattice_frame("bcc", reps, 3.165, [23, 42, 74], 4000)
    model, calc = _random_model(basis, 11)
    # UF3_FORCE_COLLECTIVE under the launcher with ONE rank: the decomposed route (a block of centres = the whole frame) and its
    # all_reduce on the device buffer, so that a one-GPU box exercises what the ranks of an 8-GPU node run
    forced = bool(distributed and os.environ.get("UF3_FORCE_COLLECTIVE"))
    native = bool(distributed and os.environ.get("UF3_NATIVE_RCCL"))       # the sum through uf3_allreduce_sum_f64 (librccl behind the C ABI)
    # An MD step -- every atom moves (seeded +-0.01 A walk, one device kernel) and the evaluator runs its MD route (neighbour
    # lists kept with a 0.5 A skin, rebuilt inside the timed region when an atom nears skin / 2): the whole frame at N = 1, a
    # block of centres on whole-frame lists per rank at N > 1.  The device-resident loop is the package's
    # (uf3_amd.parallel.ShardedEvaluator: flat buffer [forces | energy | dE/d(strain)], uf3_eval_centres_dev, one all_reduce).
    moving = True      # (every rank applies the same seeded walk: the ranks of a decomposed frame see the same positions)
    ev = parallel.ShardedEvaluator(calc, atoms, device=dev.index, md_skin=MD_SKIN if moving else 0.0, native=native,
                                   force_collective=forced)
    ctx, n, flat, d_pos = ev.ctx, ev.n, ev.flat, ev.positions
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    if moving:
        g = torch.Generator(device=dev).manual_seed(17)
        pool = (torch.rand((64, n, 3), dtype=torch.float64, device=dev, generator=g) * 2.0 - 1.0) * MD_WALK
        order = np.random.default_rng(17).integers(0, 64, 1 << 16)
        signs = np.random.default_rng(18).choice([-1.0, 1.0], 1 << 16)
    counter = [0]

    def step():
        k = counter[0] & 0xffff
        counter[0] += 1
        d_pos.add_(pool[order[k]], alpha=float(signs[k]))
        ev.step()

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    host = flat.cpu().numpy()
    if moving:
        from uf3_amd.data.atoms import Atoms
        atoms = Atoms(numbers=atoms.get_atomic_numbers(), positions=d_pos.cpu().numpy(), cell=atoms.get_cell(), pbc=True)   # (what the last step saw)
        md_stats = ctx.md_stats()
        ev.close()
    counts = _counts_of(basis, atoms, dev.index)              # (p, T of the frame the last step saw)
    f, e = host[:3 * n].reshape(n, 3), float(host[3 * n])
    assert os.environ.get("UF3_BENCH_NOCHECK") or (np.isfinite(host).all() and np.abs(f.sum(0)).max() < 1e-8 * max(1.0, np.abs(f).max()) * n ** 0.5)
    dt = elapsed / args.steps
    cpu = None
    if not args.no_cpu_baseline and n <= 60000:
        from oracle import oracle as O
        t1 = time.perf_counter()
        ref = O.evaluate(O.OracleBasis(basis), atoms, model.coefficients)
        dt_cpu = time.perf_counter() - t1
        err = max(abs(e - ref[0]) / abs(ref[0]), np.abs(f - ref[1]).max() / np.abs(ref[1]).max())
        assert err < 1e-9, err
        cpu = dict(value=round(n / dt_cpu), unit="atom-steps/s", cores=1, kind="port",
                   sample=f"the same frame once through oracle/uf3_oracle.c's evaluator, single thread, {dt_cpu:.1f} s; "
                          f"the reduced GPU energy / forces matched to {err:.1e}")
    out = dict(metric="evaluated atom-steps/sec (50k-atom ternary frame, energy + forces + strain derivative)", value=round(n / dt),
               unit="atom-steps/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt * 1e3, 4),
               higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload=f"C5: {n}-atom ternary bcc frame, 2+3-body notebook basis, F={int(basis.n_feats)}", mode="eval",
                           atoms_per_frame=n, sharding=(f"blocks of centres x{world}, one all_reduce(SUM) of 3N+7 doubles per step"
                                                        if world > 1 or forced else "whole frame on one GPU"),
                           md=(dict(skin_A=MD_SKIN, walk_A=MD_WALK, list_builds=md_stats["builds"], steps_repeated=md_stats["redone"])
                               if moving else None),
                           rccl_world_size=dist.get_world_size() if distributed else 1, forced_collective=forced,
                           collective=("uf3_allreduce_sum_f64 (librccl behind the C ABI)" if native else "torch.distributed all_reduce (RCCL)")),
               roofline=_roof(_eval_flops(n, counts), 52.0 * n + 75 + 8.0 * len(calc._c3), dt,
                              "mfma", realised_per_atom=_counts_note(counts),
                              note="flops = N (100 p + 700 T) at the realised p, T: every triplet once at its centre; 700 flop per triplet is "
                                   "this build's own price (SURVEY 8d prices the evaluator by bytes only)"),
               cpu_baseline=cpu)
    print(json.dumps(out), flush=True)
    return out


# ---------------------------------------------------------------------------------------------------------------
# roofline.traffic, measured in the run: HBM bytes of one k_featurize launch group from the PMC counters
# ---------------------------------------------------------------------------------------------------------------
def measure_traffic(frames_per_step, workload, atoms, row_ld=0):
    """Two short child runs of this script under `rocprofv3 --pmc` -- FETCH_SIZE and WRITE_SIZE in SEPARATE passes, counters
    only (no trace domains) -- on the same workload and batch; the k_featurize dispatches of a step are summed and averaged
    over the steps.  Units as in tools/summarize_profiles.py (MI355X_MICROARCH.md: the counters report KiB; on gfx950
    FETCH_SIZE can come out at half the bytes of wide streaming reads -- the reads here are mostly 48-byte list entries, so the
    uncorrected sum is what `traffic` quotes and the source string says so).  Returns (bytes per launch group | None, source)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None
    totals = {}
    work = tempfile.mkdtemp(prefix="uf3_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--no-traffic",
                   "--frames-per-step", str(frames_per_step), "--atoms", str(atoms), "--row-ld", str(row_ld)] + (["--workload", workload] if workload else [])
            env = dict(os.environ, TMPDIR="/tmp", UF3_BENCH_NOCHECK="1")
            subprocess.run(cmd, env=env, cwd=work, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            per_kernel = {}
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(path, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if "k_featurize" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            per_kernel.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            if not per_kernel:
                return None, None
            steps = max(len(v) for v in per_kernel.values())
            totals[counter] = sum(sum(v) for v in per_kernel.values()) / steps * 1024.0          # KiB -> bytes
        return (totals["FETCH_SIZE"] + totals["WRITE_SIZE"],
                f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of this command at "
                f"--steps 2 (k_featurize launches of a step summed; FETCH {totals['FETCH_SIZE'] / 1e9:.2f} GB + WRITE "
                f"{totals['WRITE_SIZE'] / 1e9:.2f} GB, uncorrected)")
    except Exception:  # noqa: BLE001 - no profiler, no counters: the committed figure is quoted instead
        return None, None
    finally:
        shutil.rmtree(work, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------
# sub-lines of the other BASELINE configurations (rank 0, N = 1, after the headline)
# ---------------------------------------------------------------------------------------------------------------
# (no hard-coded pair / triplet counts: every line prices its flops at the p and T REALISED on its own frame -- realised_counts)
EVAL_FLOP_PER_TRIPLET = 700.0      # the BUILD'S OWN price of an evaluated triplet (3 legs x ~60 + the 64-term contraction with value and 3
                                   # partials ~450 + forces): SURVEY 8d prices the evaluator by bytes only
PEAK_FP64_TF, PEAK_HBM_GBS = 78.6, 8000.0


def _executed_fp64(n_atoms, n_feat, frames_per_step, launch_ms):
    """The EXECUTED fp64 work of the headline launch group (the algorithmic figure prices every triplet at SURVEY 8d's
    3.1 kflop; the bond-factorised kernel issues about half of that): wave-instruction counts of the committed PMC pass
    (profiles/round*_fp64_counters.json, tools/profile_round.sh) over THIS run's launch time; null when no pass matches
    the workload."""
    for name in ("round6_fp64_counters.json", "round5_fp64_counters.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if pmc["workload"] == dict(atoms_per_frame=n_atoms, n_feat=n_feat, frames_per_step=frames_per_step):
                tf = pmc["executed_fp64_flops_per_launch"] / (launch_ms * 1e-3) / 1e12
                return dict(flops_per_launch=pmc["executed_fp64_flops_per_launch"], achieved=round(tf, 3), unit="TFLOP/s",
                            frac=round(tf / PEAK_FP64_TF, 5),
                            source="profiles/" + name + ": 64 x (2 FMA + MUL + ADD + TRANS) fp64 wave-instructions of one launch group "
                                   "(rocprofv3 --pmc, not measured in this run), lanes switched off by the exec mask included")
        except (OSError, KeyError, ValueError):
            pass
    return None


def realised_counts(fz, atoms):
    """(directed pairs inside their pair range, pairs j < k of 3-body neighbours) per atom of a frame, read back from the
    lists the device builds (uf3_neighbors_debug): SURVEY 8d's p and T as realised -- the rattle moves the 8 neighbours at
    5.48 A across the 5.5 A cut-off."""
    pairs, n3 = fz.neighbor_indices(atoms)
    n = len(atoms)
    q = np.bincount(n3[:, 0], minlength=1).astype(np.float64) if len(n3) else np.zeros(1)
    return sum(len(v) for v in pairs.values()) / n, float((q * (q - 1) / 2).sum()) / n


def _featurizer_flops(n_atoms, counts):
    """~100 flop per directed pair, ~3.1 kflop per triplet, at the realised (p, T) of the line's own frame"""
    return n_atoms * (100.0 * counts[0] + 3100.0 * counts[1])


def _eval_flops(n_atoms, counts):
    """the evaluator: ~100 flop per directed pair, EVAL_FLOP_PER_TRIPLET per triplet (every triplet once, at its centre)"""
    return n_atoms * (100.0 * counts[0] + EVAL_FLOP_PER_TRIPLET * counts[1])


def _counts_of(basis, atoms, device=None):
    """realised (p, T) of a frame through a featurizer of its basis"""
    from uf3_amd.representation import process
    return realised_counts(process.BasisFeaturizer(basis, device=device), atoms)


def _counts_note(counts):
    return dict(pairs=round(counts[0], 2), triplet_candidates=round(counts[1], 2),
                note="realised on this line's own frame, from the device's lists (uf3_neighbors_debug)")


def _roof(flops, bytes_, seconds, bound, **more):
    tf, gbs = flops / seconds / 1e12, bytes_ / seconds / 1e9
    main = (dict(bound="mfma", achieved=round(tf, 3), peak=PEAK_FP64_TF, unit="TFLOP/s", frac=round(tf / PEAK_FP64_TF, 5))
            if bound == "mfma" else
            dict(bound="hbm", achieved=round(gbs, 2), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 5)))
    main.update(fp64=dict(achieved=round(tf, 3), peak=PEAK_FP64_TF, unit="TFLOP/s", frac=round(tf / PEAK_FP64_TF, 5),
                          algorithmic_flops_per_step=flops),
                hbm=dict(achieved=round(gbs, 2), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 5),
                         algorithmic_bytes_per_step=bytes_), traffic=None)
    main.update(more)
    return main


def _timed(torch, dev, ctx, step, steps, warmup):
    """(seconds per step by the wall clock with the library's event timing OFF -- the events cost host time, which shows on
    sub-millisecond steps --, kernel-class milliseconds per step from a second, shorter loop with it on)"""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    ctx.synchronize()
    ctx.timing_reset(False)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    n2 = max(2, steps // 2)
    ctx.timing_reset(True)
    for _ in range(n2):
        step()
    torch.cuda.synchronize(dev)
    timing = ctx.timing_read()
    ctx.timing_reset(False)
    return dt, {k: v / n2 for k, v in timing.items()}


def extra_fit(torch, dev, basis, frames, batch, d_pos, d_z, d_xe, d_xf, steps=5, warmup=2):
    """BASELINE config 4 on one GPU: featurize a batch into HBM rows + X^T X / X^T y of energy and force rows (+ the packed,
    frozen-folded pieces once at the end, as `bench.py --mode fit` does)."""
    from uf3_amd import pipeline
    from uf3_amd.regression import least_squares as ls
    from uf3_amd.representation import process
    fz = process.BasisFeaturizer(basis, device=dev.index)
    ctx, db = fz._dev()
    F, B, n_atoms = db.n_feat, batch.n_frames, len(frames[0])
    model = ls.WeightedLinearModel(basis, regularizer=basis.get_regularization_matrix(
        ridge_1b=1e-8, ridge_2b=0.0, ridge_3b=1e-8, curvature_2b=1e-8, curvature_3b=0.0))
    acc = pipeline.DeviceFitAccumulator(model, fz, with_forces=True)
    g = torch.Generator(device=dev).manual_seed(7)
    d_counts = torch.from_numpy(np.diff(batch.offsets).astype(np.float64)).to(dev)
    d_ye = torch.randn((B,), dtype=torch.float64, device=dev, generator=g)
    d_yf = torch.randn((3 * batch.n_atoms,), dtype=torch.float64, device=dev, generator=g)
    xe, xf = d_xe.view(-1)[:B * F].view(B, F), d_xf.view(-1)[:3 * batch.n_atoms * F].view(batch.n_atoms, 3, F)

    def step():
        acc.add_device_batch(batch.struct, B, batch.n_atoms, d_pos, d_z, d_counts, d_ye, d_yf, x_e=xe, x_f=xf)

    dt, t = _timed(torch, dev, ctx, step, steps, warmup)
    flat = acc.packed()
    assert bool(torch.isfinite(flat).all())
    n_keep = int(acc._keep.numel())
    tiles = (F + 15) // 16
    # Gram flops: the 16 x 16 tiles of the upper triangle of the dense product over all F columns (the kernels skip the lower one,
    # and the species-wise launches of a multi-element batch the blocks a species takes no part in)
    gram_tri = 2.0 * (3 * n_atoms + 1) * 256.0 * (tiles * (tiles + 1) / 2) * B
    gram_full = 2.0 * (3 * n_atoms + 1) * n_keep * n_keep * B
    counts = realised_counts(fz, frames[0])
    flops = _featurizer_flops(n_atoms, counts) * B + gram_tri
    bytes_ = (52 * n_atoms + 75) * B                               # fused-Gram mode: inputs + targets
    return dict(metric="fitted frames/sec (featurize + X^T X / X^T y accumulate)", value=round(B / dt, 2), unit="frames/s",
                ms_per_step=round(dt * 1e3, 4), steps=steps, frames_per_step=B, atoms_per_frame=n_atoms, n_feat=F,
                n_unfrozen_columns=n_keep,
                roofline=_roof(flops, bytes_, dt, "mfma", featurize_ms_per_step=round(t["featurize_ms"], 4),
                               gram_ms_per_step=round(t["gram_ms"], 4), realised_per_atom=_counts_note(counts),
                               gram_tflops_executed_triangle=round(gram_tri / max(t["gram_ms"], 1e-9) / 1e9, 3),
                               gram_flops_full_matrix=gram_full,
                               note="fp64 flops = featurizer (SURVEY 8d) + the upper-triangle tiles the Gram kernels execute; "
                                    "hbm = SURVEY 8d's fused-Gram bytes (52 N per frame), information only: the rows "
                                    "themselves stay in HBM between the two kernels"))


def extra_lead0(torch, dev, frames, d_xf, steps=5, warmup=2):
    """the bandwidth-heavier featurize case: W/Mo without the leading 3-body trim, F = 1798 (432 MB of rows per frame)"""
    from uf3_amd import _lib, synthetic
    from uf3_amd.representation import process
    basis = synthetic.notebook_basis(['Mo', 'W'], lead3=0)
    fz = process.BasisFeaturizer(basis, device=dev.index)
    ctx, db = fz._dev()
    B = 24                                                      # 432 MB of rows per frame: 10.4 GB of the headline's 13.4 GB buffer
    batch = _lib.FrameBatch(frames[:B])
    F, n_atoms = db.n_feat, len(frames[0])
    d_pos = torch.from_numpy(batch.pos).to(dev)
    d_z = torch.from_numpy(batch.z).to(dev)
    xe = torch.empty((B, F), dtype=torch.float64, device=dev)
    xf = d_xf.view(-1)[:3 * batch.n_atoms * F].view(batch.n_atoms, 3, F)

    def step():
        fz.featurize_device(batch.struct, d_pos.data_ptr(), d_z.data_ptr(), xe.data_ptr(), xf.data_ptr())

    dt, t = _timed(torch, dev, ctx, step, steps, warmup)
    s = xf[:n_atoms].sum(dim=0).abs().max().item()
    assert np.isfinite(s) and s < 1e-6 * xf[:n_atoms].abs().max().item(), s           # translation invariance of the rows
    bytes_ = synthetic.algorithmic_bytes(n_atoms, F) * B
    counts = realised_counts(fz, frames[0])
    return dict(metric="featurized frames/sec (10k-atom, 2-elem, 2+3-body, no leading trim)", value=round(B / dt, 2),
                unit="frames/s", ms_per_step=round(dt * 1e3, 4), steps=steps, frames_per_step=B, atoms_per_frame=n_atoms,
                n_feat=F, roofline=_roof(_featurizer_flops(n_atoms, counts) * B, bytes_, dt, "hbm",
                                         featurize_ms_per_step=round(t["featurize_ms"], 4), realised_per_atom=_counts_note(counts),
                                         note="6.7 flop/B: the one workload SURVEY 8d calls HBM-bound on paper"))


def _random_model(basis, seed):
    from uf3_amd.forcefield import calculator
    from uf3_amd.regression import least_squares as ls
    model = ls.WeightedLinearModel(basis)
    coeff = np.random.default_rng(seed).normal(0, 0.05, basis.n_feats)
    coeff[basis.col_idx] = 0.0
    model.coefficients = coeff
    return model, calculator.UFCalculator(model)


MD_SKIN = 0.5          # Angstrom: skin of the evaluator's persistent neighbour lists in the MD-step lines
MD_WALK = 0.01         # Angstrom: every step moves every coordinate by a seeded uniform(-MD_WALK, MD_WALK) draw


def extra_eval_50k(torch, dev, cpu=True, steps=200, warmup=20):
    """BASELINE config 5 on one GPU: an MD step = move every atom (a seeded +-0.01 A random walk, applied on the device) + energy
    and forces of the 50 000-atom ternary frame (uf3_eval_dev, inputs and outputs in HBM).  The evaluator runs its MD route:
    neighbour lists kept out to r_cut + 0.5 A, rebuilt (cell list + list build, inside the timed region) whenever an atom has
    moved 0.7 x 0.25 A from where they were built.  The reference's calculator rebuilds everything on every call
    (uf3/forcefield/calculator.py:124-153)."""
    from uf3_amd import _lib, synthetic
    basis = synthetic.notebook_basis(['V', 'Mo', 'W'])
    atoms = synthetic.lattice_frame("bcc", (25, 25, 40), 3.165, [23, 42, 74], 4000)
    model, calc = _random_model(basis, 11)
    ctx = _lib.get_context(dev.index)
    db = _lib.device_basis(basis, ctx)
    batch = _lib.FrameBatch([atoms])
    n = batch.n_atoms
    d_pos = torch.from_numpy(batch.pos).to(dev)
    d_z = torch.from_numpy(batch.z).to(dev)
    d_e = torch.empty((1,), dtype=torch.float64, device=dev)
    d_f = torch.empty((n, 3), dtype=torch.float64, device=dev)
    # the walk: a pool of seeded displacement fields, one drawn (with a sign) per step -- one small device kernel per step
    g = torch.Generator(device=dev).manual_seed(17)
    pool = (torch.rand((64, n, 3), dtype=torch.float64, device=dev, generator=g) * 2.0 - 1.0) * MD_WALK
    order = np.random.default_rng(17).integers(0, 64, 1 << 16)
    signs = np.random.default_rng(18).choice([-1.0, 1.0], 1 << 16)
    counter = [0]
    fields = [pool[i] for i in range(64)]                                   # (views made once: the step's host time is the GPU's idle time)
    alphas = [float(x) for x in signs]
    prev = ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)      # (the walk runs on torch's stream: one order of events)
    ctx.md_skin(MD_SKIN)

    # (the call's arguments made once -- none of the addresses changes --: what the host does between two steps is GPU idle time)
    eval_args = (db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()), C.c_void_p(d_z.data_ptr()), _lib._p(calc._c1),
                 _lib._p(calc._c2), _lib._p(calc._c3), C.c_void_p(d_e.data_ptr()), C.c_void_p(d_f.data_ptr()))
    uf3_eval_dev, picks = ctx.lib.uf3_eval_dev, [int(x) for x in order]

    def step():
        k = counter[0] & 0xffff
        counter[0] += 1
        d_pos.add_(fields[picks[k]], alpha=alphas[k])
        rc = uf3_eval_dev(*eval_args)
        if rc:
            ctx.check(rc)

    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize(dev)
        s0 = ctx.md_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        s1 = ctx.md_stats()
        ctx.timing_reset(True)
        n2 = max(2, steps // 4)
        for _ in range(n2):
            step()
        torch.cuda.synchronize(dev)
        t = {k: v / n2 for k, v in ctx.timing_read().items()}
        ctx.timing_reset(False)
        f = d_f.cpu().numpy()
        e_gpu = float(d_e.item())
        final = d_pos.cpu().numpy()
        # the rebuild-everything route on the final positions (what every call did until round 4), timed on frozen atoms
        ctx.md_skin(0.0)
        d_f2 = torch.empty_like(d_f)

        def plain():
            ctx.check(ctx.lib.uf3_eval_dev(db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()),
                                           C.c_void_p(d_z.data_ptr()), _lib._p(calc._c1), _lib._p(calc._c2), _lib._p(calc._c3),
                                           C.c_void_p(d_e.data_ptr()), C.c_void_p(d_f2.data_ptr())))
        for _ in range(3):
            plain()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20):
            plain()
        torch.cuda.synchronize(dev)
        dt_plain = (time.perf_counter() - t0) / 20
        f_plain = d_f2.cpu().numpy()
        assert np.abs(f - f_plain).max() <= 1e-12 * np.abs(f_plain).max() and abs(e_gpu - float(d_e.item())) <= 1e-12 * abs(e_gpu)
    finally:
        ctx.md_skin(0.0)
        ctx.restore_stream(prev)
    counts = _counts_of(basis, atoms, dev.index)
    out = dict(metric="evaluated atom-steps/sec (50k-atom ternary frame, energy + forces, atoms moving every step)", value=round(n / dt),
               unit="atom-steps/s", ms_per_step=round(dt * 1e3, 4), steps=steps, atoms_per_frame=n, n_feat=int(basis.n_feats),
               md=dict(skin_A=MD_SKIN, walk_A=MD_WALK, list_builds_in_timed_steps=s1["builds"] - s0["builds"],
                       steps_repeated=s1["redone"] - s0["redone"],
                       rebuild_everything_route=dict(value=round(n / dt_plain), ms_per_step=round(dt_plain * 1e3, 4),
                                                     note="md skin 0 (cell list + candidate walk + list sort on every call), frozen positions")),
               roofline=_roof(_eval_flops(n, counts), 52.0 * n + 75 + 8.0 * len(calc._c3), dt, "mfma",
                              eval_kernels_ms_per_step=round(t["eval_ms"], 4), neighbor_ms_per_step=round(t["neighbor_ms"], 4),
                              realised_per_atom=_counts_note(counts),
                              note="flops = N (100 p + 700 T) at the realised p, T: every triplet once at its centre; 700 flop per triplet "
                                   "(3 legs x ~60 + 64-term contraction with value and 3 partials ~450 + forces) is this build's own price -- "
                                   "SURVEY 8d prices the evaluator by bytes only; bytes = 52 N + the coefficient grids"))
    if cpu:
        from oracle import oracle as O
        from uf3_amd.data.atoms import Atoms
        moved = Atoms(numbers=atoms.get_atomic_numbers(), positions=final, cell=atoms.get_cell(), pbc=True)
        ob = O.OracleBasis(basis)
        t0 = time.perf_counter()
        ref = O.evaluate(ob, moved, model.coefficients)
        dt_cpu = time.perf_counter() - t0
        e_ref, f_ref = ref[0], ref[1]
        err = max(abs(e_gpu - e_ref) / abs(e_ref), np.abs(f - f_ref).max() / np.abs(f_ref).max())
        assert err < 1e-9, err
        out["cpu_baseline"] = dict(value=round(n / dt_cpu), unit="atom-steps/s", cores=1, kind="port",
                                   sample=f"the frame of the last step once through oracle/uf3_oracle.c's evaluator, single thread, "
                                          f"{dt_cpu:.1f} s; GPU energy / forces (MD route, lists several steps old) matched to {err:.1e}")
    return out


def extra_eval_128(calls=2000):
    """latency of one MD step on a small cell: move the atoms (seeded +-0.01 A walk, host arrays) + UFCalculator.evaluate_frames
    (host arrays in and out), 128-atom W frame, neighbour lists kept with a 0.5 A skin and rebuilt inside the timed loop"""
    from uf3_amd import _lib, synthetic
    basis = synthetic.notebook_basis(['W'])
    atoms = synthetic.lattice_frame("bcc", (4, 4, 4), 3.165, [74], seed=3)
    model, _ = _random_model(basis, 1)
    from uf3_amd.forcefield import calculator
    calc = calculator.UFCalculator(model, md_skin=MD_SKIN)
    ctx = _lib.get_context(None)
    ctx.use_own_stream()            # (what a UFCalculator user has: the context's own stream, not the null stream torch's tensors live on)
    n = len(atoms)
    rng = np.random.default_rng(19)
    noise = rng.uniform(-MD_WALK, MD_WALK, (256, n, 3))
    noise = np.concatenate([noise, -noise])                 # (a pool of displacement fields and their negatives: one in-place add per step)
    pick = rng.integers(0, 512, calls + 64)
    for k in range(32):
        np.add(atoms.positions, noise[pick[k]], out=atoms.positions)
        calc.evaluate_frames([atoms])
    s0 = ctx.md_stats()
    t0 = time.perf_counter()
    for k in range(32, 32 + calls):
        np.add(atoms.positions, noise[pick[k]], out=atoms.positions)
        calc.evaluate_frames([atoms])
    dt = (time.perf_counter() - t0) / calls
    s1 = ctx.md_stats()
    e, f, _ = calc.evaluate_frames([atoms])
    plain = calculator.UFCalculator(model, md_skin=0.0)
    for _ in range(10):
        e0, f0, _ = plain.evaluate_frames([atoms])
    t0 = time.perf_counter()
    for _ in range(300):
        plain.evaluate_frames([atoms])
    dt_plain = (time.perf_counter() - t0) / 300
    assert np.abs(f - f0).max() <= 1e-12 * np.abs(f0).max() and abs(e[0] - e0[0]) <= 1e-12 * abs(e0[0])
    counts = _counts_of(basis, atoms)
    return dict(metric="latency of one MD step: move + energy + force call (128-atom W frame, host arrays in and out)", value=round(dt * 1e6, 2),
                unit="us/call", higher_is_better=False, ms_per_step=round(dt * 1e3, 5), steps=calls, atoms_per_frame=n,
                md=dict(skin_A=MD_SKIN, walk_A=MD_WALK, list_builds_in_timed_steps=s1["builds"] - s0["builds"],
                        steps_repeated=s1["redone"] - s0["redone"],
                        rebuild_everything_route=dict(value=round(dt_plain * 1e6, 2), unit="us/call", note="md skin 0, frozen positions")),
                roofline=_roof(_eval_flops(n, counts), 52.0 * n + 75, dt, "mfma", realised_per_atom=_counts_note(counts),
                               note="launch / latency bound: one dependent chain upload -> evaluation on the lists -> download; flops at the "
                                    "realised p, T, 700 per triplet (this build's own price)"))


def extra_lines(dev, ctx, fz, frames, batch, d_pos, d_z, d_xe, d_xf, cpu=True):
    import torch
    from uf3_amd import _lib, synthetic
    out = {}

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001 - a failing sub-line must not take the headline with it
            out[name] = dict(error=f"{type(exc).__name__}: {exc}"[:300])
        out[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    guarded("fit_c4", lambda: extra_fit(torch, dev, fz.bspline_config, frames, batch, d_pos, d_z, d_xe, d_xf))

    def fit_w():
        frames_w = [synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, [74], 3000 + k) for k in range(len(frames))]
        batch_w = _lib.FrameBatch(frames_w)
        pos_w, z_w = torch.from_numpy(batch_w.pos).to(dev), torch.from_numpy(batch_w.z).to(dev)
        return extra_fit(torch, dev, synthetic.notebook_basis(['W']), frames_w, batch_w, pos_w, z_w, d_xe, d_xf)

    guarded("fit_w", fit_w)
    guarded("lead0", lambda: extra_lead0(torch, dev, frames, d_xf))
    guarded("eval_50k", lambda: extra_eval_50k(torch, dev, cpu=cpu))
    guarded("eval_128", extra_eval_128)
    return out


def compact_configs(extra):
    """{config: [value, unit, ms_per_step, fp64_frac, hbm_frac]} of the sub-lines (same numbers as `extra`, which holds the
    details); fit_*: the fp64 fraction is ALGORITHMIC featurizer flops + EXECUTED upper-triangle Gram flops over the step"""
    table = dict(columns=["value", "unit", "ms_per_step", "fp64_frac", "hbm_frac"],
                 note="fp64_frac / hbm_frac = algorithmic flops / bytes of SURVEY 8d over 78.6 TF / 8 TB/s; fit_*: algorithmic "
                      "featurizer flops + executed upper-triangle Gram flops; eval_128 is a latency (us per call, lower is better)")
    for name, line in extra.items():
        if "error" in line:
            table[name] = ["error", line["error"][:80], None, None, None]
            continue
        r = line.get("roofline", {})
        table[name] = [line.get("value"), line.get("unit"), line.get("ms_per_step"),
                       r.get("fp64", {}).get("frac"), r.get("hbm", {}).get("frac")]
        if "cpu_baseline" in line:
            table[name + "_cpu_port"] = [line["cpu_baseline"]["value"], line["cpu_baseline"]["unit"], None, None, None]
    return table


def reference_estimate(port_one_core, port_all, cores):
    """What the NumPy reference itself would do, as an ESTIMATE: the port's rate on this box divided by the reference : port
    ratio calibrated in the build container (profiles/reference_calibration.json, tools/experiments/reference_timing.py --
    the reference cannot run 10k-atom frames).  None when the calibration file is absent."""
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "reference_calibration.json")))
        ratio = float(cal["reference_to_port_ratio"])
    except (OSError, KeyError, ValueError):
        return None
    return dict(label="estimate", reference_to_port_ratio=ratio, calibrated_at_atoms=cal.get("ratio_taken_at_atoms"),
                frames_per_s_one_process=port_one_core / ratio, frames_per_s_all_cores=port_all / ratio, cores=cores,
                source="profiles/reference_calibration.json: reference (numba absent: jitted loops interpreted, pessimistic) "
                       "vs oracle/uf3_oracle.c on identical small frames, one process each")


def _cpu_worker(job):
    """one frame of the workload through the CPU restatement (a process of the cpu_baseline pool)"""
    reps, seed, elements, numbers, lead3, fit = job
    from oracle import oracle as O
    from uf3_amd import synthetic
    ob = O.OracleBasis(synthetic.notebook_basis(elements, lead3=lead3))
    frame = synthetic.lattice_frame("bcc", reps, 3.165, numbers, seed)
    t0 = time.time()
    ref = O.featurize(ob, frame)
    if fit:
        xf = ref["xf"].reshape(-1, ref["xf"].shape[-1])
        _ = xf.T @ xf, xf.T @ np.ones(len(xf))
    return t0, time.time()


if __name__ == "__main__":
    main()
