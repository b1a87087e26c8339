#!/usr/bin/env python3
"""
Secondary measurements (not the bench.py contract): X^T X on the fp64 matrix cores and the
energy/force evaluator, device-resident, HIP-event timed through the library's own timers.

    python tools/bench_kernels.py [--quick]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    import torch
    from uf3_amd import _lib, synthetic
    from uf3_amd.data import composition
    from uf3_amd.representation import bspline, process
    dev = torch.device("cuda", 0)
    ctx = _lib.get_context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    out = {}

    # ---- Gram: rows of one 10k-atom frame (3N+1 = 30001 rows) at F' = 425 and a 4-frame block ------
    for rows, F in ([(30001, 425)] if args.quick else [(30001, 425), (120004, 425), (30001, 70)]):
        x = torch.randn((rows, F), dtype=torch.float64, device=dev)
        y = torch.randn((rows,), dtype=torch.float64, device=dev)
        g = torch.empty((F, F), dtype=torch.float64, device=dev)
        o = torch.empty((F,), dtype=torch.float64, device=dev)
        call = lambda: ctx.check(ctx.lib.uf3_gram_dev(ctx.handle, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),  # noqa: E731
                                                     rows, F, F, 0, C.c_void_p(g.data_ptr()), C.c_void_p(o.data_ptr())))
        call(); torch.cuda.synchronize()
        ref = x.T @ x
        err = ((g - ref).abs().max() / ref.abs().max()).item()
        ctx.timing_reset(True)
        n = 5
        for _ in range(n):
            call()
        t = ctx.timing_read()["gram_ms"] / n
        ctx.timing_reset(False)
        t0 = time.perf_counter()
        for _ in range(n):
            ref = x.T @ x
        torch.cuda.synchronize()
        t_blas = (time.perf_counter() - t0) / n * 1e3
        flops = 2.0 * rows * F * F
        out[f"gram_{rows}x{F}"] = dict(ms=round(t, 4), tflops_full=round(flops / t / 1e9, 2), rel_err=err,
                                       torch_matmul_ms=round(t_blas, 4), peak_tflops=78.6)

    # ---- evaluator: 10k-atom binary frame and (unless --quick) the 50k-atom ternary config C5 ------
    cases = [("eval_10k_binary", synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, [42, 74], 3000), synthetic.notebook_basis(['Mo', 'W']))]
    if not args.quick:
        cases.append(("eval_50k_ternary", synthetic.lattice_frame("bcc", (25, 25, 40), 3.165, [23, 42, 74], 4000),
                      synthetic.notebook_basis(['V', 'Mo', 'W'])))
    from uf3_amd.regression import least_squares as ls
    from uf3_amd.forcefield import calculator
    for name, atoms, basis in cases:
        model = ls.WeightedLinearModel(basis)
        rng = np.random.default_rng(11)
        coeff = rng.normal(0, 0.05, basis.n_feats)
        coeff[basis.col_idx] = 0.0
        model.coefficients = coeff
        calc = calculator.UFCalculator(model)
        db = _lib.device_basis(basis, ctx)
        batch = _lib.FrameBatch([atoms])
        d_pos = torch.from_numpy(batch.pos).to(dev)
        d_z = torch.from_numpy(batch.z).to(dev)
        d_e = torch.empty((1,), dtype=torch.float64, device=dev)
        d_f = torch.empty((batch.n_atoms, 3), dtype=torch.float64, device=dev)
        call = lambda: ctx.check(ctx.lib.uf3_eval_dev(db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()),  # noqa: E731
                                                     C.c_void_p(d_z.data_ptr()), _lib._p(calc._c1), _lib._p(calc._c2),
                                                     _lib._p(calc._c3), C.c_void_p(d_e.data_ptr()), C.c_void_p(d_f.data_ptr())))
        call(); torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()                      # wall clock without the library's event timing (it costs host time)
        for _ in range(n):
            call()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        ctx.timing_reset(True)
        for _ in range(n):
            call()
        torch.cuda.synchronize()
        t = ctx.timing_read()
        ctx.timing_reset(False)
        f = d_f.cpu().numpy()
        out[name] = dict(atoms=batch.n_atoms, n_feat=basis.n_feats, eval_kernel_ms=round(t["eval_ms"] / n, 4),
                         neighbor_ms=round(t["neighbor_ms"] / n, 4), wall_ms_per_step=round(wall, 4),
                         atom_steps_per_s=round(batch.n_atoms / (wall * 1e-3)), energy=float(d_e.item()),
                         net_force=float(np.abs(f.sum(axis=0)).max()), max_force=float(np.abs(f).max()))
    # ---- one large frame decomposed over ranks (N4): what ONE rank of a world of 1 / 2 / 4 / 8 spends on its block of the
    # 50k-atom ternary frame (uf3_eval_centres: cell list of the frame, every triplet centred in the block once, 3-body lists
    # of the block's halo, collection pass over block + halo).  The frame is in lattice order (x slowest), so an index block is a slab.
    if not args.quick:
        name, atoms, basis = cases[-1]
        d_v = torch.empty((6,), dtype=torch.float64, device=dev)
        dec = {}
        for world in (1, 2, 4, 8):
            lo, hi = 0, (batch.n_atoms + world - 1) // world
            call = lambda: ctx.check(ctx.lib.uf3_eval_centres_dev(  # noqa: E731
                db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()), C.c_void_p(d_z.data_ptr()),
                _lib._p(calc._c1), _lib._p(calc._c2), _lib._p(calc._c3), lo, hi, C.c_void_p(d_e.data_ptr()),
                C.c_void_p(d_f.data_ptr()), C.c_void_p(d_v.data_ptr())))
            call(); torch.cuda.synchronize()
            ctx.timing_reset(True)
            t0 = time.perf_counter()
            for _ in range(n):
                call()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e3
            t = ctx.timing_read()
            ctx.timing_reset(False)
            dec[f"world_{world}"] = dict(block_atoms=hi - lo, neighbor_ms=round(t["neighbor_ms"] / n, 4),
                                         eval_kernel_ms=round(t["eval_ms"] / n, 4), wall_ms=round(wall, 4))
        out["decomposed_50k_ternary_one_rank"] = dec
        # the same share on the MD route (round 5): whole-frame lists kept with a 0.5 A skin, the atoms moving +-0.01 A per step
        # (one device add per step, inside the timed loop like the list rebuilds)
        g = torch.Generator(device=dev).manual_seed(17)
        pool = (torch.rand((32, batch.n_atoms, 3), dtype=torch.float64, device=dev, generator=g) * 2.0 - 1.0) * 0.01
        pick = np.random.default_rng(17).integers(0, 32, 4096)
        sign = np.random.default_rng(18).choice([-1.0, 1.0], 4096)
        prev = ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ctx.md_skin(0.5)
        dec_md = {}
        step_no = [0]
        try:
            for world in (1, 2, 4, 8):
                lo, hi = 0, (batch.n_atoms + world - 1) // world

                def call():
                    k = step_no[0] & 4095
                    step_no[0] += 1
                    d_pos.add_(pool[pick[k]], alpha=float(sign[k]))
                    ctx.check(ctx.lib.uf3_eval_centres_dev(
                        db.handle, C.byref(batch.struct), C.c_void_p(d_pos.data_ptr()), C.c_void_p(d_z.data_ptr()),
                        _lib._p(calc._c1), _lib._p(calc._c2), _lib._p(calc._c3), lo, hi, C.c_void_p(d_e.data_ptr()),
                        C.c_void_p(d_f.data_ptr()), C.c_void_p(d_v.data_ptr())))
                for _ in range(5):
                    call()
                torch.cuda.synchronize()
                s0 = ctx.md_stats()
                m = 10 * n
                t0 = time.perf_counter()
                for _ in range(m):
                    call()
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / m * 1e3
                s1 = ctx.md_stats()
                dec_md[f"world_{world}"] = dict(block_atoms=hi - lo, wall_ms=round(wall, 4), steps=m,
                                                list_builds=s1["builds"] - s0["builds"], steps_repeated=s1["redone"] - s0["redone"])
        finally:
            ctx.md_skin(0.0)
            ctx.restore_stream(prev)
        out["decomposed_50k_ternary_one_rank_md_route"] = dec_md

    # ---- fit accumulation (BASELINE config 4, one GPU's share): frames -> rows -> X^T X / X^T y, rows stay in HBM ---
    if not args.quick:
        from uf3_amd import pipeline
        # (the 512-frame call: the 0.7 ms between a call's arrival and its first copy -- Python, the chunk plan, the first pack -- and
        # the ramp of small chunks are per call; a fit over a data set is one long call)
        for name, els, n_fr in (("fit_accumulate_10k_W", ['W'], 128), ("fit_accumulate_10k_WMo", ['Mo', 'W'], 128),
                                ("fit_accumulate_10k_W_512_frames", ['W'], 512)):
            basis = synthetic.notebook_basis(els)
            zs = [74] if els == ['W'] else [42, 74]
            frames = [synthetic.lattice_frame("bcc", (10, 20, 25), 3.165, zs, 5000 + k) for k in range(n_fr)]
            rng = np.random.default_rng(3)
            energies = rng.normal(-8.9 * 10000, 5.0, len(frames))
            forces = [rng.normal(0, 0.5, (len(f), 3)) for f in frames]
            model = ls.WeightedLinearModel(basis)
            fz = process.BasisFeaturizer(basis, device=0)
            acc = pipeline.DeviceFitAccumulator(model, fz)
            acc.add_frames(frames, energies, forces)                       # warm-up (capacities, staging sets, row buffers)
            torch.cuda.synchronize()
            dt = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                acc.add_frames(frames, energies, forces)
                torch.cuda.synchronize()
                dt = min(dt, time.perf_counter() - t0)
            out[name] = dict(frames=len(frames), n_feat=basis.n_feats, wall_ms=round(dt * 1e3, 2),
                             frames_per_s=round(len(frames) / dt, 1),
                             note="host frame packing into pinned staging + H2D on a copy stream + featurize + Gram of energy and force rows "
                                  "(DeviceFitAccumulator.add_frames, chunks of at most 320 000 atoms planned ahead; best of two timed calls)")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
