"""Condense rocprofv3 CSV output (gpurun_out/<run>/...) into the small summaries kept under profiles/.

    python tools/summarize_profiles.py gpurun_out/r1c profiles/round1

Reads   <run>/trace/*_kernel_stats.csv           (rocprofv3 --kernel-trace --stats --output-format csv)
        <run>/pmc_fetch, <run>/pmc_write         (--pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)
        <run>/pmc1, <run>/pmc2, <run>/pmc3       (passes of SQ counters; pmc3 = MFMA / LDS activity)
        <run>/pmc4                               (fp64 wave-instructions by kind: the EXECUTED work, <prefix>_fp64_counters.json)
        <run>/eval_counters.txt                  (tools/md_counters.sh: the evaluator's routes)
        <run>/bench_default.json, <run>/kernels.json
Writes  <prefix>_kernel_stats.csv, <prefix>_hbm_counters.json, <prefix>_sq_counters.txt,
        <prefix>_bench_n1.json, <prefix>_secondary_kernels.json
Counters are summed over the k_featurize<...> specialisations of one step (one "launch group") and
averaged over the steps seen.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def counter_rows(directory):
    for path in glob.glob(os.path.join(directory, "*_counter_collection.csv")):
        with open(path, newline="") as fh:
            yield from csv.DictReader(fh)


def featurize_counters(directory):
    """{counter: {kernel name: [values per dispatch]}} for the featurizer kernels"""
    out = defaultdict(lambda: defaultdict(list))
    for row in counter_rows(directory):
        name = row["Kernel_Name"]
        if "k_featurize" in name:
            out[row["Counter_Name"]][name].append(float(row["Counter_Value"]))
    return out


def group_mean(per_kernel):
    """mean over steps of the sum over the specialisations launched in one step"""
    steps = max(len(v) for v in per_kernel.values())
    return sum(sum(v) for v in per_kernel.values()) / steps


def main(run, prefix):
    stats = glob.glob(os.path.join(run, "trace", "*_kernel_stats.csv"))[0]
    shutil.copy(stats, prefix + "_kernel_stats.csv")
    shutil.copy(os.path.join(run, "bench_default.json"), prefix + "_bench_n1.json")
    if os.path.exists(os.path.join(run, "kernels.json")):
        shutil.copy(os.path.join(run, "kernels.json"), prefix + "_secondary_kernels.json")
    extra = glob.glob(os.path.join(run, "trace_extra", "*_kernel_stats.csv"))
    if extra:       # kernel statistics of the default run WITH the sub-lines of the other BASELINE configurations
        shutil.copy(extra[0], prefix + "_kernel_stats_all_configs.csv")

    with open(stats, newline="") as fh:
        feat = [r for r in csv.DictReader(fh) if "k_featurize" in r["Name"]]
    # one launch group per step = the MODE 0 launch + the trio launches; the first call of a context may use a
    # different trio specialisation than the steady state (list capacity not tuned yet), so average over steps
    steps = max(int(r["Calls"]) for r in feat)
    rocprof_group_ms = sum(float(r["TotalDurationNs"]) for r in feat) / steps / 1e6
    bench = json.load(open(os.path.join(run, "bench_default.json")))
    roof = bench["roofline"]
    alg_bytes = roof["hbm"]["algorithmic_bytes_per_launch"] if "hbm" in roof else roof["algorithmic_bytes_per_launch"]

    fetch = featurize_counters(os.path.join(run, "pmc_fetch"))["FETCH_SIZE"]
    write = featurize_counters(os.path.join(run, "pmc_write"))["WRITE_SIZE"]
    fetch_kib, write_kib = group_mean(fetch), group_mean(write)
    hbm = {
        "command": "python bench.py --steps 6 --warmup 2 --no-cpu-baseline (%d frames x 10k atoms per step, F=434)"
                   % bench["config"]["frames_per_step"],
        "tool": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv)",
        "kernel": "k_featurize<E,F,R,MODE>: the specialised launches of one step, summed",
        "per_specialisation_KiB": {
            "FETCH_SIZE": {k: sum(v) / len(v) for k, v in fetch.items()},
            "WRITE_SIZE": {k: sum(v) / len(v) for k, v in write.items()},
        },
        "FETCH_SIZE_KiB_per_launch": fetch_kib,
        "WRITE_SIZE_KiB_per_launch": write_kib,
        "hbm_bytes_per_launch_raw": (fetch_kib + write_kib) * 1024,
        "hbm_bytes_per_launch_fetch_x2": (2 * fetch_kib + write_kib) * 1024,
        "algorithmic_bytes_per_launch": alg_bytes,
        "known_write_bytes_per_launch (rows + 3-body lists at capacity 16)":
            alg_bytes + 48 * 16 * bench["config"]["atoms_per_frame"] * bench["config"]["frames_per_step"],
        "kernel_time_agreement": {
            "hip_events_ms_per_step (bench.py, live)": bench["roofline"]["launch_ms"],
            "rocprofv3_kernel_stats_ms_per_step": rocprof_group_ms,
            "per_specialisation_ms (calls)": {r["Name"]: [float(r["AverageNs"]) / 1e6, int(r["Calls"])] for r in feat},
        },
        "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE can report 1/2 of the bytes of a wide coalesced "
                "streaming read; the reads here are mostly 48-B gathers of neighbour-list entries, so the "
                "uncorrected sum is quoted as `traffic` and the x2-corrected sum as the upper bound. Every "
                "specialisation reads the neighbour lists again; nothing is read-modify-written: rows are "
                "written once per (atom, column range).  WRITE_SIZE (uncalibrated per the guide) against this kernel's "
                "known writes -- rows (= the algorithmic bytes) + 3-body lists (48 B x capacity 16 per atom), each written "
                "exactly once: see write_amplification below (rows are 3472 B = 54.25 cache lines, so the column segments "
                "of a block start and end mid-line and the partial lines of neighbouring segments are written twice).",
        "write_amplification": (write_kib * 1024) / (alg_bytes + 48 * 16 * bench["config"]["atoms_per_frame"] * bench["config"]["frames_per_step"]),
        "workload": {"atoms_per_frame": bench["config"]["atoms_per_frame"], "n_feat": bench["config"]["n_feat"],
                     "frames_per_step": bench["config"]["frames_per_step"]},
    }
    json.dump(hbm, open(prefix + "_hbm_counters.json", "w"), indent=1)

    lines = ["# rocprofv3 --pmc (separate passes of <= 8 SQ counters), %d frames x 10k atoms per step (F=434)"
             % bench["config"]["frames_per_step"],
             "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; "
             "SQ_INSTS_* count wave-instructions", ""]
    sq = {}
    # executed fp64 work of one launch group: 64 lanes per wave-instruction (idle lanes included: an upper bound), 2 per FMA
    fp = featurize_counters(os.path.join(run, "pmc4"))
    if fp:
        g = {k: group_mean(v) for k, v in fp.items()}
        flops = 64.0 * (2 * g.get("SQ_INSTS_VALU_FMA_F64", 0) + g.get("SQ_INSTS_VALU_MUL_F64", 0) + g.get("SQ_INSTS_VALU_ADD_F64", 0)
                        + g.get("SQ_INSTS_VALU_TRANS_F64", 0)) + 512.0 * g.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)
        sec = bench["roofline"]["launch_ms"] * 1e-3
        json.dump({"tool": "rocprofv3 --pmc SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU (one pass)",
                   "wave_instructions_per_launch_group": g,
                   "executed_fp64_flops_per_launch": flops,
                   "executed_fp64_tflops": flops / sec / 1e12, "executed_fp64_frac_of_78.6TF": flops / sec / 1e12 / 78.6,
                   "algorithmic_fp64_flops_per_launch": bench["roofline"].get("algorithmic_flops_per_launch"),
                   "note": "flops = 64 lanes x (2 FMA + MUL + ADD + TRANS wave-instructions) + 512 per MFMA_MOPS unit: lanes switched off by the "
                           "exec mask are counted, so this is an upper bound on the useful work and the pipe's true occupancy by fp64 issue",
                   "workload": hbm["workload"]}, open(prefix + "_fp64_counters.json", "w"), indent=1)
    if os.path.exists(os.path.join(run, "eval_counters.txt")):
        shutil.copy(os.path.join(run, "eval_counters.txt"), prefix + "_eval_counters.txt")
    for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
        sq.update(featurize_counters(os.path.join(run, d)))
    kernels = sorted({k for per in sq.values() for k in per})
    for kern in kernels + ["ALL k_featurize launches of one step"]:
        lines.append("## " + kern)
        for name in sorted(sq):
            per = sq[name]
            val = group_mean(per) if kern.startswith("ALL") else (
                sum(per[kern]) / len(per[kern]) if kern in per else float("nan"))
            lines.append(f"{name:<30} {val:.4g}")
        lines.append("")
    open(prefix + "_sq_counters.txt", "w").write("\n".join(lines))
    print(json.dumps({"rocprof_group_ms": rocprof_group_ms, "hip_events_ms": bench["roofline"]["launch_ms"],
                      "traffic_raw": hbm["hbm_bytes_per_launch_raw"]}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
