# SQ counters and kernel times of the evaluator's MD route against the rebuild-everything route (50 000-atom ternary frame,
# tools/experiments/md_eval.py).  UF3_LIB_PATH picks the library build.  Counters in passes of their own, no trace domains.
set -u
export TMPDIR=/tmp
RUN=gpurun_out/mdpmc; rm -rf $RUN; mkdir -p $RUN
export STEPS=${STEPS:-12}
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $RUN/p1 -o p --output-format csv -- python tools/experiments/md_eval.py 50k > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES -d $RUN/p2 -o p --output-format csv -- python tools/experiments/md_eval.py 50k > /dev/null 2>&1
STEPS=100 timeout 200 rocprofv3 --kernel-trace --stats -d $RUN/tr -o t --output-format csv -- python tools/experiments/md_eval.py 50k > $RUN/run.log 2>&1
grep eval_ $RUN/run.log
python - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
def short(k):
    m = re.match(r"(?:void )?(\w+)(<[^>]*>)?", k)
    return (m.group(1) + (m.group(2) or "")) if m else k[:50]
for f in glob.glob("gpurun_out/mdpmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not any(s in k for s in ("k_eval", "k_prepare", "k_frame", "k_bin", "k_build_sup", "k_md")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k, "per atom:", "  ".join(f"{c[3:]} {v / max(n[k].values()) / 50000:.0f}" for c, v in sorted(acc[k].items())))
for f in glob.glob("gpurun_out/mdpmc/tr/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]: print(short(r["Name"]), r["Calls"], r["AverageNs"])
PY
