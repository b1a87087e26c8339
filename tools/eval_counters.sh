set -u
export TMPDIR=/tmp
RUN=gpurun_out/evpmc; mkdir -p $RUN
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $RUN/p1 -o p --output-format csv -- python bench.py --mode eval --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES -d $RUN/p2 -o p --output-format csv -- python bench.py --mode eval --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $RUN/tr -o t --output-format csv -- python bench.py --mode eval --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/evpmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "k_eval" not in k and "k_prepare" not in k and "k_frame" not in k and "k_bin" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    print(k, "per atom:", "  ".join(f"{c[3:]} {v / max(n[k].values()) / 50000:.0f}" for c, v in sorted(acc[k].items())))
for f in glob.glob("gpurun_out/evpmc/tr/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
