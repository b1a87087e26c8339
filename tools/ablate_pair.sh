#!/bin/bash
# Per-phase instruction budget of the pair launch (k_featurize MODE 0): PMC passes with UF3_DEBUG_SKIP ablations
# (1 candidate walk -- and with it everything behind it --, 512 pair rows, 1024 3-body list build); library built with
# -DUF3_ABLATE at exp/libuf3hip_ablate.so (see tools/ablate_counters.sh).
#     gpurun --timeout 900 -- 'bash tools/ablate_pair.sh gpurun_out/ablp "0 512 1024 1536 1"'
set -u
RUN=${1:?output directory}; SKIPS=${2:-"0 512 1024 1536 1"}
mkdir -p "$RUN"; export TMPDIR=/tmp UF3_BENCH_NOCHECK=1 UF3_LIB_PATH=$PWD/exp/libuf3hip_ablate.so
[ -f "$UF3_LIB_PATH" ] || { echo "build exp/libuf3hip_ablate.so first"; exit 1; }
for s in $SKIPS; do
  UF3_DEBUG_SKIP=$s timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
    -d $RUN/s$s -o p --output-format csv -- python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 2 --warmup 1 --frames-per-step 32 > $RUN/s$s.json 2>/dev/null
done
python - "$RUN" $SKIPS <<'PY'
import csv, glob, sys, collections
run, skips = sys.argv[1], sys.argv[2:]
for s in skips:
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(f"{run}/s{s}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_featurize<" not in k or ", 0, " not in k: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    if not acc: print(s, "no data"); continue
    launches = max(n.values())
    print(f"skip {s:>4}: " + "  ".join(f"{k[3:]} {v / launches / 320000:.0f}" for k, v in sorted(acc.items())))
PY
