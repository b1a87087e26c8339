#!/bin/bash
# One A/B round on the GPU box: a fast parity subset, the headline bench without the extras, optionally the per-atom
# instruction counters of the matrix-core launch.
#     gpurun --timeout 600 -- 'bash tools/quick_ab.sh gpurun_out/ab1 [pmc] [extra bench args]'
set -u
RUN=${1:?output directory}; PMC=${2:-}; shift; shift || true
mkdir -p "$RUN"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "reference_capture or full_size_properties or matrix_core_and_generic or random_bases or atoms_outside or mid_size or ragged_batch_of_large" > $RUN/pytest.log 2>&1
tail -3 $RUN/pytest.log
UF3_DEBUG_LDS=1 timeout 200 python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 10 --warmup 3 "$@" > $RUN/bench.json 2> $RUN/bench.err
grep "uf3 featurize mode" $RUN/bench.err | sort | uniq -c | head -4
python - "$RUN" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print("frames/s", d["value"], "ms/step", d["ms_per_step"], "launch_ms", d["roofline"].get("launch_ms"))
PY
if [ "$PMC" = "pmc" ]; then
  UF3_BENCH_NOCHECK=1 timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
    -d $RUN/pmc -o p --output-format csv -- python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 2 --warmup 1 --frames-per-step ${UF3_PMC_FRAMES:-32} "$@" > /dev/null 2>&1
  python - "$RUN" <<'PY'
import csv, glob, sys, collections
run = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(f"{run}/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_featurize" not in k: continue
        mode = "pair" if ", 0, " in k else "trio"
        acc[mode][r["Counter_Name"]] += float(r["Counter_Value"]); n[mode][r["Counter_Name"]] += 1
for mode in acc:
    launches = max(n[mode].values())
    atoms = float(__import__("os").environ.get("UF3_PMC_ATOMS", "320000"))      # (32 frames of 10 k atoms per launch)
    print(mode, "per atom:", "  ".join(f"{k[3:]} {v / launches / atoms:.0f}" for k, v in sorted(acc[mode].items())))
PY
fi
