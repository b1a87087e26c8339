#!/bin/bash
# Per-phase instruction budget of the matrix-core featurizer launch: PMC passes with UF3_DEBUG_SKIP ablations
# (1 two-body, 2 centre role, 4 neighbour role, 8 MFMA steps, 16 leg evaluation + staging, 32 row stores).
#     gpurun --timeout 900 -- 'bash tools/ablate_counters.sh gpurun_out/abl "0 8 16 24 6"'
# The switches are compiled in with -DUF3_ABLATE only (build that library first, here, where hipcc is:
#     (cd uf3_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DUF3_ABLATE -shared -o ../../exp/libuf3hip_ablate.so uf3_hip.hip)
set -u
RUN=${1:?output directory}; SKIPS=${2:-"0 1 2 3 4 8 16 32 64"}
mkdir -p "$RUN"; export TMPDIR=/tmp UF3_BENCH_NOCHECK=1 UF3_LIB_PATH=$PWD/exp/libuf3hip_ablate.so
[ -f "$UF3_LIB_PATH" ] || { echo "build exp/libuf3hip_ablate.so first (see the header of this script)"; exit 1; }
for s in $SKIPS; do
  UF3_DEBUG_SKIP=$s timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
    -d $RUN/s$s -o p --output-format csv -- python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 2 --warmup 1 --frames-per-step ${UF3_ABL_FRAMES:-32} ${UF3_ABL_ARGS:-} > $RUN/s$s.json 2>/dev/null
done
python - "$RUN" $SKIPS <<'PY'
import csv, glob, sys, collections
run, skips = sys.argv[1], sys.argv[2:]
for s in skips:
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(f"{run}/s{s}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_featurize" not in k or ", 0, " in k: continue      # (the pair launch: MODE 0)
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    if not acc: print(s, "no data"); continue
    launches = max(n.values())
    per_atom = {k: v / launches / float(__import__("os").environ.get("UF3_ABL_ATOMS", "320000")) for k, v in acc.items()}
    print(f"skip {s:>3}: " + "  ".join(f"{k[3:]} {per_atom[k]:.0f}" for k in sorted(per_atom)))
PY
