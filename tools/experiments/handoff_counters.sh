#!/bin/bash
# Per-atom SQ counters and per-launch HBM traffic of the hand-off launch pair (UF3_F3_HANDOFF=1: k_feat3_w + k_featurize3<HO>)
# beside the one-kernel k_featurize3 (UF3_F3_HANDOFF=0); 32 frames of 10 k atoms per launch.
#     gpurun --timeout 900 -- 'bash tools/experiments/handoff_counters.sh gpurun_out/hoc'
set -u
RUN=${1:?output directory}; mkdir -p $RUN; export TMPDIR=/tmp UF3_BENCH_NOCHECK=1
B="python bench.py --no-cpu-baseline --no-extra --no-traffic --steps 2 --warmup 1 --frames-per-step 32"
for h in 0 1; do
  UF3_F3_HANDOFF=$h timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
      -d $RUN/sq$h -o p --output-format csv -- $B > /dev/null 2>&1
  UF3_F3_HANDOFF=$h timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES \
      -d $RUN/sqb$h -o p --output-format csv -- $B > /dev/null 2>&1
  UF3_F3_HANDOFF=$h timeout 200 rocprofv3 --pmc FETCH_SIZE -d $RUN/fetch$h -o p --output-format csv -- $B > /dev/null 2>&1
  UF3_F3_HANDOFF=$h timeout 200 rocprofv3 --pmc WRITE_SIZE -d $RUN/write$h -o p --output-format csv -- $B > /dev/null 2>&1
done
python - "$RUN" <<'PY'
import csv, glob, sys, collections
run = sys.argv[1]
ATOMS = 320000.0
for h in (0, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for d in ("sq", "sqb", "fetch", "write"):
        for f in glob.glob(f"{run}/{d}{h}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "k_feat3_w" in k: name = "k_feat3_w"
                elif "k_featurize3" in k: name = "k_featurize3" + ("<HO>" if k.rstrip(">(Feat3Args)").endswith("true") and k.count(",") >= 4 else "")
                elif "k_featurize<" in k: name = "k_featurize<MODE 0>"
                else: continue
                v = float(r["Counter_Value"])
                if v == 0 and "featurize3" in name and r["Counter_Name"] in ("SQ_INSTS_VALU",): pass
                acc[name][r["Counter_Name"]] += v; n[name][r["Counter_Name"]] += 1
    print(f"--- UF3_F3_HANDOFF={h} (per atom; FETCH/WRITE in bytes per atom = KB units x 1024 / atoms)")
    for name in sorted(acc):
        out = []
        for c in sorted(acc[name]):
            launches = n[name][c]
            v = acc[name][c] / launches
            if c in ("FETCH_SIZE", "WRITE_SIZE"): out.append(f"{c} {v * 1024 / ATOMS:.0f} B")      # (guide: FETCH/WRITE_SIZE count KiB)
            else: out.append(f"{c[3:]} {v / ATOMS:.0f}")
        print(f"{name:24s} launches/counter {max(n[name].values()):3d}: " + "  ".join(out))
PY
