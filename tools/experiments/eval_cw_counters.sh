#!/bin/bash
# SQ counters per atom of the evaluator's MD + TAB kernel with the coefficient rows out of LDS (CW instances) and out of global memory
# (UF3_EVAL_NO_CW=1): 50 000-atom ternary frame (tools/experiments/md_eval.py).   gpurun -- 'bash tools/experiments/eval_cw_counters.sh gpurun_out/cw'
set -u
RUN=${1:?output directory}; mkdir -p $RUN; export TMPDIR=/tmp STEPS=12
for v in cw nocw; do
  if [ $v = nocw ]; then export UF3_EVAL_NO_CW=1; else unset UF3_EVAL_NO_CW; fi
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD \
      -d $RUN/$v -o p --output-format csv -- python tools/experiments/md_eval.py 50k > /dev/null 2>&1
done
python - "$RUN" <<'PY'
import csv, glob, collections, sys
for v in ("cw", "nocw"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(f"{sys.argv[1]}/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_eval<" not in r["Kernel_Name"] or ", true, true" not in r["Kernel_Name"].replace("(bool)1", "true"): 
                if "k_eval<" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"]
            if "collect" in k: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(v, "per atom:", "  ".join(f"{c[3:]} {x / n[c] / 50000:.0f}" for c, x in sorted(acc.items())), " launches", max(n.values()) if n else 0)
PY
