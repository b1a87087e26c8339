export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf gpurun_out/tr_$lib; STEPS=100 UF3_LIB_PATH=/root/repo/uf3_amd/csrc/libuf3hip_$lib.so timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/tr_$lib -o t --output-format csv -- python tools/experiments/md_eval.py 50k 2>&1 | grep "eval_\|Error"
  python - <<PY
import csv,glob,re
for f in glob.glob("gpurun_out/tr_$lib/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print("$lib", r["Name"][:50], r["Calls"], r["AverageNs"])
PY
done
