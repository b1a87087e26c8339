export TMPDIR=/tmp PYTHONPATH=$PWD
rm -rf gpurun_out/mdtr; mkdir -p gpurun_out/mdtr
timeout 250 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/mdtr -o t --output-format csv -- python tools/experiments/md_step_hist.py 2>&1 | tail -2
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/mdtr/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:34]) for r in csv.DictReader(open(f))]
for f in glob.glob("gpurun_out/mdtr/**/*memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[12:]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_sup_reverse")]
i = idx[len(idx) // 2]
j = i - 12
t0 = rows[j][0]
prev = None
for s, e, n in rows[j:i + 22]:
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:6.1f} us  gap {((s - prev) / 1e3) if prev else 0:7.1f}  {n}")
    prev = e
PY
