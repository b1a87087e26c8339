export TMPDIR=/tmp PYTHONPATH=$PWD
mkdir -p gpurun_out/evtr
UF3_MD_SKIN=${UF3_MD_SKIN:-0} timeout 200 rocprofv3 --kernel-trace -d gpurun_out/evtr -o t --output-format csv -- python tools/experiments/eval_trace.py 4 2>&1 | tail -1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/evtr/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows[:12]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{r["Kernel_Name"][:40]:40s} start {(s - t0) / 1e3:8.2f} us  dur {(e - s) / 1e3:6.2f} us  gap {((s - prev_end) / 1e3) if prev_end else 0:6.2f}')
    prev_end = e
PY
python tools/experiments/eval_trace.py 4 | tail -1
