# GPU timeline of one host-inclusive fit call (uf3_fit_add over 128 frames of 10 k atoms): kernels and copies in start order, idle gaps
export TMPDIR=/tmp PYTHONPATH=$PWD
rm -rf gpurun_out/fittr; mkdir -p gpurun_out/fittr
NATIVE=1 NF=128 ONLY=${ONLY:-W} timeout 250 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/fittr -o t --output-format csv -- python tools/experiments/fit_host.py 2>&1 | grep "rep"
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/fittr/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:46]) for r in csv.DictReader(open(f))]
for f in glob.glob("gpurun_out/fittr/**/*memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")) for r in csv.DictReader(open(f))]
rows.sort()
# the last call: from the last gap longer than 2 ms backwards... take the final 128-frame call = everything after the last long pause
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 8):i]) > 1_500_000: cut = i
rows = rows[cut:]
t0 = rows[0][0]
busy_end = rows[0][0]
idle = 0
for s, e, n in rows:
    gap = s - busy_end
    if not n.startswith("COPY"):
        if gap > 0: idle += gap
        flag = f"   <-- idle {gap / 1e3:.0f} us" if gap > 30_000 else ""
        if e - s > 100_000 or flag: print(f"{(s - t0) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  {n}{flag}")
        busy_end = max(busy_end, e)
    else:
        print(f"{(s - t0) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  {n}")
print(f"span {(max(r[1] for r in rows) - t0) / 1e6:.2f} ms, kernel-idle {idle / 1e6:.2f} ms")
PY
