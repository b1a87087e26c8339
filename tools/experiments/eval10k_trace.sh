#!/bin/bash
# kernel timeline of one rebuild-everything evaluator call on the 10k-atom binary frame (tools/bench_kernels.py --quick): which launches a call is made of
export TMPDIR=/tmp; rm -rf gpurun_out/e10k; mkdir -p gpurun_out/e10k
timeout 250 rocprofv3 --kernel-trace -d gpurun_out/e10k -o t --output-format csv -- python tools/bench_kernels.py --quick > /dev/null 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/e10k/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))]
rows.sort()
# the last complete call chain ending with k_frame_sum
ends = [i for i, r in enumerate(rows) if r[2].startswith("k_frame_sum")]
last = ends[-1]; first = ends[-2] + 1
t0 = rows[first][0]
for s, e, n in rows[first:last + 1]:
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f} us  {n}")
print(f"chain: {(rows[last][1] - t0) / 1e3:.1f} us")
PY
