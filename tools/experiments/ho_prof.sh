#!/bin/bash
# per-kernel times of one bench configuration under rocprofv3 (kernel trace + stats):  ho_prof.sh <outdir> [env assignments...] -- [bench args]
RUN=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift || true
mkdir -p $RUN; cd /tmp; export TMPDIR=/tmp
env "${envs[@]}" UF3_BENCH_NOCHECK=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$RUN/tr -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-traffic --steps 4 --warmup 2 "$@" > $GRAFT_REPO_ROOT/$RUN/bench.json 2> $GRAFT_REPO_ROOT/$RUN/bench.err
f=$(find $GRAFT_REPO_ROOT/$RUN/tr -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e6:9.3f} ms  total {float(r["TotalDurationNs"]) / 1e6:9.1f} ms  {r["Percentage"]}%')
PY
