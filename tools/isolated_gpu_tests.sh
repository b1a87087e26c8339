cd /root/repo
# every GPU test on its own (a fresh process and context each): order dependence shows here
for t in $(python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" | grep -v "two_gpu\|launcher"); do
  r=$(timeout 300 python -m pytest "$t" -m gpu -x -q 2>&1 | tail -1)
  case "$r" in *failed*|*error*) echo "FAIL $t :: $r";; esac
done
echo isolated-run-done
