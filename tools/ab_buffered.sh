#!/bin/bash
# tools/ab_buffered.sh <rounds> "<ENV=..>" ...: bench.py --workload buffered under each environment in turn (see ab_env.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
OUT=$(mktemp)
for r in $(seq 1 $N); do
  i=0
  for envs in "$@"; do
    env $envs python $R/bench.py --workload buffered --steps 20 --warmup 5 --no-cpu-baseline ${ABARGS:-} 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print($i, j['ms_per_step'], r['walk_ms'], r['write_ms'], r['read_ms'])" >> $OUT
    i=$((i+1))
  done
done
python - "$OUT" "$@" <<'PY'
import sys, statistics as st
rows = [l.split() for l in open(sys.argv[1])]
for i, name in enumerate(sys.argv[2:]):
    sel = [r for r in rows if int(r[0]) == i]
    if not sel: continue
    med = lambda k: st.median(float(r[k]) for r in sel)
    print("%-60s step %.4f  walk %.4f  write %.4f (min %.4f)  reads %.4f   n=%d" % (name or "(default)", med(1), med(2), med(3), min(float(r[3]) for r in sel), med(4), len(sel)))
PY
