#!/bin/bash
# tools/ab_env.sh <rounds> "<ENV=.. ENV=..>" "<ENV=..>" ...: the headline bench under each environment in turn, <rounds> times over
# (alternating runs: boxes drift by a few percent over minutes), then the median mix-kernel and callback times per environment.
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
OUT=$(mktemp)
for r in $(seq 1 $N); do
  i=0
  for envs in "$@"; do
    env $envs python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-buffered ${ABARGS:-} 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print($i, j['ms_per_step'], r['avg_kernel_ms'], r['prepass_ms'], r.get('reduce_ms', 0))" >> $OUT
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
    print("%-44s step %.4f  mix %.4f (min %.4f)  pre %.4f  red %.4f   n=%d" % (name or "(default)", med(1), med(2), min(float(r[2]) for r in sel), med(3), med(4), len(sel)))
PY
