#!/bin/bash
# per-kernel times of the Seek-set kinds: tools/dbg/kinds_prof.sh <sources> <kinds>
R=${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-4096}; K=${2:-cycle}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o kp -- python $R/tools/bench_seek_kinds.py --sources $S --kinds $K --callbacks 16 --warm 64 > /tmp/kp.log 2>&1
grep -v simple_timer /tmp/kp.log | tail -4
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/kp/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print('%-70s calls %6s avg_us %9.2f total_ms %9.3f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
