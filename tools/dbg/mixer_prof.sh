#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-4096}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -o mp -- python $R/tools/bench_mixer_scale.py --sources $S > /tmp/mp.log 2>&1
grep sources /tmp/mp.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/mp/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    if 'at::native' in r['Name'] or 'rocclr' in r['Name']: continue
    print('%-70s calls %6s avg_us %9.2f total_ms %9.3f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
