#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mg && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mg -o mg -- python $R/tools/dbg/mixer_gen_prof.py ${1:-65536} > /tmp/mg.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/mg/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    if 'at::native' in r['Name'] or 'rocclr' in r['Name']: continue
    print('%-70s calls %6s avg_us %9.2f total_ms %9.3f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
