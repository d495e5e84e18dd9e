#!/bin/bash
# A/B of buffered_write builds: tools/dbg/bw_ab.sh "<lib>:<ENV=..>" ...   (empty lib = the product library)
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  lib=${spec%%:*}; envs=${spec#*:}; [ "$envs" == "$spec" ] && envs=""
  echo "== $spec"
  env ${lib:+ODDIO_HIP_LIB=$R/oddio_amd/$lib} $envs python $R/bench.py --workload buffered --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('ms/step %.4f frac %.3f walk %.4f write %.4f (%.3f) read %.4f (%.3f)' % (j['ms_per_step'], r['frac'], r['walk_ms'], r['write_ms'], r['write_frac'], r['read_ms'], r['read_frac']))"
done
