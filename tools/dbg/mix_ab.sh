#!/bin/bash
# A/B of spatial_mix builds on the headline workload: tools/dbg/mix_ab.sh "<lib>:<ENV=..>" ...   (empty lib = the product library)
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "$@"; do
  lib=${spec%%:*}; envs=${spec#*:}; [ "$envs" == "$spec" ] && envs=""
  echo "== $spec"
  env ${lib:+ODDIO_HIP_LIB=$R/oddio_amd/$lib} $envs python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-buffered 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('ms/step %.4f mix %.4f frac %.3f frac_cb %.3f prepass %.4f reduce %.4f' % (j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['frac_callback'], r['prepass_ms'], r['reduce_ms']))"
done
