#!/bin/bash
# tools/footprint_probe.sh: the headline bench with the same 262 144 sources drawing on fewer and fewer distinct clips -- the bytes a
# callback fetches stay the same (every source has its own window position), the footprint they are spread over shrinks from 64 GiB
# to 16 MiB.  Separates what HBM itself costs the mix kernel from what the address translation of a 64-GiB random pattern costs.
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in 262144 65536 16384 4096 1024 256 64; do
  for rep in 1 2; do
    python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-buffered --sustained 0 --clips $c 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('clips %7d (%6.2f GiB)  step %.4f  mix %.4f ms' % ($c, $c * 65536 * 4 / 2**30, j['ms_per_step'], r['avg_kernel_ms']))"
  done
done
