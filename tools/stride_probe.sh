#!/bin/bash
# tools/stride_probe.sh: the headline bench with clips of 65 536 samples (a 256-KiB = 2^18-byte stride between the sources' windows) and of
# a few lengths that are not powers of two -- the same sources, windows and bytes; only the addresses' low bits stop coinciding.
R=${GRAFT_REPO_ROOT:-/root/repo}
for L in 65536 66560 65600 69632 65536 66560; do
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-buffered --sustained 0 --clip-len $L 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('clip-len %6d (stride %7d B)  step %.4f  mix %.4f ms  frac %.3f' % ($L, 4 * $L, j['ms_per_step'], r['avg_kernel_ms'], r['frac']))"
done
