#!/bin/bash
# A/B: run bench.py against each variants_*.so in the repo root (ODDIO_HIP_LIB override)
cd ${GRAFT_REPO_ROOT:-.}
for so in variants_*.so; do
  ODDIO_HIP_LIB=$PWD/$so python bench.py --no-cpu-baseline ${ABARGS:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$so', 'step_ms=%.4f mix_ms=%.4f frac=%.3f fcb=%.3f pre=%.3f red=%.3f'%(d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['frac_callback'], r['prepass_ms'], r.get('reduce_ms', 0)))"
done
