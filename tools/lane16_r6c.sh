#!/bin/bash
# Third leg of the LANE16 evidence: the A/B with the two settings alternating (off, on, off, on, off, on: six processes on one box), 12 callbacks
# per timing (the scene's sources start one second into their 65 536-sample clips: 17 callbacks of 1024 frames are left).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lane16c
mkdir -p "$OUT"
cd "$ROOT"
SIZES=640,768,960,1024
for i in 1 2 3; do for v in 0 1; do
  echo "## run $i ODDIO_HIP_LANE16=$v" >> "$OUT/ab_lane16.txt"
  REPS=12 MODES=FAST,TRACKED ODDIO_HIP_LANE16=$v timeout 100 python tools/modes_by_callback.py 262144 $SIZES 2>> "$OUT/ab_lane16.err" | grep frames >> "$OUT/ab_lane16.txt"
done; done
cat "$OUT/ab_lane16.txt"; tail -3 "$OUT/ab_lane16.err"
