#!/bin/bash
# Second leg of the LANE16 evidence (one gpurun call): the A/B again with 32 callbacks per timing and the two settings alternating
# (off, on, off, on: four processes on one box), then the driver's default bench line and its rocprofv3 kernel trace on the final tree.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lane16b
mkdir -p "$OUT"
cd "$ROOT"
SIZES=640,768,960,1024
for i in 1 2; do for v in 0 1; do
  echo "## run $i ODDIO_HIP_LANE16=$v" >> "$OUT/ab_lane16.txt"
  REPS=32 MODES=FAST,TRACKED ODDIO_HIP_LANE16=$v timeout 100 python tools/modes_by_callback.py 262144 $SIZES 2>/dev/null | grep frames >> "$OUT/ab_lane16.txt"
done; done
cat "$OUT/ab_lane16.txt"
timeout 150 python bench.py > "$OUT/bench_default_run.json" 2> "$OUT/bench_default_run.err"; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o final -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-buffered --sustained 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.log"
cd "$ROOT"
python tools/timed_launches.py "$OUT"/prof/*/final_kernel_trace.csv 20 35 > "$OUT/mix_launches.csv" 2> "$OUT/mix_launches.txt" || python tools/timed_launches.py "$OUT/prof/final_kernel_trace.csv" 20 35 > "$OUT/mix_launches.csv" 2> "$OUT/mix_launches.txt"
cat "$OUT/mix_launches.txt"
find "$OUT/prof" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
