#!/bin/bash
# Round-6 last evidence run (one gpurun call): the GPU suite on the tree with spatial_mix_pair<.., LANE16>, the same-box A/B of callbacks of
# 16 k < 1024 frames with the LANE16 instantiations off / on, the driver's default bench line, the PMC passes of the mix kernels (their
# sources changed: pmc_latest.json's stamp).  Everything lands under gpurun_out/lane16.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lane16
mkdir -p "$OUT"
cd "$ROOT"
(timeout 360 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -4 "$OUT/pytest_gpu.log")
SIZES=528,640,768,960,1008,1024
MODES=FAST,TRACKED ODDIO_HIP_LANE16=0 timeout 120 python tools/modes_by_callback.py 262144 $SIZES > "$OUT/ab_lane16_off.txt" 2>&1
MODES=FAST,TRACKED ODDIO_HIP_LANE16=1 timeout 120 python tools/modes_by_callback.py 262144 $SIZES > "$OUT/ab_lane16_on.txt" 2>&1
cat "$OUT/ab_lane16_off.txt" "$OUT/ab_lane16_on.txt"
timeout 150 python bench.py > "$OUT/bench_default_run.json" 2> "$OUT/bench_default_run.err"; echo "bench rc=$?"
timeout 300 tools/profile_pmc.sh lane16/pmc --steps 8 --warmup 2 --no-cpu-baseline --no-buffered --precondition-ms 0 --sustained 0 > "$OUT/pmc.log" 2>&1
python tools/make_pmc_json.py "$OUT/pmc/summary.json" 262144 "$OUT/pmc_latest.json" >> "$OUT/pmc.log" 2>&1
tail -2 "$OUT/pmc.log"
rm -rf "$OUT"/pmc/*/*.db "$OUT"/pmc/*/*/*.db 2>/dev/null
du -sh "$OUT"
