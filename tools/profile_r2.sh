#!/bin/bash
# Round-2 evidence run (on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the same command,
# PMC passes, micro-benchmarks.  Everything lands under gpurun_out/$TAG; the summaries to keep are copied to profiles/.
set -u
TAG=${1:-r2final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_protocol.json" 2> "$OUT/bench_driver_protocol.err"
python bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --precondition-ms 0 > "$OUT/bench_no_precondition.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o final -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.log"
cd "$ROOT"
tools/profile_pmc.sh $TAG/pmc --steps 8 --warmup 2 --no-cpu-baseline --precondition-ms 0 > "$OUT/pmc.log" 2>&1
python tools/make_pmc_json.py "$OUT/pmc/summary.json" 262144 "$OUT/pmc_latest.json" >> "$OUT/pmc.log" 2>&1
make -C tools/ubench -s all > /dev/null 2>&1
./tools/ubench/hbm_ceiling > "$OUT/ubench_hbm_ceiling.txt" 2>&1
./tools/ubench/dma_probe > "$OUT/ubench_dma_probe.txt" 2>&1
./tools/ubench/valu_rate > "$OUT/ubench_valu_rate.txt" 2>&1
python tools/bench_general.py > "$OUT/bench_general.txt" 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_general" -o general -- python "$ROOT/tools/bench_general.py" > /dev/null 2> "$OUT/prof_general.log")
python tools/event_overhead.py > "$OUT/event_overhead.txt" 2>&1
for re in 8 32 320; do python bench.py --no-cpu-baseline --steps 128 --warmup 128 --reset-every $re 2>/dev/null | python tools/brief.py "128+128 callbacks, motion reset every $re:"; done > "$OUT/workload_drift.txt"
ls "$OUT"
