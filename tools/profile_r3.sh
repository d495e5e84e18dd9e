#!/bin/bash
# Round-3 evidence run (on the GPU box through gpurun): bench lines (the driver's protocol three times: README quotes
# the median), rocprofv3 kernel stats of the same command, PMC passes, the ORDERED-mode stage times, micro-benchmarks.
# Everything lands under gpurun_out/$TAG; the summaries to keep are copied to profiles/ (tools/collect_r3.sh).
set -u
TAG=${1:-r3final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_protocol.json" 2> "$OUT/bench_driver_protocol.err"
for i in 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_protocol_run$i.json" 2>/dev/null; done
python bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --precondition-ms 0 > "$OUT/bench_no_precondition.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o final -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.log"
cd "$ROOT"
python tools/timed_launches.py "$OUT/prof/final_kernel_trace.csv" 20 16 > "$OUT/mix_launches.csv" 2> "$OUT/mix_launches.txt"
tools/profile_pmc.sh $TAG/pmc --steps 8 --warmup 2 --no-cpu-baseline --precondition-ms 0 > "$OUT/pmc.log" 2>&1
python tools/make_pmc_json.py "$OUT/pmc/summary.json" 262144 "$OUT/pmc_latest.json" >> "$OUT/pmc.log" 2>&1
python tools/ordered_probe.py > "$OUT/ordered_probe.txt" 2>/dev/null
make -C tools/ubench -s all > /dev/null 2>&1
./tools/ubench/dma_probe > "$OUT/ubench_dma_probe.txt" 2>&1
./tools/ubench/occ_probe > "$OUT/ubench_occ_probe.txt" 2>&1
./tools/ubench/dpp_chain > "$OUT/ubench_dpp_chain.txt" 2>&1
python tools/bench_general.py > "$OUT/bench_general.txt" 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_general" -o general -- python "$ROOT/tools/bench_general.py" > /dev/null 2> "$OUT/prof_general.log")
ls "$OUT"
