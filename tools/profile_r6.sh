#!/bin/bash
# Round-6 evidence run (on the GPU box through gpurun): bench lines (the driver's protocol three times, once with spatial_mix_pair off), the
# two smaller configurations, the buffered workload by itself, rocprofv3 kernel stats of the driver's command, PMC passes of both
# workloads, the Seek-kind, general-path and leaf-scale tables.  Everything lands under gpurun_out/$TAG; tools/collect_r6.sh copies what is kept.
set -u
TAG=${1:-r6final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_protocol.json" 2> "$OUT/bench_driver_protocol.err"
for i in 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_protocol_run$i.json" 2>/dev/null; done
ODDIO_HIP_PAIR=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-buffered > "$OUT/bench_tile_kernel.json" 2>/dev/null
tools/ab_env.sh 4 "ODDIO_HIP_PAIR=1" "ODDIO_HIP_PAIR=0" > "$OUT/ab_pair.txt" 2>&1
ABARGS="--clips 64" tools/ab_env.sh 2 "ODDIO_HIP_PAIR=1" "ODDIO_HIP_PAIR=0" > "$OUT/ab_pair_l2_resident.txt" 2>&1
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --precondition-ms 0 > "$OUT/bench_no_precondition.json" 2>/dev/null
python bench.py --sources 4096 --steps 20 --warmup 5 > "$OUT/config2_4096.json" 2>/dev/null
python bench.py --sources 65536 --steps 20 --warmup 5 > "$OUT/config4_shape_65536.json" 2>/dev/null
python bench.py --workload buffered --steps 20 --warmup 5 > "$OUT/buffered_driver_protocol.json" 2>/dev/null
python bench.py --gpus 2 --share-devices --sources 4096 --clip-len 65536 --steps 4 --warmup 2 --no-cpu-baseline > "$OUT/two_ranks_one_gpu.json" 2> "$OUT/two_ranks_one_gpu.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o final -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --sustained 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.log"
cd "$ROOT"
# launches of spatial_mix_pair<true, true, ..> after the timed region: 8 (stage timing) + 2 + 6 (host output) + 3 + 16 (TRACKED's first pass is this instantiation)
python tools/timed_launches.py "$OUT/prof/final_kernel_trace.csv" 20 35 > "$OUT/mix_launches.csv" 2> "$OUT/mix_launches.txt"
tools/profile_pmc.sh $TAG/pmc --steps 8 --warmup 2 --no-cpu-baseline --no-buffered --precondition-ms 0 --sustained 0 > "$OUT/pmc.log" 2>&1
python tools/make_pmc_json.py "$OUT/pmc/summary.json" 262144 "$OUT/pmc_latest.json" >> "$OUT/pmc.log" 2>&1
tools/profile_pmc.sh $TAG/pmc_buffered --workload buffered --steps 8 --warmup 2 --no-cpu-baseline --precondition-ms 20 > "$OUT/pmc_buffered.log" 2>&1
python tools/make_pmc_json.py "$OUT/pmc_buffered/summary.json" 262144 "$OUT/pmc_buffered_latest.json" buffered >> "$OUT/pmc_buffered.log" 2>&1
ODDIO_HIP_FUSED_WALK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-buffered > "$OUT/bench_fused_walk.json" 2>/dev/null
tools/ab_env.sh 3 "ODDIO_HIP_FUSED_WALK=0" "ODDIO_HIP_FUSED_WALK=1" > "$OUT/ab_fused_walk.txt" 2>&1
python bench.py --gpus 2 --share-devices --mode sharded --reduce p2p --sources 65536 --clip-len 65536 --steps 8 --warmup 2 --no-cpu-baseline > "$OUT/two_ranks_sharded_65536.json" 2> "$OUT/two_ranks_sharded_65536.err"
(python -m pytest tests -m gpu -q -x -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log")
python tools/ordered_probe.py > "$OUT/ordered_probe.txt" 2>/dev/null
python tools/bench_seek_kinds.py > "$OUT/seek_kinds.txt" 2>/dev/null
python tools/bench_general.py > "$OUT/bench_general.txt" 2>&1
python tools/bench_general.py --scale 65536 > "$OUT/leaf_scale_65536.txt" 2>&1
python tools/bench_mixer_scale.py > "$OUT/mixer_scale.txt" 2>&1
ls "$OUT"
