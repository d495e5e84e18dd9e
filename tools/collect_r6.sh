#!/bin/bash
# Copies what tools/profile_r6.sh left under gpurun_out/$TAG into profiles/ (tracked) under the names DESIGN.md, README.md and
# profiles/README.md use.  usage: tools/collect_r5.sh [tag] [name]      (here, after the gpurun call has merged gpurun_out/ back)
set -u
TAG=${1:-r6final}
NAME=${2:-final}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/gpurun_out/$TAG
DST=$ROOT/profiles
cpy() { if [ -s "$SRC/$1" ]; then cp "$SRC/$1" "$DST/$2"; echo "  $2"; else echo "  MISSING $1" >&2; fi; }
cpy bench_driver_protocol.json        r06_${NAME}_bench_driver_protocol.json
cpy bench_driver_protocol_run2.json   r06_${NAME}_bench_driver_protocol_run2.json
cpy bench_driver_protocol_run3.json   r06_${NAME}_bench_driver_protocol_run3.json
cpy bench_tile_kernel.json            r06_${NAME}_bench_tile_kernel.json
cpy ab_pair.txt                       r06_${NAME}_ab_pair_vs_tile.txt
cpy ab_pair_l2_resident.txt           r06_${NAME}_ab_pair_vs_tile_l2_resident.txt
cpy bench_no_precondition.json        r06_${NAME}_bench_no_precondition.json
cpy config2_4096.json                 r06_${NAME}_config2_4096.json
cpy config4_shape_65536.json          r06_${NAME}_config4_shape_65536.json
cpy buffered_driver_protocol.json     r06_${NAME}_buffered_driver_protocol.json
cpy two_ranks_one_gpu.json            r06_${NAME}_two_ranks_one_gpu.json
cpy bench_under_rocprof.json          r06_${NAME}_bench_under_rocprof.json
cpy prof/final_kernel_stats.csv       r06_${NAME}_kernel_stats.csv
cpy mix_launches.csv                  r06_${NAME}_mix_launches.csv
cpy mix_launches.txt                  r06_${NAME}_mix_launches.txt
cpy pmc/summary.json                  r06_${NAME}_pmc_summary.json
cpy pmc_latest.json                   pmc_latest.json
cpy pmc_buffered/summary.json         r06_${NAME}_buffered_pmc_summary.json
cpy pmc_buffered_latest.json          pmc_buffered_latest.json
cpy ordered_probe.txt                 r06_${NAME}_ordered_probe.txt
cpy seek_kinds.txt                    r06_${NAME}_seek_kinds.txt
cpy bench_general.txt                 r06_${NAME}_general_paths.txt
cpy leaf_scale_65536.txt              r06_${NAME}_leaf_scale_65536.txt
cpy mixer_scale.txt                   r06_${NAME}_mixer_scale.txt
cpy bench_fused_walk.json             r06_${NAME}_bench_fused_walk.json
cpy ab_fused_walk.txt                 r06_${NAME}_ab_fused_walk.txt
cpy two_ranks_sharded_65536.json      r06_${NAME}_two_ranks_sharded_65536.json
cpy pytest_gpu.log                    r06_${NAME}_pytest_gpu.log
