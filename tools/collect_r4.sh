#!/bin/bash
# Copies what tools/profile_r4.sh left under gpurun_out/$TAG into profiles/ (tracked) under the names DESIGN.md, README.md and
# profiles/README.md use.  usage: tools/collect_r4.sh [tag]      (here, after the gpurun call has merged gpurun_out/ back)
set -u
TAG=${1:-r4final}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/gpurun_out/$TAG
DST=$ROOT/profiles
cpy() { if [ -s "$SRC/$1" ]; then cp "$SRC/$1" "$DST/$2"; echo "  $2"; else echo "  MISSING $1" >&2; fi; }
cpy bench_driver_protocol.json        r04_final_bench_driver_protocol.json
cpy bench_driver_protocol_run2.json   r04_final_bench_driver_protocol_run2.json
cpy bench_driver_protocol_run3.json   r04_final_bench_driver_protocol_run3.json
cpy bench_no_precondition.json        r04_final_bench_no_precondition.json
cpy config2_4096.json                 r04_config2_4096.json
cpy config4_shape_65536.json          r04_config4_shape_65536.json
cpy buffered_driver_protocol.json     r04_buffered_driver_protocol.json
cpy bench_under_rocprof.json          r04_final_bench_under_rocprof.json
cpy prof/final_kernel_stats.csv       r04_final_kernel_stats.csv
cpy mix_launches.csv                  r04_final_mix_launches.csv
cpy mix_launches.txt                  r04_final_mix_launches.txt
cpy pmc/summary.json                  r04_final_pmc_summary.json
cpy pmc_latest.json                   pmc_latest.json
cpy pmc_buffered/summary.json         r04_buffered_pmc_summary.json
cpy pmc_buffered_latest.json          pmc_buffered_latest.json
cpy ordered_probe.txt                 r04_final_ordered_probe.txt
cpy seek_kinds.txt                    r04_seek_kinds.txt
cpy seek_kinds_4096.txt               r04_seek_kinds_4096.txt
cpy bench_general.txt                 r04_general_paths.txt
