#!/bin/bash
# Round-6 soak (GPU box): tests/soak_fuzz.py over fresh seeds in every sum mode, with small scenes sent through spatial_mix_pair
# (ODDIO_HIP_PAIR_MIN_GROUPS=1: callbacks of 513..1024 frames), through the two-kernel ORDERED path, and on the bounds-checked build.
# usage: tools/soak_r5.sh <first seed> <seeds per leg>      -> one summary line per leg
R=${GRAFT_REPO_ROOT:-/root/repo}
F=${1:-20000}; N=${2:-150}
leg() { name=$1; shift; out=$(env "$@" timeout 1500 python $R/tests/soak_fuzz.py $F $N 2>&1 | grep -v amdgpu.ids | tail -4 | tr '\n' ' '); echo "$name: $out"; F=$((F + N)); }
leg "ORDERED"                          ODDIO_FUZZ_MODE=
leg "ORDERED two-kernel path"          ODDIO_FUZZ_MODE= ODDIO_HIP_ORDERED_SERIAL_MAX=0
leg "FAST, pair kernel on"             ODDIO_FUZZ_MODE=fast ODDIO_HIP_PAIR_MIN_GROUPS=1
leg "FAST_UNFUSED, pair kernel on"     ODDIO_FUZZ_MODE=unfused ODDIO_HIP_PAIR_MIN_GROUPS=1
leg "TRACKED, pair kernel on"          ODDIO_FUZZ_MODE=tracked ODDIO_HIP_PAIR_MIN_GROUPS=1
leg "TRACKED, larger scenes"           ODDIO_FUZZ_MODE=tracked ODDIO_HIP_PAIR_MIN_GROUPS=1 ODDIO_FUZZ_LIVE=400 ODDIO_FUZZ_OPS=30
leg "FAST, bounds-checked build"       ODDIO_FUZZ_MODE=fast ODDIO_HIP_PAIR_MIN_GROUPS=1 ODDIO_HIP_LIB=$R/oddio_amd/libodd_hip_debug.so
leg "TRACKED, bounds-checked build"    ODDIO_FUZZ_MODE=tracked ODDIO_HIP_PAIR_MIN_GROUPS=1 ODDIO_HIP_LIB=$R/oddio_amd/libodd_hip_debug.so
# round 6: the walk inside the mix kernel (opt-in) under the same seeds; played sources carry random wrapper nests in every leg (tests/test_hip_fuzz.py ODDIO_FUZZ_CHAINS)
leg "FAST, fused walk"                 ODDIO_FUZZ_MODE=fast ODDIO_HIP_PAIR_MIN_GROUPS=1 ODDIO_HIP_FUSED_WALK=1
leg "TRACKED, fused walk"              ODDIO_FUZZ_MODE=tracked ODDIO_HIP_PAIR_MIN_GROUPS=1 ODDIO_HIP_FUSED_WALK=1
leg "FAST, Downmix stereo windows"     ODDIO_FUZZ_MODE=fast ODDIO_HIP_PAIR_MIN_GROUPS=1 ODDIO_HIP_DOWNMIX_PRESUM=0
# round 6, after the late-fill defect: the same fuzz with a second process on the GPU -- timing that one process never produces
( while true; do python $R/bench.py --workload buffered --steps 200 --warmup 2 --no-cpu-baseline --sustained 0 > /dev/null 2>&1; done ) &
NEIGHBOUR=$!
sleep 25
leg "FAST beside a memory-bound neighbour"     ODDIO_FUZZ_MODE=fast ODDIO_HIP_PAIR_MIN_GROUPS=1
leg "ORDERED beside a memory-bound neighbour"  ODDIO_FUZZ_MODE=
kill $NEIGHBOUR 2>/dev/null; wait $NEIGHBOUR 2>/dev/null
