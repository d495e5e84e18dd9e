#!/bin/bash
# Soak of spatial_mix_pair<.., LANE16> (GPU box): tests/soak_fuzz.py's scene test with every callback a multiple of 16 frames in 528..1024 and small
# scenes sent through the pair kernel, in the three modes that reach it and on the bounds-checked build.  usage: tools/soak_lane16.sh <first seed> <seconds per leg>
R=${GRAFT_REPO_ROOT:-/root/repo}
F=${1:-90000}; T=${2:-40}
export ODDIO_FUZZ_FRAMES=528,640,768,960,1008,1024,1024 ODDIO_HIP_PAIR_MIN_GROUPS=1 ODDIO_SOAK_ONLY=test_random_operations_bit_exact ODDIO_SOAK_VERBOSE=1
leg() { name=$1; shift; out=$(env "$@" timeout $T python $R/tests/soak_fuzz.py $F 100000 2>&1 | grep -v amdgpu.ids); n=$(echo "$out" | grep -c "^seed [0-9]*$"); bad=$(echo "$out" | grep -c "FAILED\|fault\|Error\|error")
  echo "$name: seeds started $n (from $F), failures / errors $bad"; echo "$out" | grep "FAILED\|fault\|Error\|error" | head -3; F=$((F + 1000)); }
# (all kinds: a Cycle or Downmix source takes the scene off the pair kernel -- tools/soak_r6.sh covers those legs)
leg "FAST, Seek kinds without rows"      ODDIO_FUZZ_MODE=fast ODDIO_FUZZ_PLAIN=1
leg "FAST_UNFUSED, plain kinds"          ODDIO_FUZZ_MODE=unfused ODDIO_FUZZ_PLAIN=1
leg "TRACKED, plain kinds"               ODDIO_FUZZ_MODE=tracked ODDIO_FUZZ_PLAIN=1
leg "TRACKED, plain kinds, bounds build" ODDIO_FUZZ_MODE=tracked ODDIO_FUZZ_PLAIN=1 ODDIO_HIP_LIB=$R/oddio_amd/libodd_hip_debug.so
