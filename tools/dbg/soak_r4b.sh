#!/bin/bash
# soak of the round-4 paths (staged Cycle tiles, stereo Downmix windows, Mixer rows): ORDERED at the default serial threshold (32:
# most seeds cross it both ways), FAST, two-kernel ORDERED everywhere, one-wave ORDERED everywhere, bounds-checked build
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tests/soak_fuzz.py 30000 ${1:-500} 2>&1 | tail -3
ODDIO_FUZZ_MODE=fast python tests/soak_fuzz.py 31000 ${2:-300} 2>&1 | tail -3
ODDIO_HIP_ORDERED_SERIAL_MAX=0 python tests/soak_fuzz.py 32000 ${2:-300} 2>&1 | tail -3
ODDIO_HIP_ORDERED_SERIAL_MAX=1000000 python tests/soak_fuzz.py 34000 ${2:-300} 2>&1 | tail -3
ODDIO_HIP_LIB=$R/oddio_amd/libodd_hip_debug.so python tests/soak_fuzz.py 33000 ${2:-300} 2>&1 | tail -3
