#!/bin/bash
# tools/dbg/contend.sh <reps> [ENV=..]  (CONTEND_TESTS="tests/a.py tests/b.py" overrides the test files): the buffered set's bit-exact tests <reps> times while another process keeps the GPU's memory system
# busy (the buffered bench in a loop) -- a race that needs a slow window fetch shows up as parity failures here.
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
( while true; do python $R/bench.py --workload buffered --steps 200 --warmup 2 --no-cpu-baseline --sustained 0 > /dev/null 2>&1; done ) &
LOADER=$!
sleep 25
fails=0
for i in $(seq 1 $N); do
  out=$(env "$@" python -m pytest ${CONTEND_TESTS:-$R/tests/test_hip_buffered_fast.py $R/tests/test_hip_buffered.py} -q -x -m gpu -p no:cacheprovider -k "not 65536" 2>&1)
  echo "$out" | tail -3 | grep -q "passed" && ! echo "$out" | tail -3 | grep -q "failed" || { fails=$((fails + 1)); mkdir -p $R/gpurun_out/contend; echo "$out" > $R/gpurun_out/contend/rep_$i.txt; echo "rep $i: $(echo "$out" | grep -m3 -i "fault\|FAILED\|Error" | tr '\n' ' ' | cut -c1-300)"; }
done
kill $LOADER 2>/dev/null; wait $LOADER 2>/dev/null
echo "contend [$*]: $N reps, $fails with failures"
