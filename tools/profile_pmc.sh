#!/bin/bash
# PMC passes for the mix kernel (run on the GPU box through gpurun).  Counters are collected in
# their own runs, separate from --kernel-trace --stats, one group per pass (SQ has 8 slots,
# TCC 4: FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/profile_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="${*:---steps 8 --warmup 2 --no-cpu-baseline}"
run_pass() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o "$name" -- python "$ROOT/bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE TCC_HIT TCC_MISS
run_pass tlb TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST GRBM_GUI_ACTIVE
python "$ROOT/tools/summarize_pmc.py" "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
