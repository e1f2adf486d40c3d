#!/bin/bash
# PMC passes over the reduce micro-benchmark: HBM bytes of reduce_update4_kernel (FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 runs, no trace domains) plus LDS / VALU activity.
# usage: tools/pmc_reduce.sh <outdir> [bench_reduce args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 tools/bench_reduce.py --iters 3 "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
done
echo done
