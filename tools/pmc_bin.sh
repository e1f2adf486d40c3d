#!/bin/bash
# usage: tools/pmc_bin.sh <outdir> <binary> [args...]  - clock + MFMA-busy counters for one binary
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d "$out/pass1" -- "$@" > "$out/pass1.log" 2>&1 || { echo "pass failed"; tail -5 "$out/pass1.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$@" > "$out/trace.log" 2>&1 || { echo "trace failed"; exit 1; }
echo ok
