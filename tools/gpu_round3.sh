#!/bin/bash
# One GPU session of round 3: [tests,] bench, rocprofv3 kernel stats of the bench, PMC passes of the forward and the reduce
# kernels (one rocprofv3 run per counter set, no trace domains), kernel stats of the loss, timings and kernel stats of the training / TFT steps.  Digest: tools/profile_digest.py.
# usage: tools/gpu_round3.sh [notests|pmconly]
set -u -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round3
if [ "${1:-}" = "pmconly" ]; then
  mkdir -p $out
else
rm -rf $out; mkdir -p $out
if [ "${1:-}" != "notests" ]; then
  timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $out/tests.log 2>&1 || { echo TESTS FAILED; tail -30 $out/tests.log; exit 1; }
  tail -2 $out/tests.log
fi
timeout -k 10 400 python -u bench.py > $out/bench.json 2> $out/bench.err || { echo BENCH FAILED; tail -20 $out/bench.err; exit 1; }
cat $out/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/prof.log 2>&1 || { echo PROF FAILED; tail -5 $out/prof.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_loss -- python3 tools/bench_loss.py > $out/prof_loss.log 2>&1 || { echo LOSS PROF FAILED; tail -5 $out/prof_loss.log; exit 1; }
# widened rows: training step (reference batch size and config-5 size), TFT step
: > $out/train.jsonl
for cmd in "tools/bench_train.py" "tools/bench_config5.py --rows 65536" "tools/bench_config5.py --rows 1048576" "tools/bench_config5.py --rows 65536 --dtype bf16" "tools/bench_config5.py --rows 1048576 --dtype bf16" "tools/bench_train_kernels.py" "tools/bench_tft.py"; do
  timeout -k 10 200 python3 $cmd 2> $out/train.err | tail -1 >> $out/train.jsonl || { echo "TRAIN BENCH FAILED: $cmd"; tail -5 $out/train.err; exit 1; }
done
cat $out/train.jsonl
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -- python3 tools/bench_config5.py --rows 65536 > $out/prof_train.log 2>&1 || { echo TRAIN PROF FAILED; tail -5 $out/prof_train.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train_bf16 -- python3 tools/bench_config5.py --rows 65536 --dtype bf16 > $out/prof_train_bf16.log 2>&1 || { echo TRAIN BF16 PROF FAILED; tail -5 $out/prof_train_bf16.log; exit 1; }
fi
mkdir -p $out/pmc_fwd $out/pmc_red
pmc() {  # pmc <dir> <counters...> -- <program args...>
  local dir=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout -s KILL 150 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d "$dir" -- "$@" > "$dir.log" 2>&1 || { echo "pmc pass $dir failed"; tail -5 "$dir.log"; return 1; }
}
F="python3 tools/bench_forward.py --dtype bf16_fast --rows 3650000 --pairs 256 --iters 2"
pmc $out/pmc_fwd/pass1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -- $F || exit 1
pmc $out/pmc_fwd/pass2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES -- $F || exit 1
pmc $out/pmc_fwd/pass3 GRBM_GUI_ACTIVE GRBM_COUNT -- $F || exit 1
pmc $out/pmc_fwd/pass4 FETCH_SIZE -- $F || exit 1
pmc $out/pmc_fwd/pass5 WRITE_SIZE -- $F || exit 1
R="python3 tools/bench_reduce.py --rows 262144 --pairs 8192 --iters 3 --no-loss"
pmc $out/pmc_red/pass1 FETCH_SIZE -- $R || exit 1
pmc $out/pmc_red/pass2 WRITE_SIZE -- $R || exit 1
pmc $out/pmc_red/pass3 GRBM_GUI_ACTIVE GRBM_COUNT -- $R || exit 1
pmc $out/pmc_red/pass4 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- $R || exit 1
# loss: HBM bytes per pass (FETCH_SIZE / WRITE_SIZE in separate runs) at the config-5 total and at one GPU's share of it
for n in 36500000 4562500; do
  L="python3 tools/bench_loss.py --n $n --iters 10"
  mkdir -p $out/pmc_loss_$n
  pmc $out/pmc_loss_$n/pass1 FETCH_SIZE -- $L || exit 1
  pmc $out/pmc_loss_$n/pass2 WRITE_SIZE -- $L || exit 1
  pmc $out/pmc_loss_$n/pass3 GRBM_GUI_ACTIVE GRBM_COUNT -- $L || exit 1
done
bash tools/pmc_train.sh $out/pmc_train > $out/pmc_train.log 2>&1 || { echo "pmc_train failed"; tail -5 $out/pmc_train.log; exit 1; }
echo ROUND3 OK
