#!/bin/bash
# One GPU session for the widened rows (training step, fit loop, TFT, large batches, main.py leg): timings as JSON
# lines + the rocprofv3 kernel stats of the fit loop.  Results under gpurun_out/train_round/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/train_round; rm -rf $out; mkdir -p $out
timeout -k 10 200 python3 tools/bench_train.py > $out/train_step.jsonl 2> $out/err.log || { echo bench_train FAILED; tail -5 $out/err.log; exit 1; }
timeout -k 10 200 python3 tools/fit_time.py >> $out/fit_time.txt 2>> $out/err.log || { echo fit_time FAILED; exit 1; }
timeout -k 10 200 python3 tools/bench_tft.py > $out/tft_step.jsonl 2>> $out/err.log || { echo bench_tft FAILED; exit 1; }
for r in 4096 65536 1048576; do
  timeout -k 10 300 python3 tools/bench_config5.py --rows $r >> $out/config5.jsonl 2>> $out/err.log || { echo bench_config5 $r FAILED; exit 1; }
done
timeout -k 10 300 python3 tools/bench_pipeline.py --combos 4 2>> $out/err.log | grep '^{' > $out/pipeline.jsonl || { echo bench_pipeline FAILED; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/fit_time.py > $out/prof.log 2>&1 || { echo PROF FAILED; tail -5 $out/prof.log; exit 1; }
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python3 tools/rocprof_summary.py "$f" > $out/fit_kernel_stats.csv
rm -rf $out/prof
cat $out/train_step.jsonl $out/fit_time.txt $out/tft_step.jsonl $out/config5.jsonl $out/pipeline.jsonl
echo TRAIN ROUND OK
