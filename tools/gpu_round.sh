#!/bin/bash
# One GPU session: parity tests, bench, rocprofv3 kernel stats of the bench, PMC passes of the forward kernel.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/round; rm -rf $out; mkdir -p $out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/tests.log 2>&1 || { echo TESTS FAILED; tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 400 python -u bench.py > $out/bench.json 2> $out/bench.err || { echo BENCH FAILED; tail -20 $out/bench.err; exit 1; }
cat $out/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/prof.log 2>&1 || { echo PROF FAILED; tail -5 $out/prof.log; exit 1; }
tools/pmc_forward.sh $out/pmc --rows 3650000 --pairs 256 --iters 2 || exit 1
echo ROUND OK
