#!/bin/bash
# A/B of forward-kernel builds on ONE device in one session: tools/ab_run.sh <rounds> <binary> <binary> ...
# every binary is tools/fwd_ablate.hip built with another -D; the rounds are interleaved (cdna_hip_programming.md rule 24).
rounds=$1; shift
for r in $(seq $rounds); do
  for b in "$@"; do
    echo -n "$b: "; timeout -k 10 60 $b 365000 1024 10 1 || exit 1
  done
done
