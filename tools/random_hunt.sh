#!/bin/bash
# random_hunt.sh -- widen tests/gpu_random_check.py beyond what the test suite runs: several seeds, more
# cases, every forced kernel shape.  usage (on the GPU box): tools/random_hunt.sh <cases> <seed>...
cases=$1; shift
fail=0
for seed in "$@"; do
  for env in "AWPU_FAST_PAIRS=1" "AWPU_FAST_PAIRS=0 AWPU_FAST_VARIANT=1,8,32" "AWPU_FAST_PAIRS=0 AWPU_FAST_VARIANT=1,4,32" \
             "AWPU_FAST_PAIRS=0 AWPU_FAST_VARIANT=1,2,8" "AWPU_FAST_PAIRS=0 AWPU_FAST_VARIANT=1,8,8" \
             "AWPU_FAST_PAIRS=0 AWPU_FAST_VARIANT=2,4,8" "AWPU_FAST_PAIRS=0 AWPU_FAST_VARIANT=1,4,24" "AWPU_TEST_MATH=exact" "AWPU_TEST_PATH=device" "AWPU_TEST_PATH=device AWPU_FAST_PAIRS=1" "AWPU_TEST_INTERP=fir8" "AWPU_TEST_REUSE=1" "AWPU_TEST_REUSE=1 AWPU_FAST_PAIRS=1" "AWPU_TEST_REUSE=1 AWPU_TEST_MATH=exact" "AWPU_FAST_PAIRS=1 AWPU_TEST_GRID=1 AWPU_FAST_PAIRCOLS=1" "AWPU_FAST_PAIRS=1 AWPU_TEST_GRID=1 AWPU_FAST_PAIRCOLS=0" "AWPU_FAST_PAIRS=1 AWPU_TEST_GRID=1 AWPU_FAST_DEBUG=4096" "AWPU_TEST_PATH=device AWPU_TEST_GRID=1 AWPU_FAST_PAIRS=1 AWPU_FAST_PAIRCOLS=1" "X=1"; do
    out=$(env $env timeout -k 10 300 python3 tests/gpu_random_check.py $seed $cases 2>&1 | grep -v amdgpu.ids | tail -1)
    echo "seed $seed [$env] $out"
    case "$out" in OK*) ;; *) fail=1;; esac
  done
done
exit $fail
