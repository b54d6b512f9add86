#!/bin/bash
# broader randomised parity sweep than the test suite runs: every forced kernel shape of tests/test_gpu_random.py
# with other seeds and more cases (run under gpurun; prints one line per shape and seed)
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
fail=0
run() {  # run <label> <seed> <cases> ENV=VAL...
  local label=$1 seed=$2 cases=$3; shift 3
  if out=$(env "$@" timeout -k 10 300 python tests/gpu_random_check.py "$seed" "$cases" 2>&1); then
    echo "ok   $label seed $seed: $(echo "$out" | tail -1)"
  else
    echo "FAIL $label seed $seed: $(echo "$out" | tail -3)"; fail=1
  fi
}
for seed in ${SEEDS:-11 12 13}; do
  run pairs $seed 20 AWPU_SHAPE=pair
  run pairs_grid $seed 20 AWPU_SHAPE=pair_vertical AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1
  run quads_random $seed 20 AWPU_SHAPE=quad AWPU_TEST_GRID=1
  run quads_coincide $seed 20 AWPU_SHAPE=quad AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1
  run stationary $seed 20 AWPU_SHAPE=stationary
  run stationary_grid $seed 20 AWPU_SHAPE=stationary AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1
  run quadh_every_call $seed 20 AWPU_SHAPE=quadh AWPU_TEST_GRID=1
  run single_db $seed 12 AWPU_SHAPE=single_db
  run single_small $seed 12 AWPU_SHAPE=single_small
  run fir8_planes $seed 20 AWPU_TEST_INTERP=fir8 AWPU_SHAPE=fir8_planes
  run fir8_planes_grid $seed 20 AWPU_TEST_INTERP=fir8 AWPU_SHAPE=fir8_planes AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1
  run fir8 $seed 12 AWPU_TEST_INTERP=fir8
  run default_grid $seed 20 AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1
  run default $seed 20 X=1
  run device $seed 12 AWPU_TEST_PATH=device
  run reuse $seed 10 AWPU_TEST_REUSE=1
  run exact $seed 20 AWPU_TEST_MATH=exact
  run exact_grid $seed 20 AWPU_TEST_MATH=exact AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1
  run exact_verify $seed 8 AWPU_TEST_MATH=exact AWPU_SHAPE=exact_verify
  run exact_nd2_random $seed 20 AWPU_TEST_MATH=exact AWPU_TEST_GRID=1 AWPU_SHAPE=exact_nd2
  run exact_nd2_coincide $seed 20 AWPU_TEST_MATH=exact AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1 AWPU_SHAPE=exact_nd2
  run exact_nd1_coincide $seed 20 AWPU_TEST_MATH=exact AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1 AWPU_SHAPE=exact_nd1
  run exact_quad_r4 $seed 12 AWPU_TEST_MATH=exact AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1 AWPU_SHAPE=exact_quad
  run exact_ndp_random $seed 20 AWPU_TEST_MATH=exact AWPU_TEST_GRID=1 AWPU_TEST_BATCH=1 AWPU_SHAPE=exact_ndp
  run exact_ndp_coincide $seed 20 AWPU_TEST_MATH=exact AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1 AWPU_TEST_BATCH=1 AWPU_SHAPE=exact_ndp
  run exact_single_default $seed 20 AWPU_TEST_MATH=exact AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1 AWPU_TEST_BATCH=1
  run fast_single_default $seed 20 AWPU_TEST_GRID=1 AWPU_TEST_COINCIDE=1 AWPU_TEST_BATCH=1
done
exit $fail
