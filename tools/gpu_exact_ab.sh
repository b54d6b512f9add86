#!/bin/bash
# reference-order kernels side by side on one box (run under gpurun): the bit-identity tests of AWPU_MATH_F32_EXACT on the default
# shape, then the headline bench per forced shape.  usage: tools/gpu_exact_ab.sh <tag> [shape ...]
set -uo pipefail
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AWPU_NO_BUILD=1
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s -k "${TESTS:-exact_mode or reference_order}" > $out/pytest_gpu.log 2>&1 \
  && echo "pytest ok: $(tail -1 $out/pytest_gpu.log)" || { echo "pytest FAILED"; tail -60 $out/pytest_gpu.log; exit 1; }
for rep in $(seq 1 ${REPS:-2}); do
for shape in "${@:-default}"; do
  if [ "$shape" = default ]; then unset AWPU_SHAPE; else export AWPU_SHAPE=$shape; fi
  timeout -k 10 200 python bench.py --math exact --cpu-seconds 0 --no-extras ${BENCH_ARGS:-} > $out/${shape}_$rep.json 2> $out/${shape}_$rep.err \
    || { echo "bench $shape FAILED"; tail -20 $out/${shape}_$rep.err; exit 1; }
  python - "$out/${shape}_$rep.json" $shape <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("%-12s value %.0f frames/s  kernel %.3f ms  valu %.3f  parity %.2e" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms"], d["valu"]["frac"], d["parity_max_rel_err"]))
PY
done
done
