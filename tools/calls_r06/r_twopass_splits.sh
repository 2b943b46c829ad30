#!/bin/bash
# round 6 call R: splits of the two list-free sweeps chosen by rounds (tools/bench_c3.py; SSE_TWO_PASS_SPLITS forces a count)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for f in 0 4 8 16; do
  echo "== SSE_TWO_PASS_SPLITS=$f (0 = automatic)"
  SSE_TWO_PASS_SPLITS=$f timeout 300 python tools/bench_c3.py 2>&1 | grep "two_pass_rows=262144\|two-pass" | cut -c1-200
done
timeout 600 python -m pytest tests/test_gpu_score.py -x -q -k "two_pass" 2>&1 | tail -3
