#!/bin/bash
# round 6 call I: which earlier test file makes the qna test run at 52 instead of 26 ms/step inside the whole suite?
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for f in test_gpu_rccl test_gpu_score test_gpu_encode test_gpu_fullsize test_gpu_cnn; do
  echo "== after $f"
  SSE_QNA_EPOCHS=20 timeout 900 python -m pytest tests/$f.py tests/test_gpu_trained_parity.py -x -q -k "not standin and not crosslingual" 2>&1 | grep "\[qna\] makefile\|passed\|failed" | cut -c1-200
done
