#!/bin/bash
# round 6 call H: the qna learned-model test alone (is 52 ms/step an in-suite effect?), then after the other trained-parity tests
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_gpu_trained_parity.py -x -q -k qna 2>&1 | grep "qna\|passed\|failed"
timeout 900 python -m pytest tests/test_gpu_trained_parity.py -x -q 2>&1 | grep "\[qna\] makefile\|passed\|failed"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_trained_parity.py -x -q -k "not standin and not crosslingual" 2>&1 | grep "\[qna\] makefile\|passed\|failed"
