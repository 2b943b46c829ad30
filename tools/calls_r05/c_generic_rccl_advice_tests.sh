#!/bin/bash
# round 5, call: any-shape LSTM path, torch-free RCCL exchange, ADVICE regression tests
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r05c; mkdir -p $o
timeout 800 python -m pytest tests/test_gpu_generic.py tests/test_gpu_rccl.py tests/test_gpu_score.py::test_alternating_shapes_share_the_pinned_mirror_block tests/test_gpu_train.py::test_tape_size_limit_is_an_explicit_error tests/test_gpu_score.py::test_small_index_path_equals_the_list_sweep -m gpu -q --timeout 300 -p no:cacheprovider > $o/tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed\|^FAILED\|^E  " $o/tests.log | head -40
