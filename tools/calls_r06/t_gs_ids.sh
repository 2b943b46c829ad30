#!/bin/bash
# round 6 call T: gate-split kernel with token ids two steps ahead: bit-identity tests, shapes, clock table
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_fuzz.py -x -q -k "gate_split or large_batches or small_cells or out_of_range" 2>&1 | tail -3
timeout 300 python tools/bench_shapes.py 2>&1 | grep "E="
SSE_HIP_LIB=$PWD/sequence-semantic-embedding_amd/libsse_gsclk.so timeout 300 python tools/bench_shapes.py 50 96 64 80 2>&1 | grep "gs clock" | cut -c1-330 | sed -n 1p\;5p
