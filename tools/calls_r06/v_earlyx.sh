#!/bin/bash
# round 6 call V: gate-split kernel, x part of a step before the h wait (GS_EARLYX): A/B + bit-identity tests
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do
for lib in libsse_noearly libsse_hip; do
  echo "== $lib"; SSE_HIP_LIB=$PWD/sequence-semantic-embedding_amd/$lib.so timeout 300 python tools/bench_shapes.py 2>&1 | grep "E=" | cut -c1-120
done; done
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_fuzz.py -x -q -k "gate_split or large_batches or small_cells" 2>&1 | tail -3
