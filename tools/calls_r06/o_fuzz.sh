#!/bin/bash
# round 6 call O: randomised sweeps of the new large-batch paths (three seeds) + the classic fuzz at two more seeds
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
o=gpurun_out/r06o; mkdir -p $o
for seed in 0 1 2; do timeout 600 python tools/fuzz_large_batches.py 14 $seed 2>&1 | grep -v amdgpu | tee -a $o/fuzz_large.txt | grep -v " ok  " | cut -c1-200; done
for seed in 11 12; do timeout 600 python tools/fuzz_parity.py 40 $seed 2>&1 | grep -v amdgpu | tail -2 | tee -a $o/fuzz.txt | cut -c1-200; done
timeout 300 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
