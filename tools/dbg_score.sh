#!/bin/bash
# Phase clocks of the bf16 sweep with parts of the top-k epilogue switched off (measurement build: -DSSE_SCORE_CLOCK
# -DSSE_SCORE_MEASURE, env SSE_SCORE_DBG bits: 1 no epilogue, 2 no hit body, 4 hit tests without parking, 8 no publish).
# Results are invalid with any bit set; only the cycle counts are read.  usage: bash tools/dbg_score.sh <tag> <lib.so> bits...
export TMPDIR=/tmp
tag=$1; lib=$2; shift; shift
o=gpurun_out/$tag; mkdir -p $o
for bits in "$@"; do
  SSE_SCORE_DBG=$bits SSE_BENCH_PASSES=12 SSE_HIP_LIB=$(pwd)/sequence-semantic-embedding_amd/$lib timeout 200 python tools/bench_score.py > $o/run.txt 2>&1
  echo "== SSE_SCORE_DBG=$bits" | tee -a $o/dbg.txt
  grep "score clock <4,1>" $o/run.txt | grep "wave 0\|wave 4" | tail -2 | sed 's/\[score clock <4,1> KG=16\] //' | cut -c1-220 | tee -a $o/dbg.txt
done
