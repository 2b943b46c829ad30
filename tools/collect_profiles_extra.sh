#!/bin/bash
# rocprofv3 evidence for the kernels the default bench line does not exercise at size: the
# few-queries (NQ = 1) HBM-bound sweep and the Q = 1 latency path, the HBM-bound helpers (row l2-normalise, index
# re-layout), the text-CNN encoders, the isolated training kernels.  Run on the GPU box from the repo root:
#   tools/collect_profiles_extra.sh <tag>
# then on the build box:  python tools/summarize_extra.py <tag>   (writes profiles/<tag>_*.txt|csv)
set -u
exec </dev/null
tag=${1:-extra}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
run() {  # name, command...
  name=$1; shift
  ( cd "$root" && timeout 300 "$@" > "$out/$name.txt" 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${name}_stats" -o p -- "$@" > "$out/${name}_stats.log" 2>&1 )
}
pmc() {  # name, counters, command...
  name=$1; ctr=$2; shift 2
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/${name}_pmc_$(echo $ctr | tr ' ' '_' | cut -c1-24)" -o p -- "$@" > /dev/null 2>&1 )
}
if [ -n "${R6:-}" ]; then
  # round-6 set: the small-cell encoder (gate-split kernel against lstm_fwd_kernel<2,1,1>: times, MFMA-busy), configs[2] scoring and
  # the two-pass path, the real-data legs with and without the device-side PAD-prefix bucketing
  SSE_FWD_GS=1 run shapes_gs1 python "$root/tools/bench_shapes.py"
  SSE_FWD_GS=0 run shapes_gs0 python "$root/tools/bench_shapes.py"
  SSE_FWD_GS=1 pmc shapes_gs1 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python "$root/tools/bench_shapes.py"
  SSE_FWD_GS=0 pmc shapes_gs0 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python "$root/tools/bench_shapes.py"
  run c3 python "$root/tools/bench_c3.py"
  run score env SSE_BENCH_PASSES=10 python "$root/tools/bench_score.py"
  run score_16k python "$root/tools/bench_score.py" 16384 1250000 256
  run train_default python "$root/tools/bench_train_default.py"
elif [ -n "${R5:-}" ]; then
  # round-5 set: the text-CNN kernels (times, MFMA-busy, FETCH / WRITE), configs[2] scoring on the real rows, the any-shape path
  run cnn python "$root/tools/bench_cnn.py"
  pmc cnn "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python "$root/tools/bench_cnn.py"
  pmc cnn FETCH_SIZE python "$root/tools/bench_cnn.py"
  pmc cnn WRITE_SIZE python "$root/tools/bench_cnn.py"
  run c3 python "$root/tools/bench_c3.py"
  run generic python "$root/tools/bench_generic.py"
  run train_default python "$root/tools/bench_train_default.py"
elif [ -n "${R4:-}" ]; then
  # round-4 set: what changed this round -- the fp32 training kernels (times, MFMA-busy, FETCH / WRITE), the single-query
  # call, the bf16 sweep, the recipe shapes
  SSE_TRAIN_SERIAL=1 run train python "$root/tools/bench_train.py" 8192
  SSE_TRAIN_SERIAL=1 pmc train "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python "$root/tools/bench_train.py" 8192
  SSE_TRAIN_SERIAL=1 pmc train FETCH_SIZE python "$root/tools/bench_train.py" 8192
  SSE_TRAIN_SERIAL=1 pmc train WRITE_SIZE python "$root/tools/bench_train.py" 8192
  run train_default python "$root/tools/bench_train_default.py"
  run latency python "$root/tools/bench_latency.py"
  run score env SSE_BENCH_PASSES=10 python "$root/tools/bench_score.py"
  pmc score "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" env SSE_BENCH_PASSES=4 python "$root/tools/bench_score.py"
  run shapes python "$root/tools/bench_shapes.py"
elif [ -n "${QUICK:-}" ]; then
  # round-3 set (GPU-minutes were short): the kernels that changed this round, one plain + one --stats run each, and the
  # MFMA-busy pass of the training step
  run demo python "$root/tools/bench_demo_query.py" 2500000
  run mid_batch env N=20 python "$root/tools/dbg_cluster.py" 64 600 1000 1024 2048 3072
  SSE_TRAIN_SERIAL=1 run train python "$root/tools/bench_train.py" 8192
  SSE_TRAIN_SERIAL=1 pmc train "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python "$root/tools/bench_train.py" 8192
  run train_default python "$root/tools/bench_train_default.py"
  run cnn python "$root/tools/bench_cnn.py"
  run train_concurrent python "$root/tools/bench_train.py" 128 1024 8192
else
run demo python "$root/tools/bench_demo_query.py" 10000000
pmc demo FETCH_SIZE python "$root/tools/bench_demo_query.py" 10000000
run hbm python "$root/tools/bench_hbm_kernels.py"
pmc hbm FETCH_SIZE python "$root/tools/bench_hbm_kernels.py"
pmc hbm WRITE_SIZE python "$root/tools/bench_hbm_kernels.py"
run cnn python "$root/tools/bench_cnn.py"
SSE_TRAIN_SERIAL=1 run train python "$root/tools/bench_train.py" 8192
run train_default python "$root/tools/bench_train_default.py"
run query_encode python "$root/tools/bench_query_encode.py"
run x3 python "$root/tools/bench_x3.py"
pmc x3 "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python "$root/tools/bench_x3.py"
run x3_default_shape python "$root/tools/bench_x3.py" 16384 96 64 80
run train_concurrent python "$root/tools/bench_train.py" 128 1024 8192
fi
find "$out" -name "*.csv" -size +20M -delete
ls "$out"
