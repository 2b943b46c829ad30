#!/bin/bash
# who segfaults at exit under rocprofv3?  bisect: library load only / one handle closed explicitly / one handle left to the exit hook
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
cd /tmp
run() { echo "$1: SIGSEGV lines: $(timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p$RANDOM -o p -- python -c "$2" 2>&1 | grep -c SIGSEGV)"; }
P="import sys; sys.path.insert(0,'$R'); import numpy as np; import sse_amd"
run "load only" "$P; sse_amd.load_library()"
M="m=sse_amd.SSEModel(dict(forward_only=True, network_mode='dual-encoder', predict_nbest=10, max_seq_length=12, vocab_size=200, embedding_size=50, encoding_size=64, src_cell_size=96, tgt_cell_size=96, learning_rate=0.9, learning_rate_decay_factor=0.99, targetSpaceSize=40)); m.init_variables(seed=0)"
run "handle, no kernels, closed" "$P; $M; m.handle.close()"
run "handle, no kernels, left open" "$P; $M"
run "encode 9 rows (cluster kernel, cooperative), closed" "$P; $M; m.encode_source(np.ones((9,12),np.int32)); m.handle.close()"
run "encode 9 rows, plain launch, closed" "$P; $M; m.handle.set_option('lstm_cluster_coop',0); m.encode_source(np.ones((9,12),np.int32)); m.handle.close()"
run "encode 1100 rows (matrix kernel), closed" "$P; $M; m.encode_source(np.ones((1100,12),np.int32)); m.handle.close()"
run "encode 300 rows small kernel only, closed" "$P; $M; m.handle.set_option('lstm_persist_rows',0); m.handle.set_option('lstm_cluster_rows',0); m.encode_source(np.ones((300,12),np.int32)); m.handle.close()"
