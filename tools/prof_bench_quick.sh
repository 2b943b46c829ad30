cd /root/repo; export TMPDIR=/tmp; root=/root/repo; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/pb -o p -- python $root/bench.py --no-cpu-baseline --no-train-leg > $root/gpurun_out/pb.log 2>&1
tail -1 $root/gpurun_out/pb.log | cut -c1-220
grep -E "rescore|score_topk|lstm_fwd_kernel<2" $root/gpurun_out/pb/p_kernel_stats.csv | cut -c1-130
