#!/usr/bin/env python
"""Two RCCL ranks as two PROCESSES on the one GPU of a test box, torch-free exchange through the C ABI ("add a
2-rank test, even if both ranks share one GPU across two processes").  Rank 0 writes the ncclUniqueId to a file; each rank
holds half of the index (id_base = its offset), runs sse_score_topk_sharded_dev with world = 2 and checks the merged result
against the float64 oracle.  RCCL builds that refuse two ranks on one device ("Duplicate GPU detected") make
ncclCommInitRank fail: reported as `refused`, exit code 3.

    python tools/rccl_two_ranks_one_gpu.py            # parent: spawns both ranks
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(rank, world, path):
    import torch
    import sse_amd
    from oracle import sse_oracle as O
    from sse_amd.sharded import shard_bounds
    params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=4, vocab_size=50, embedding_size=8,
                  encoding_size=32, src_cell_size=16, tgt_cell_size=16, learning_rate=0.9, learning_rate_decay_factor=0.99, targetSpaceSize=7)
    m = sse_amd.SSEModel(params)
    m.init_variables(seed=0)
    h = m.handle
    if rank == 0:
        uid = h.rccl_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 60:
                raise SystemExit("rank 0 never wrote the unique id")
            time.sleep(0.05)
        uid = open(path, "rb").read()
    print("rank %d: RCCL instance %s" % (rank, h.rccl_library_path()), flush=True)
    try:
        comm = h.rccl_comm_init_rank(world, rank, uid)
    except sse_amd.SSEError as e:
        print("rank %d: refused: %s" % (rank, e), flush=True)
        raise SystemExit(3)
    rng = np.random.RandomState(5)
    N, S, Q, k = 9001, 32, 300, 10
    t = rng.standard_normal((N, S)).astype(np.float32)
    q = rng.standard_normal((Q, S)).astype(np.float32)
    a, b = shard_bounds(N, world)[rank]
    rows = torch.from_numpy(t[a:b]).cuda()
    qd = torch.from_numpy(q).cuda()
    h.index_set_dev(rows.data_ptr(), b - a, S, id_base=a)
    out_s = torch.empty((Q, k), dtype=torch.float64, device="cuda")
    out_i = torch.empty((Q, k), dtype=torch.int64, device="cuda")
    for _ in range(3):
        h.score_topk_sharded_dev(comm, world, qd.data_ptr(), Q, k, out_s.data_ptr(), out_i.data_ptr())
    torch.cuda.synchronize()
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), k)
    ok = bool(np.array_equal(out_i.cpu().numpy(), wids) and np.abs(out_s.cpu().numpy() - wsc).max() < 1e-12)
    print("rank %d of %d: shard rows [%d, %d), merged top-%d of %d queries equal to the unsharded oracle: %s" % (rank, world, a, b, k, Q, ok), flush=True)
    h.rccl_comm_destroy(comm)
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    if len(sys.argv) == 4:
        rank_main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
    world = 2
    path = os.path.join(tempfile.mkdtemp(), "uid")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), path], env=env) for r in range(world)]
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=150))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    print("exit codes", rcs)
    sys.exit(max(rcs) if all(r >= 0 for r in rcs) else 4)
