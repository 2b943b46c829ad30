"""N>1 path on CPU: world_size-2 `gloo` processes exercise the product's shard
arithmetic and the all-gather of per-shard top-k (sequence-semantic-embedding_amd/
sharded.py); the per-shard top-k and the k-way merge are done by the oracle here
(on the GPU they are sse_score_topk_dev / sse_merge_topk_dev, covered by
tests/test_gpu_score.py::test_sharded_index_equals_unsharded)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q_np, t_np, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["SSE_NO_TORCH"] = "1"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sse_amd
    from oracle import sse_oracle as O
    start, end = sse_amd.shard_bounds(t_np.shape[0], world)[rank]
    assert (start, end) == sse_amd.split_rows(t_np.shape[0], rank, world)
    sc, ids = O.topk(O.scores_f64(q_np, t_np[start:end].astype(np.float64)), k)
    ids = ids + start                                             # global row ids (id_base)
    gs, gi = sse_amd.all_gather_topk(torch.from_numpy(sc.copy()), torch.from_numpy(ids.copy()))
    assert gs.shape == (world, q_np.shape[0], k)
    # k-way merge, order rule (score desc, id asc)
    ms = gs.permute(1, 0, 2).reshape(q_np.shape[0], -1).numpy()
    mi = gi.permute(1, 0, 2).reshape(q_np.shape[0], -1).numpy()
    order = np.lexsort((mi, -ms), axis=1)[:, :k]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), s=np.take_along_axis(ms, order, 1), i=np.take_along_axis(mi, order, 1))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_and_balance():
    import sse_amd
    for n, w in [(10, 3), (8, 8), (5, 8), (10_000_000, 8), (1, 1)]:
        b = sse_amd.shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [e - s for s, e in b]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharded_topk_equals_unsharded(tmp_path):
    from oracle import sse_oracle as O
    rng = np.random.RandomState(0)
    Q, N, S, k, world = 37, 501, 16, 10, 2
    q = rng.standard_normal((Q, S)).astype(np.float32)
    t = rng.standard_normal((N, S)).astype(np.float32)
    t[400] = t[7]                                                  # an exact tie across the two shards
    port = _free_port()
    mp.spawn(_worker, args=(world, port, q, t, k, str(tmp_path)), nprocs=world, join=True)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), k)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(z["i"], wids)
        assert np.array_equal(z["s"], wsc)
