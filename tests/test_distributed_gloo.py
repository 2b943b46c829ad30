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


class _CpuShardHandle(object):
    """Stands in for the HIP handle on CPU tensors (raw pointers in, oracle arithmetic): lets the product's
    ShardedIndex.score_topk -- packed single all-gather, query blocks, strided merge -- run under gloo."""

    def __init__(self, S):
        self.S = S

    @staticmethod
    def _view(ptr, shape, dtype):
        import ctypes
        n = int(np.prod(shape))
        buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def index_set_dev(self, ptr, n, S, id_base=0, stream=0):
        self.rows, self.base = self._view(ptr, (n, S), np.float32).astype(np.float64), id_base

    def score_topk_dev(self, q_ptr, Q, k, s_ptr, i_ptr, stream=0):
        from oracle import sse_oracle as O
        sc, ids = O.topk(O.scores_f64(self._view(q_ptr, (Q, self.S), np.float32), self.rows), k)
        self._view(s_ptr, (Q, k), np.float64)[:] = sc
        self._view(i_ptr, (Q, k), np.int64)[:] = ids + self.base

    def merge_topk_strided_dev(self, s_ptr, i_ptr, stride, P, Q, k, os_ptr, oi_ptr, stream=0):
        s = self._view(s_ptr, (P * stride,), np.float64)
        i = self._view(i_ptr - 0, (P * stride,), np.int64)
        ms = np.stack([s[p * stride:p * stride + Q * k].reshape(Q, k) for p in range(P)], 1).reshape(Q, -1)
        mi = np.stack([i[p * stride:p * stride + Q * k].reshape(Q, k) for p in range(P)], 1).reshape(Q, -1)
        order = np.lexsort((mi, -ms), axis=1)[:, :k]
        self._view(os_ptr, (Q, k), np.float64)[:] = np.take_along_axis(ms, order, 1)
        self._view(oi_ptr, (Q, k), np.int64)[:] = np.take_along_axis(mi, order, 1)


def _sharded_worker(rank, world, port, q_np, t_np, k, out_dir, block=16):
    sys.path.insert(0, ROOT)
    os.environ["SSE_NO_TORCH"] = "1"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sse_amd
    sh = sse_amd.ShardedIndex(_CpuShardHandle(t_np.shape[1]), rank, world, t_np.shape[0])
    rows = torch.from_numpy(t_np[sh.start:sh.end].copy())
    sh.set_local_rows(rows)
    s, i = sh.score_topk(torch.from_numpy(q_np.copy()), k, block=block)   # 37 queries -> 3 blocks, gathers overlapped
    np.savez(os.path.join(out_dir, "sh%d.npz" % rank), s=s.numpy(), i=i.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_index_class_blocks_and_packed_gather(tmp_path):
    from oracle import sse_oracle as O
    rng = np.random.RandomState(1)
    Q, N, S, k, world = 37, 403, 16, 10, 2
    q = rng.standard_normal((Q, S)).astype(np.float32)
    t = rng.standard_normal((N, S)).astype(np.float32)
    t[300] = t[5]
    port = _free_port()
    mp.spawn(_sharded_worker, args=(world, port, q, t, k, str(tmp_path)), nprocs=world, join=True)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), k)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "sh%d.npz" % r))
        assert np.array_equal(z["i"], wids) and np.abs(z["s"] - wsc).max() < 1e-12   # (BLAS blocks a row slice differently)


def test_eight_rank_sharded_index_uneven_blocks_and_shards(tmp_path):
    """The node-level shape of BASELINE configs[3]: 8 ranks, N not divisible by 8 (shards of 51 and 50 rows), query
    blocks that do not divide Q (37 = 5 x 7 + 2), ties across shard boundaries -- exactly the unsharded ranking."""
    from oracle import sse_oracle as O
    rng = np.random.RandomState(2)
    Q, N, S, k, world = 37, 403, 16, 10, 8
    q = rng.standard_normal((Q, S)).astype(np.float32)
    t = rng.standard_normal((N, S)).astype(np.float32)
    t[300] = t[5]
    t[51] = t[50]                                                  # a tie across the boundary of shards 0 | 1
    q[3] = t[50]
    port = _free_port()
    mp.spawn(_sharded_worker, args=(world, port, q, t, k, str(tmp_path), 7), nprocs=world, join=True)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), k)
    assert wids[3, :2].tolist() == [50, 51]
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "sh%d.npz" % r))
        assert np.array_equal(z["i"], wids) and np.abs(z["s"] - wsc).max() < 1e-12


# --------------------------------------------------------------------------
# data-parallel train step (sequence-semantic-embedding_amd/data_parallel.py): the product's exchange logic with
# the numpy oracle as the gradient engine (on the GPU the engine is the HIP handle:
# tests/test_gpu_train.py::test_data_parallel_two_logical_ranks)
# --------------------------------------------------------------------------

class OracleEngine(object):
    """train_grad_count / train_bind_arena / train_grads / train_apply on top of oracle/sse_oracle.py."""

    def __init__(self, params, cfg, lr):
        from oracle import sse_oracle as O
        self.O, self.p, self.cfg, self.lr = O, params, cfg, np.float32(lr)
        self.acc = O.new_optimizer_state(params)
        self.names = sorted(params)

    def train_grad_count(self):
        return sum(self.p[n].size for n in self.names) + 4

    def embedding_slice(self):
        off = sum(self.p[n].size for n in self.names[:self.names.index("word_embedding")])
        return (off,) + tuple(self.p["word_embedding"].shape)

    def train_bind_arena(self, tensor):
        self.arena = tensor.numpy()                                # shares memory with the torch tensor

    # the (row id, gradient row) exchange of data_parallel.py, numpy restatement of csrc/train.hip's emb_grad_pack /
    # emb_grad_unpack kernels (same packed layout: [count, 0, 0, 0 | ids (cap padded to x4) | rows cap x E])
    @property
    def max_seq_length(self):
        return int(self.cfg["max_seq_length"])

    def dp_packed_floats(self, cap):
        return 4 + ((cap + 3) & ~3) + cap * self.p["word_embedding"].shape[1]

    def dp_pack_embedding(self, cap, packed):
        off, V, E = self.embedding_slice()
        emb = self.arena[off:off + V * E].reshape(V, E)
        ids = np.nonzero((emb != 0).any(axis=1))[0][::-1]           # (any slot order: the device's is not deterministic either)
        assert len(ids) <= cap, "packed exchange buffer too small"
        buf = packed.numpy()
        cap4 = (cap + 3) & ~3
        buf[:4].view(np.int32)[:] = (len(ids), 0, 0, 0)
        buf[4:4 + len(ids)].view(np.int32)[:] = ids
        buf[4 + cap4:4 + cap4 + len(ids) * E] = emb[ids].ravel()

    def dp_unpack_embedding(self, gathered, world, cap):
        off, V, E = self.embedding_slice()
        emb = self.arena[off:off + V * E].reshape(V, E)
        emb[:] = 0
        n, cap4 = self.dp_packed_floats(cap), (cap + 3) & ~3
        for r in range(world):                                      # rank order: the same sums on every rank
            buf = gathered.numpy()[r * n:(r + 1) * n]
            cnt = int(buf[:1].view(np.int32)[0])
            ids = buf[4:4 + cnt].view(np.int32)
            emb[ids] += buf[4 + cap4:4 + cap4 + cnt * E].reshape(cnt, E)

    def train_grads(self, src, tgt, labels, rows_global):
        O = self.O
        loss, acc, grads = O.gradients(self.p, self.cfg, src, tgt, labels)
        w = np.float32(len(labels)) / np.float32(rows_global)      # oracle gradients are means over the local rows
        off, slices_sq = 0, 0.0
        for n in self.names:
            g = grads[n]
            if isinstance(g, tuple):
                slices_sq += float(np.sum(np.square(g[1] * w, dtype=np.float64)))
                g = O.dense_embedding_grad(g, self.p[n].shape[0])
            self.arena[off:off + g.size] = (g * w).ravel()
            off += g.size
        self.arena[off:off + 4] = (slices_sq, loss * w, acc * w, len(labels))

    def train_apply(self):
        O = self.O
        tail = self.arena[-4:]
        tot, off, dense = float(tail[0]), 0, {}
        for n in self.names:
            g = self.arena[off:off + self.p[n].size].reshape(self.p[n].shape).copy()
            off += g.size
            dense[n] = g
            if n != "word_embedding":
                tot += float(np.sum(np.square(g, dtype=np.float64)))
        gn = np.float32(np.sqrt(tot))
        scale = O.MAX_GRAD_NORM * min(np.float32(1.0) / gn, np.float32(1.0) / O.MAX_GRAD_NORM) if gn > 0 else np.float32(1)
        for n in self.names:
            g = (dense[n] * np.float32(scale)).astype(np.float32)
            self.acc[n] += g * g                                    # rows with zero gradient: acc, w unchanged
            self.p[n] -= self.lr * g / np.sqrt(self.acc[n])
        return float(tail[1]), float(tail[2])


def _dp_cfg():
    from util import model_params
    return model_params("dual-encoder", 60, 8, 12, 16, 16, 6)


def _dp_batch(seed, rows):
    rng = np.random.RandomState(seed)
    src = rng.randint(2, 60, size=(rows, 6)).astype(np.int32)
    tgt = rng.randint(2, 60, size=(rows, 6)).astype(np.int32)
    return src, tgt, (np.arange(rows) % 2 == 0).astype(np.float32)


def _dp_worker(rank, world, port, out_dir, uneven, sparse=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["SSE_NO_TORCH"] = "1"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sse_amd
    from oracle import sse_oracle as O
    cfg = _dp_cfg()
    eng = OracleEngine(O.init_params(cfg, seed=3), cfg, 0.9)
    tr = sse_amd.DataParallelTrainer(eng, sparse_embedding=sparse)
    hist = []
    for step in range(3):
        src, tgt, z = _dp_batch(step, 22)
        if uneven:                                                  # rank 0: 15 rows, rank 1: 7
            sl = slice(0, 15) if rank == 0 else slice(15, 22)
            hist.append(tr.train_step(src[sl], tgt[sl], z[sl]))
        else:
            hist.append(tr.train_step(*sse_amd.split_batch(src, tgt, z, rank, world), rows_global=22))
    assert tr.last_exchange == ("sparse" if sparse else "dense")
    np.savez(os.path.join(out_dir, "dp%d.npz" % rank), hist=np.array(hist), **eng.p)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("uneven,sparse", [(False, False), (True, False), (True, True)])
def test_two_rank_data_parallel_step_equals_single_process(tmp_path, uneven, sparse):
    """2 gloo ranks x half a batch == the oracle's single-process step on the whole batch
    (loss, train_acc and every parameter after 3 steps), and both ranks end bit-identical.  sparse: the embedding
    gradient travels as (row id, gradient row) pairs instead of the dense [V,E] block."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import sse_oracle as O
    world, port = 2, _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path), uneven, sparse), nprocs=world, join=True)
    cfg = _dp_cfg()
    p = O.init_params(cfg, seed=3)
    acc = O.new_optimizer_state(p)
    want = [O.train_step(p, acc, cfg, *_dp_batch(step, 22), 0.9) for step in range(3)]
    z0, z1 = (np.load(os.path.join(str(tmp_path), "dp%d.npz" % r)) for r in range(2))
    assert np.allclose(z0["hist"], np.array(want), rtol=1e-5, atol=1e-6)
    for n in p:
        assert np.array_equal(z0[n], z1[n]), n                     # ranks stay in lock-step, bit for bit
        assert np.abs(z0[n] - p[n]).max() < 2e-5, n


def _dp_threshold_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["SSE_NO_TORCH"] = "1"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sse_amd
    from oracle import sse_oracle as O
    from util import model_params
    cfg = model_params("dual-encoder", 2000, 8, 12, 16, 16, 6)
    eng = OracleEngine(O.init_params(cfg, seed=3), cfg, 0.9)
    tr = sse_amd.DataParallelTrainer(eng)                           # sparse_embedding=None: automatic
    kinds = []
    for rows in (41, 42):                                           # 2 * 41 * 6 = 492 < 2000 // 4 <= 2 * 42 * 6
        rng = np.random.RandomState(rows)
        src = rng.randint(2, 2000, size=(rows, 6)).astype(np.int32)
        tgt = rng.randint(2, 2000, size=(rows, 6)).astype(np.int32)
        z = (np.arange(rows) % 2 == 0).astype(np.float32)
        tr.train_step(*sse_amd.split_batch(src, tgt, z, rank, world), rows_global=rows)
        kinds.append(tr.last_exchange)
    np.savez(os.path.join(out_dir, "thr%d.npz" % rank), kinds=np.array(kinds), **eng.p)
    dist.barrier()
    dist.destroy_process_group()


def test_sparse_or_dense_is_decided_identically_on_every_rank(tmp_path):
    """the automatic sparse / dense choice used the LOCAL row count; with 41 global rows split 21 / 20
    over two ranks at V = 2000, T = 6 one rank chose the dense all-reduce and the other the sparse exchange -- mismatched
    collectives.  Now decided from (rows_global, T, V): both ranks go sparse at 41 rows, dense at 42, stay bit-identical,
    and match the single-process steps."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import sse_oracle as O
    from util import model_params
    world, port = 2, _free_port()
    mp.spawn(_dp_threshold_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z0, z1 = (np.load(os.path.join(str(tmp_path), "thr%d.npz" % r)) for r in range(2))
    assert z0["kinds"].tolist() == z1["kinds"].tolist() == ["sparse", "dense"]
    cfg = model_params("dual-encoder", 2000, 8, 12, 16, 16, 6)
    p = O.init_params(cfg, seed=3)
    acc = O.new_optimizer_state(p)
    for rows in (41, 42):
        rng = np.random.RandomState(rows)
        src = rng.randint(2, 2000, size=(rows, 6)).astype(np.int32)
        tgt = rng.randint(2, 2000, size=(rows, 6)).astype(np.int32)
        O.train_step(p, acc, cfg, src, tgt, (np.arange(rows) % 2 == 0).astype(np.float32), 0.9)
    for n in p:
        assert np.array_equal(z0[n], z1[n]), n
        assert np.abs(z0[n] - p[n]).max() < 2e-5, n
