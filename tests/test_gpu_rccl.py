"""The real RCCL collectives of the N>1 paths, on the one GPU a test box has: a 1-rank `nccl` process group
(127.0.0.1 rendezvous) and the product's DataParallelTrainer / ShardedIndex with the collectives forced on.
Checks what the CPU `gloo` tests cannot: RCCL on buffers the HIP library writes through raw pointers (stream
ordering between the library's streams and torch's), float64 / int64 all-gather, float32 all-reduce in place."""
import os
import socket

import numpy as np
import pytest

from oracle import sse_oracle as O
from tests.util import make_pair, model_params, random_ids

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_data_parallel_step_through_rccl_all_reduce(nccl_group):
    import sse_amd
    params = model_params("dual-encoder", 200, 50, 128, 128, 64, 12, lr=0.9)
    m, p = make_pair(params, seed=4)
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(1)
    src = np.repeat(random_ids(rng, 24, 12, 200, 0.5), 2, axis=0)
    tgt = random_ids(rng, 48, 12, 200, 0.5)
    z = np.tile(np.array([1.0, 0.0], np.float32), 24)
    tr = sse_amd.DataParallelTrainer(m.handle, device="cuda:0", always_reduce=True)
    for _ in range(3):
        want = O.train_step(p, st, params, src, tgt, z, 0.9)
        got = tr.train_step(src, tgt, z)
        assert got[0] == pytest.approx(float(want[0]), rel=1e-4)
    got_vars = m.get_variables()
    for name, w in p.items():
        assert np.abs(got_vars[name].reshape(w.shape) - w).max() < 1e-3, name
    assert tr.global_rows(48) == 48


def test_sharded_scoring_through_rccl_all_gather(nccl_group):
    import torch
    import sse_amd
    params = model_params("dual-encoder", 50, 8, 16, 16, 32, 4)
    m, _ = make_pair(params)
    rng = np.random.RandomState(3)
    t = rng.standard_normal((5000, 32)).astype(np.float32)
    q = rng.standard_normal((300, 32)).astype(np.float32)
    sh = sse_amd.ShardedIndex(m.handle, 0, 1, 5000, always_gather=True)
    sh.set_local_rows(torch.from_numpy(t).cuda())
    s, i = sh.score_topk(torch.from_numpy(q).cuda(), 10)
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), 10)
    assert np.array_equal(i.cpu().numpy(), wids)
    assert np.abs(s.cpu().numpy() - wsc).max() < 1e-12


def test_data_parallel_on_a_side_stream_and_sparse_embedding_exchange(nccl_group):
    """sse_set_stream: the trainer hands torch's CURRENT stream to the library, so a step issued inside
    `with torch.cuda.stream(s)` orders its kernels, the RCCL collective and the update on that stream (round 2
    raised unless the default stream was current).  sparse_embedding: the word-embedding gradient travels as
    (row id, gradient row) pairs -- same update as the dense all-reduce."""
    import torch
    import sse_amd
    params = model_params("dual-encoder", 5000, 50, 128, 128, 64, 12, lr=0.9)
    (ma, p), (mb, _) = make_pair(params, seed=5), make_pair(params, seed=5)
    rng = np.random.RandomState(2)
    src = np.repeat(random_ids(rng, 32, 12, 5000, 0.5), 2, axis=0)
    tgt = random_ids(rng, 64, 12, 5000, 0.5)
    z = np.tile(np.array([1.0, 0.0], np.float32), 32)
    dense = sse_amd.DataParallelTrainer(ma.handle, device="cuda:0", always_reduce=True, sparse_embedding=False)
    sparse = sse_amd.DataParallelTrainer(mb.handle, device="cuda:0", always_reduce=True)      # automatic: 2*64*12 < 5000/4? no -> forced below
    side = torch.cuda.Stream()
    for step in range(3):
        a = dense.train_step(src, tgt, z)
        with torch.cuda.stream(side):
            sparse.sparse_embedding = True
            b = sparse.train_step(src, tgt, z)
        assert dense.last_exchange == "dense" and sparse.last_exchange == "sparse"
        assert b == pytest.approx(a, rel=1e-6, abs=1e-7)
    side.synchronize()
    va, vb = ma.get_variables(with_slots=True), mb.get_variables(with_slots=True)
    for k in va:
        assert np.abs(va[k] - vb[k]).max() < 2e-6, k
    # automatic choice: sparse only when a step touches fewer than V/4 rows
    assert sparse._use_sparse(64, src, False) is True or 2 * 64 * 12 >= 5000 // 4
    auto = sse_amd.DataParallelTrainer(ma.handle, device="cuda:0", always_reduce=True)
    assert auto._use_sparse(8, src[:8], False) == (2 * 8 * 12 < 5000 // 4)
    ma.handle.set_stream(0)
    mb.handle.set_stream(0)


def test_torch_free_rccl_exchange_through_the_c_abi():
    """the exchange step of the sharded index WITHOUT torch.distributed -- an RCCL communicator created
    through the C ABI (sse_rccl_get_unique_id / sse_rccl_comm_init_rank), then sse_score_topk_sharded_dev (shard sweep ->
    ncclAllGather of the packed lists -> merge on one stream) and sse_allgather_merge_topk_dev on ready-made lists.  One rank
    (one GPU here); the multi-rank merge semantics are the gloo tests' (same merge kernel, same packing).  torch only holds
    the device buffers."""
    import torch
    import sse_amd
    params = model_params("dual-encoder", 50, 8, 16, 16, 32, 4)
    m, _ = make_pair(params)
    h = m.handle
    rng = np.random.RandomState(13)
    t = rng.standard_normal((7000, 32)).astype(np.float32)
    q = rng.standard_normal((260, 32)).astype(np.float32)
    uid = h.rccl_unique_id()
    assert len(uid) == 128
    sh = sse_amd.RcclShardedIndex(h, 0, 1, 7000, uid)
    td, qd = torch.from_numpy(t).cuda(), torch.from_numpy(q).cuda()
    out_s = torch.empty((260, 10), dtype=torch.float64, device="cuda")
    out_i = torch.empty((260, 10), dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sh.set_local_rows_ptr(td.data_ptr(), 7000, 32, stream=st.cuda_stream)
        sh.score_topk_ptr(qd.data_ptr(), 260, 10, out_s.data_ptr(), out_i.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), 10)
    assert np.array_equal(out_i.cpu().numpy(), wids)
    assert np.abs(out_s.cpu().numpy() - wsc).max() < 1e-12
    # ready-made lists (e.g. of a shard scored earlier): gather + merge only; with one rank the merge of one list is the list
    ls, li = torch.from_numpy(wsc).cuda(), torch.from_numpy(wids).cuda()
    o2s, o2i = torch.empty_like(ls), torch.empty_like(li)
    h.allgather_merge_topk_dev(sh.comm, 1, ls.data_ptr(), li.data_ptr(), 260, 10, o2s.data_ptr(), o2i.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(o2i, li) and torch.equal(o2s, ls)
    # an id_base: global row ids come back
    sh2 = sse_amd.RcclShardedIndex(h, 0, 1, 7000, h.rccl_unique_id())
    h.index_set_dev(td.data_ptr(), 7000, 32, id_base=123456)
    sh2.score_topk_ptr(qd.data_ptr(), 260, 3, out_s.data_ptr(), out_i.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out_i.reshape(-1)[:780].cpu().numpy().reshape(260, 3), wids[:, :3] + 123456)   # ([Q][3] packed at the front of the buffer)
    sh.close()
    sh2.close()
    with pytest.raises(sse_amd.SSEError):
        h.allgather_merge_topk_dev(0, 1, ls.data_ptr(), li.data_ptr(), 260, 10, o2s.data_ptr(), o2i.data_ptr())


@pytest.mark.parametrize("Q,k,P", [(260, 10, 2), (37, 3, 3), (1000, 16, 8)])
def test_merge_of_the_layout_a_multi_rank_gather_produces(Q, k, P):
    """the world > 1 semantics of sse_allgather_merge_topk_dev / sse_score_topk_sharded_dev on the ONE GPU of
    a test box.  What ncclAllGather delivers on every rank is rank-major: world x [this rank's float64 score bits [Q][k] |
    int64 ids [Q][k]]  (sse_api.hip, allgather_merge_locked).  Here P handles on the one device play the ranks: each holds an
    uneven row shard with its own id_base != 0 (the last one a short tail), scores its shard into its slot of that buffer --
    laid out by hand exactly as the gather would -- and the strided merge (shard_stride = 2*Q*k words) must return the
    unsharded result: global row ids, float64 scores, the tie rule (lower row id first) across shard boundaries."""
    import torch
    import sse_amd
    from sse_amd.sharded import shard_bounds
    S, N = 32, 9001
    rng = np.random.RandomState(Q + k)
    t = rng.standard_normal((N, S)).astype(np.float32)
    t[11] = rng.choice([-4.0, 0.0, 4.0], size=S)                           # (small integers: the oracle's float64 dots of the duplicates are
                                                                           # exact in any summation order, so its tie is an exact tie too)
    t[N // 2 + 3] = t[11]                                                  # an exact tie across a shard boundary (P = 2: rows 11 and 4503)
    t[N - 1] = t[11]                                                       # ... and in the last rank's tail
    q = rng.standard_normal((Q, S)).astype(np.float32)
    q[0] = t[11]                                                           # query 0's best three are the tied rows
    params = model_params("dual-encoder", 50, 8, 16, 16, S, 4)
    handles = [make_pair(params)[0].handle for _ in range(P)]
    bounds = shard_bounds(N, P)
    assert bounds[-1][1] == N and all(b[0] > 0 for b in bounds[1:])
    qd = torch.from_numpy(q).cuda()
    n = Q * k
    gathered = torch.empty((P, 2, Q, k), dtype=torch.int64, device="cuda")  # the all-gather's receive buffer, rank-major
    for r, (h, (a, b)) in enumerate(zip(handles, bounds)):
        rows = torch.from_numpy(t[a:b]).cuda()
        h.index_set_dev(rows.data_ptr(), b - a, S, id_base=a)
        h.score_topk_dev(qd.data_ptr(), Q, k, gathered[r, 0].data_ptr(), gathered[r, 1].data_ptr())
        h.synchronize()
    out_s = torch.empty((Q, k), dtype=torch.float64, device="cuda")
    out_i = torch.empty((Q, k), dtype=torch.int64, device="cuda")
    base = gathered.data_ptr()
    handles[P - 1].merge_topk_strided_dev(base, base + n * 8, 2 * n, P, Q, k, out_s.data_ptr(), out_i.data_ptr())   # any rank merges alike
    torch.cuda.synchronize()
    wsc, wids = O.topk(O.scores_f64(q, t.astype(np.float64)), k)
    assert np.array_equal(out_i.cpu().numpy(), wids)
    assert np.abs(out_s.cpu().numpy() - wsc).max() < 1e-12
    tied = [11, N // 2 + 3, N - 1][:k]
    assert list(out_i[0, :len(tied)].cpu().numpy()) == tied
    # every rank's slot holds GLOBAL ids of its own range only
    for r, (a, b) in enumerate(bounds):
        ids_r = gathered[r, 1].cpu().numpy()
        assert ids_r.min() >= a and ids_r.max() < b
    for h in handles:
        h.close()


def test_rccl_binding_is_one_library_instance():
    """every RCCL entry point is bound from ONE library instance -- $SSE_RCCL_LIB, else an RCCL already
    mapped into the process (torch's, even when loaded RTLD_LOCAL), else librccl.so.1 -- and the library says which."""
    import torch  # noqa: F401  (maps torch's RCCL into the process before the first use below)
    params = model_params("dual-encoder", 50, 8, 16, 16, 32, 4)
    m, _ = make_pair(params)
    path = m.handle.rccl_library_path()
    assert path and "rccl" in os.path.basename(path)
    mapped = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
    assert mapped, "no RCCL mapped after use?"
    if os.path.isabs(path):
        assert path in mapped
    assert len({os.path.realpath(x) for x in mapped}) == 1, "two RCCL instances in one process: %s" % sorted(set(mapped))
    m.handle.rccl_group_start()
    m.handle.rccl_group_end()
