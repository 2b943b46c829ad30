"""Row-sharded target index over the GPUs of one node (BASELINE
configs[3]): rank g holds rows [g*N/P, (g+1)*N/P), queries are replicated, every
rank computes its shard's top-k with GLOBAL row ids, ONE all-gather per query
block (RCCL over xGMI when the process group is `nccl`) exchanges the [Q,k]
(float64 score, int64 id) lists packed in one buffer -- 16*Q*k bytes per rank; the
exact float64 scores must travel: a rank cannot re-score candidates of rows it
does not hold -- and a k-way merge with the same order rule (score desc, row id
asc) yields exactly the unsharded result.  Queries go in blocks: the gather of
block i runs on RCCL's stream while block i+1 sweeps the shard.  Every library
call is enqueued on torch's CURRENT stream (passed explicitly), which is also
the stream torch.distributed orders its collectives against.

The reference has no distributed code at all; this is the one real exchange
step of the hot path.  torch / torch.distributed are plumbing only.
"""


def shard_bounds(n_rows, world):
    """Contiguous, balanced row ranges: [(start, end)] * world; sizes differ by <= 1."""
    base, extra = divmod(int(n_rows), int(world))
    out, start = [], 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        out.append((start, start + size))
        start += size
    return out


def all_gather_topk(local_scores, local_ids, group=None, force=False):
    """All-gather the per-shard lists.  local_* are [Q,k] tensors (CUDA with the
    nccl backend, CPU with gloo); returns ([P,Q,k] scores, [P,Q,k] ids), shard-major."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    gs = torch.empty((world,) + tuple(local_scores.shape), dtype=local_scores.dtype, device=local_scores.device)
    gi = torch.empty((world,) + tuple(local_ids.shape), dtype=local_ids.dtype, device=local_ids.device)
    if world == 1 and not force:
        gs[0].copy_(local_scores)
        gi[0].copy_(local_ids)
        return gs, gi
    if local_scores.is_cuda:
        dist.all_gather_into_tensor(gs, local_scores.contiguous(), group=group)
        dist.all_gather_into_tensor(gi, local_ids.contiguous(), group=group)
    else:                                   # gloo: list form
        dist.all_gather(list(gs.unbind(0)), local_scores.contiguous(), group=group)
        dist.all_gather(list(gi.unbind(0)), local_ids.contiguous(), group=group)
    return gs, gi


class ShardedIndex(object):
    """One rank's view of the sharded index.  `handle` is an sse_amd Handle."""

    def __init__(self, handle, rank, world, n_total, group=None, always_gather=False):
        """always_gather: run the all-gather + merge even with one shard (exercises the RCCL path on one GPU)."""
        self.handle, self.rank, self.world, self.group = handle, int(rank), int(world), group
        self.always_gather = bool(always_gather)
        self.n_total = int(n_total)
        self.start, self.end = shard_bounds(n_total, world)[rank]

    def set_local_rows(self, rows):
        """rows: CUDA float32 tensor [end-start, S] -- this rank's shard, already on its GPU."""
        if rows.shape[0] != self.end - self.start:
            raise ValueError("shard of rank %d must have %d rows, got %d" % (self.rank, self.end - self.start, rows.shape[0]))
        self.handle.index_set_dev(rows.data_ptr(), rows.shape[0], rows.shape[1], id_base=self.start)

    def score_topk(self, queries, k, block=8192):
        """queries: CUDA float32 [Q,S] (identical on every rank).  Returns the global
        top-k (scores float64 [Q,k], row ids int64 [Q,k]) on every rank."""
        import torch
        import torch.distributed as dist
        Q = queries.shape[0]
        dev = queries.device
        stream = torch.cuda.current_stream(dev).cuda_stream if queries.is_cuda else 0
        out = torch.empty((2, Q, k), dtype=torch.int64, device=dev)        # [0] = float64 score bits, [1] = row ids
        fs, fi = out[0].view(torch.float64), out[1]
        if self.world == 1 and not self.always_gather:
            self.handle.score_topk_dev(queries.data_ptr(), Q, k, fs.data_ptr(), fi.data_ptr(), stream)
            return fs, fi
        world = dist.get_world_size(self.group)
        pending = None                                                     # (work, gathered, q0, n) of the previous block
        for q0 in list(range(0, Q, block)) + [None]:
            if q0 is not None:
                n = min(block, Q - q0)
                loc = torch.empty((2, n, k), dtype=torch.int64, device=dev)
                self.handle.score_topk_dev(queries[q0:q0 + n].data_ptr(), n, k, loc[0].data_ptr(), loc[1].data_ptr(), stream)
                g = torch.empty((world * 2, n, k), dtype=torch.int64, device=dev)   # concatenation along dim 0 (gloo and nccl)
                # one collective for scores and ids; async: RCCL's stream waits for the sweep just enqueued, the host
                # goes on to enqueue the next block's sweep
                work = dist.all_gather_into_tensor(g, loc, group=self.group, async_op=True)
                nxt = (work, g, loc, q0, n)
            else:
                nxt = None
            if pending is not None:
                work, g, _loc, p0, pn = pending
                work.wait()                                               # current stream waits for the gather
                self.handle.merge_topk_strided_dev(g.data_ptr(), g[1].data_ptr(), 2 * pn * k, world, pn, k,
                                                   fs[p0:p0 + pn].data_ptr(), fi[p0:p0 + pn].data_ptr(), stream)
            pending = nxt
        return fs, fi


class RcclShardedIndex(object):
    """The same sharded index WITHOUT torch.distributed: the exchange is the library's own entry point
    (sse_score_topk_sharded_dev: shard sweep -> ONE ncclAllGather of the packed lists -> k-way merge, all on one stream)
    over an RCCL communicator created through the C ABI.  What a reference-side integration that must not import torch
    uses; the host moves the 128-byte unique id from rank 0 to the other ranks itself."""

    def __init__(self, handle, rank, world, n_total, unique_id):
        self.handle, self.rank, self.world = handle, int(rank), int(world)
        self.n_total = int(n_total)
        self.start, self.end = shard_bounds(n_total, world)[rank]
        self.comm = handle.rccl_comm_init_rank(world, rank, unique_id)

    def close(self):
        if self.comm:
            self.handle.rccl_comm_destroy(self.comm)
            self.comm = None

    def set_local_rows_ptr(self, rows_ptr, n_rows, S, stream=0):
        """rows_ptr: device pointer of this rank's [end-start, S] float32 shard."""
        if n_rows != self.end - self.start:
            raise ValueError("shard of rank %d must have %d rows, got %d" % (self.rank, self.end - self.start, n_rows))
        self.handle.index_set_dev(rows_ptr, n_rows, S, id_base=self.start, stream=stream)

    def score_topk_ptr(self, q_ptr, Q, k, out_scores_ptr, out_ids_ptr, stream=0):
        """q_ptr: device float32 [Q,S] (identical on every rank); out_*: device float64 / int64 [Q,k]: the global top-k."""
        self.handle.score_topk_sharded_dev(self.comm, self.world, q_ptr, Q, k, out_scores_ptr, out_ids_ptr, stream)


def split_rows(n_rows, rank, world):
    """Independent units (sequences to encode): the slice of [0, n_rows) rank handles; no collective."""
    return shard_bounds(n_rows, world)[rank]
