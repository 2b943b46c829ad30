"""Data-parallel train step over the GPUs of one node (SURVEY 8e "Training"; BASELINE configs[4]).

Every rank holds the full model, runs forward + loss + backward on ITS pair rows with the loss defined as the mean
over the rows of all ranks, sums one flat float32 buffer over RCCL (dense gradients of every variable, the squared
norm of the raw embedding-gradient slices that `tf.clip_by_global_norm` sees, loss, train_acc, rows -- the "gradient
arena" of include/sse_hip.h), then clips by the global norm of the REDUCED gradients and applies Adagrad: the update is
bit-identical on every rank and equals the single-process step on the concatenated batch up to fp32 summation order.

The reference trains in one process (sse_train.py:170-172); this is the exchange step a multi-GPU job adds.
The engine is anything with train_grad_count / train_bind_arena / train_grads / train_apply: the HIP handle
(sequence-semantic-embedding_amd/_lib.py) in the product, the numpy oracle in the CPU `gloo` tests.
torch / torch.distributed are plumbing only.
"""


class DataParallelTrainer(object):
    def __init__(self, engine, device=None, group=None, always_reduce=False, sparse_embedding=None):
        """always_reduce: issue the collective even in a 1-rank group (exercises the RCCL path on one GPU).
        sparse_embedding: exchange the word-embedding gradient as (row id, gradient row) pairs -- SURVEY 8e's shape for
        large vocabularies (the 1M-title ranking vocabulary, reference README.md:116) -- instead of all-reducing the
        dense [V,E] block; None = automatic (sparse when the GLOBAL batch touches fewer than V/4 rows: 2 * rows_global * T < V / 4)."""
        import torch
        self.engine, self.group, self.always_reduce = engine, group, bool(always_reduce)
        self.sparse_embedding = sparse_embedding
        self.arena = torch.zeros(engine.train_grad_count(), dtype=torch.float32, device=device or "cpu")
        engine.train_bind_arena(self.arena)
        self._stream = None
        # where the dense word_embedding gradient sits in the arena: (offset, V, E); the HIP handle keeps it first
        sl = getattr(engine, "embedding_slice", None)
        self.emb_slice = tuple(sl()) if sl is not None else None
        self.last_exchange = None          # "dense" | "sparse": what the last step put on the wire (tests, bench)
        self._packed = self._gathered = None   # the packed (row id, gradient row) buffers of the sparse exchange

    @property
    def world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def global_rows(self, local_rows):
        """Sum of the ranks' row counts (ranks may hold different numbers of rows: the reference's batches are
        truncated at the end of the corpus, data.py:98)."""
        return self._rows_sum_max(local_rows)[0]

    def _rows_sum_max(self, local_rows):
        """(sum, max) of the ranks' row counts: one tiny all-gather (both are needed: the loss is a mean over the sum,
        the packed embedding exchange is sized by the max)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return int(local_rows), int(local_rows)
        t = torch.tensor([int(local_rows)], dtype=torch.int64, device=self.arena.device)
        out = torch.empty(self.world, dtype=torch.int64, device=self.arena.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        out = out.cpu()
        return int(out.sum()), int(out.max())

    def train_step(self, src_ids, tgt_ids, labels, rows_global=None, by_rows=False):
        """One step on this rank's rows; returns the GLOBAL (loss, train_acc), evaluated before the update.
        Pass rows_global when it is known and the batch is split evenly (split_batch: every rank holds at most
        ceil(rows_global / world) rows) to save the tiny extra all-gather of the row counts; ranks with arbitrary row counts
        (the reference's truncated batches, data.py:98) leave it None.
        by_rows: src_ids / tgt_ids are row numbers into the corpora uploaded with engine.corpus_upload."""
        import torch
        import torch.distributed as dist
        if self.arena.is_cuda and hasattr(self.engine, "set_stream"):
            # the train step runs on the library's own streams forked from / joined to ONE stream, and torch.distributed
            # orders the collective against torch's CURRENT stream: hand that stream to the library (sse_set_stream)
            cur = torch.cuda.current_stream(self.arena.device).cuda_stream
            if cur != self._stream:
                self.engine.set_stream(cur)
                self._stream = cur
        if rows_global is None:
            rows_global, rows_cap = self._rows_sum_max(len(labels))
        else:
            rows_cap = -(-int(rows_global) // self.world)
            if len(labels) > rows_cap:
                raise ValueError("rank holds %d rows, more than ceil(rows_global / world) = %d: leave rows_global=None for "
                                 "unevenly split batches" % (len(labels), rows_cap))
        if by_rows:
            self.engine.train_grads_rows(src_ids, tgt_ids, labels, rows_global)
        else:
            self.engine.train_grads(src_ids, tgt_ids, labels, rows_global)
        if self.world > 1 or self.always_reduce:
            if self._use_sparse(rows_global, src_ids, by_rows):
                self._exchange_sparse(rows_cap, self._seq_len(src_ids, by_rows))
            else:
                dist.all_reduce(self.arena, group=self.group)      # ONE collective per step (sum)
                self.last_exchange = "dense"
        return self.engine.train_apply()

    # ---- (row id, gradient row) exchange of the embedding gradient (SURVEY 8e "Training") ----------------------------
    def _seq_len(self, src_ids, by_rows):
        T = getattr(self.engine, "max_seq_length", None)
        if T is None:
            import numpy as np
            if by_rows:
                raise ValueError("engine without max_seq_length: cannot size the packed exchange of a step by rows")
            T = int(np.asarray(src_ids).shape[-1])
        return int(T)

    def _use_sparse(self, rows_global, src_ids, by_rows):
        """Decided from values that are IDENTICAL on every rank (rows_global, T, V): ranks whose local row counts differ
        by one (split_batch) must not disagree about which collectives the step issues."""
        if self.emb_slice is None or self.sparse_embedding is False:
            return False
        if self.sparse_embedding:
            return True
        V = self.emb_slice[1]
        return 2 * int(rows_global) * self._seq_len(src_ids, by_rows) < V // 4

    def _exchange_sparse(self, rows_cap, T):
        """Same sums as the dense all-reduce.  The dense variables and the tail go through one all-reduce of the arena
        BEHIND the embedding block (two when the block is not at the start); the embedding block travels as the rows this
        rank touched: compacted on the device into a FIXED-size packed buffer (cap = min(V, 2 * rows_cap * T) slots, the
        same on every rank: no count has to cross the host), ONE all-gather, then added into the zeroed block in rank
        order by the engine (HIP kernels behind the C ABI: sse_train_pack / unpack_embedding_grad) -- no torch kernels, no
        host synchronisation."""
        import torch
        import torch.distributed as dist
        off, V, E = self.emb_slice
        cap = max(1, min(V, 2 * int(rows_cap) * int(T)))
        n = self.engine.dp_packed_floats(cap)
        if self._packed is None or self._packed.numel() != n or self._gathered.numel() != n * self.world:
            self._packed = torch.zeros(n, dtype=torch.float32, device=self.arena.device)
            self._gathered = torch.zeros(n * self.world, dtype=torch.float32, device=self.arena.device)
        works = [dist.all_reduce(part, group=self.group, async_op=True)          # everything but the embedding block
                 for part in (self.arena[:off], self.arena[off + V * E:]) if part.numel()]
        self.engine.dp_pack_embedding(cap, self._packed)
        if self.world == 1 and not self.always_reduce:
            self._gathered.copy_(self._packed)
        else:
            dist.all_gather_into_tensor(self._gathered, self._packed, group=self.group)
        self.engine.dp_unpack_embedding(self._gathered, self.world, cap)
        for w in works:
            w.wait()
        self.last_exchange = "sparse"
        self.last_cap = cap


def split_batch(src_ids, tgt_ids, labels, rank, world):
    """Contiguous, balanced slice of a global batch for one rank (same rule as the index shards)."""
    from .sharded import shard_bounds
    s, e = shard_bounds(len(labels), world)[rank]
    return src_ids[s:e], tgt_ids[s:e], labels[s:e]
