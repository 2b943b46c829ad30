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

    @property
    def world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def global_rows(self, local_rows):
        """Sum of the ranks' row counts (ranks may hold different numbers of rows: the reference's batches are
        truncated at the end of the corpus, data.py:98)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return int(local_rows)
        t = torch.tensor([int(local_rows)], dtype=torch.int64, device=self.arena.device)
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def train_step(self, src_ids, tgt_ids, labels, rows_global=None, by_rows=False):
        """One step on this rank's rows; returns the GLOBAL (loss, train_acc), evaluated before the update.
        Pass rows_global when it is known (equal batches: world * len(labels)) to save the tiny extra all-reduce.
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
            rows_global = self.global_rows(len(labels))
        if by_rows:
            self.engine.train_grads_rows(src_ids, tgt_ids, labels, rows_global)
        else:
            self.engine.train_grads(src_ids, tgt_ids, labels, rows_global)
        if self.world > 1 or self.always_reduce:
            if self._use_sparse(rows_global, src_ids, by_rows):
                self._exchange_sparse()
            else:
                dist.all_reduce(self.arena, group=self.group)      # ONE collective per step (sum)
                self.last_exchange = "dense"
        return self.engine.train_apply()

    # ---- (row id, gradient row) exchange of the embedding gradient (SURVEY 8e "Training") ----------------------------
    def _use_sparse(self, rows_global, src_ids, by_rows):
        """Decided from values that are IDENTICAL on every rank (rows_global, T, V): ranks whose local row counts differ
        by one (split_batch) must not disagree about which collectives the step issues."""
        if self.emb_slice is None or self.sparse_embedding is False:
            return False
        if self.sparse_embedding:
            return True
        V = self.emb_slice[1]
        T = getattr(self.engine, "max_seq_length", None)
        if T is None:
            import numpy as np
            T = 1 if by_rows else int(np.asarray(src_ids).shape[-1])
        return 2 * int(rows_global) * int(T) < V // 4

    def _exchange_sparse(self):
        """Same sums as the dense all-reduce: the dense variables and the tail go through one all-reduce of the arena
        BEHIND the embedding block; the embedding block travels as the rows this rank touched (rows with any non-zero
        gradient), all-gathered with their ids and scatter-added into a zeroed block on every rank."""
        import torch
        import torch.distributed as dist
        off, V, E = self.emb_slice
        emb = self.arena[off:off + V * E].view(V, E)
        works = [dist.all_reduce(part, group=self.group, async_op=True)          # everything but the embedding block
                 for part in (self.arena[:off], self.arena[off + V * E:]) if part.numel()]
        ids = torch.nonzero((emb != 0).any(dim=1)).flatten()
        n = torch.tensor([ids.numel()], dtype=torch.int64, device=emb.device)
        counts = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(counts, n, group=self.group)
        nmax = max(1, int(max(c.item() for c in counts)))
        pad_ids = torch.full((nmax,), V, dtype=torch.int64, device=emb.device)       # V = "no row"
        pad_rows = torch.zeros((nmax, E), dtype=torch.float32, device=emb.device)
        pad_ids[:ids.numel()] = ids
        pad_rows[:ids.numel()] = emb[ids]
        all_ids = [torch.empty_like(pad_ids) for _ in range(self.world)]
        all_rows = [torch.empty_like(pad_rows) for _ in range(self.world)]
        dist.all_gather(all_ids, pad_ids, group=self.group)
        dist.all_gather(all_rows, pad_rows, group=self.group)
        emb.zero_()
        buf = torch.zeros((V + 1, E), dtype=torch.float32, device=emb.device)
        for r in range(self.world):                                   # fixed rank order: the same sum on every rank
            buf.index_add_(0, all_ids[r], all_rows[r])
        emb.copy_(buf[:V])
        for w in works:
            w.wait()
        self.last_exchange = "sparse"


def split_batch(src_ids, tgt_ids, labels, rank, world):
    """Contiguous, balanced slice of a global batch for one rank (same rule as the index shards)."""
    from .sharded import shard_bounds
    s, e = shard_bounds(len(labels), world)[rank]
    return src_ids[s:e], tgt_ids[s:e], labels[s:e]
