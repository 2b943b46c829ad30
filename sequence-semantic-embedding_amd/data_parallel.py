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
    def __init__(self, engine, device=None, group=None, always_reduce=False):
        """always_reduce: issue the collective even in a 1-rank group (exercises the RCCL path on one GPU)."""
        import torch
        self.engine, self.group, self.always_reduce = engine, group, bool(always_reduce)
        self.arena = torch.zeros(engine.train_grad_count(), dtype=torch.float32, device=device or "cpu")
        engine.train_bind_arena(self.arena)

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
        if self.arena.is_cuda and torch.cuda.current_stream(self.arena.device) != torch.cuda.default_stream(self.arena.device):
            # the train step runs on the library's own streams forked from / joined to the null stream, and
            # torch.distributed orders the all-reduce against torch's CURRENT stream: they must be the same one
            raise RuntimeError("DataParallelTrainer.train_step must be called with the default CUDA stream current")
        if rows_global is None:
            rows_global = self.global_rows(len(labels))
        if by_rows:
            self.engine.train_grads_rows(src_ids, tgt_ids, labels, rows_global)
        else:
            self.engine.train_grads(src_ids, tgt_ids, labels, rows_global)
        if self.world > 1 or self.always_reduce:
            dist.all_reduce(self.arena, group=self.group)          # ONE collective per step (sum)
        return self.engine.train_apply()


def split_batch(src_ids, tgt_ids, labels, rank, world):
    """Contiguous, balanced slice of a global batch for one rank (same rule as the index shards)."""
    from .sharded import shard_bounds
    s, e = shard_bounds(len(labels), world)[rank]
    return src_ids[s:e], tgt_ids[s:e], labels[s:e]
