"""MI355X-native Sequence-Semantic-Embedding hot path: LSTM/CNN sequence
encoders, pairwise cosine loss, full-corpus cosine scoring with fused top-k --
hand-written gfx950 HIP kernels behind a C ABI (include/sse_hip.h)."""
from ._lib import Handle, SSEConfig, SSEError, load_library, LIB_PATH, SYMBOLS  # noqa: F401
from .sse_model import SSEModel, Session, Saver, get_checkpoint_state  # noqa: F401
from .sharded import ShardedIndex, RcclShardedIndex, shard_bounds, all_gather_topk, split_rows  # noqa: F401
from .data_parallel import DataParallelTrainer, split_batch  # noqa: F401
