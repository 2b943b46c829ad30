"""ctypes binding of libsse_hip.so (C ABI in include/sse_hip.h).

There is NO CPU fallback: if the library is missing or no HIP device is
visible, loading / handle creation raises.
"""
import atexit
import ctypes as C
import weakref
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# SSE_HIP_LIB: developer knob for A/B timing of two builds of the same C ABI (tools/ab); never a CPU fallback
LIB_PATH = os.environ.get("SSE_HIP_LIB") or os.path.join(HERE, "libsse_hip.so")

MODE_IDS = {"dual-encoder": 0, "shared-encoder": 1, "source-encoder-only": 2, "source_only_cnn": 3}
SIDE_SOURCE, SIDE_TARGET = 0, 1


class SSEConfig(C.Structure):
    _fields_ = [("network_mode", C.c_int32), ("vocab_size", C.c_int32), ("embedding_size", C.c_int32),
                ("encoding_size", C.c_int32), ("src_cell_size", C.c_int32), ("tgt_cell_size", C.c_int32),
                ("max_seq_length", C.c_int32), ("target_space_size", C.c_int32), ("device", C.c_int32),
                ("learning_rate", C.c_float), ("learning_rate_decay_factor", C.c_float)]


# every symbol declared in include/sse_hip.h: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "sse_create": (C.c_int, [C.POINTER(SSEConfig), C.POINTER(_P)]),
    "sse_destroy": (None, [_P]),
    "sse_last_error": (C.c_char_p, [_P]),
    "sse_num_variables": (C.c_int, [_P]),
    "sse_variable_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sse_set_variable": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "sse_get_variable": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "sse_encode": (C.c_int, [_P, C.c_int, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "sse_encode_dev": (C.c_int, [_P, C.c_int, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "sse_set_option": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "sse_get_counter": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "sse_l2_normalize_dev": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P]),
    "sse_index_upload": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64]),
    "sse_index_upload_f64": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64]),
    "sse_index_set_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64, _P]),
    "sse_score_topk": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "sse_score_topk_dev": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "sse_encode_score_topk": (C.c_int, [_P, C.c_int, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "sse_merge_topk_dev": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "sse_merge_topk_strided_dev": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "sse_rccl_library_path": (C.c_char_p, []),
    "sse_rccl_group_start": (C.c_int, []),
    "sse_rccl_group_end": (C.c_int, []),
    "sse_rccl_get_unique_id": (C.c_int, [_P]),
    "sse_rccl_comm_init_rank": (C.c_int, [_P, C.POINTER(_P), C.c_int32, C.c_int32, _P]),
    "sse_rccl_comm_destroy": (C.c_int, [_P, _P]),
    "sse_allgather_merge_topk_dev": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "sse_score_topk_sharded_dev": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "sse_train_step": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sse_set_stream": (C.c_int, [_P, _P]),
    "sse_train_grad_count": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "sse_train_set_grad_arena": (C.c_int, [_P, _P, C.c_int64]),
    "sse_train_grads": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int64]),
    "sse_train_apply": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sse_train_packed_embedding_floats": (C.c_int64, [_P, C.c_int32]),
    "sse_train_pack_embedding_grad": (C.c_int, [_P, C.c_int32, _P]),
    "sse_train_unpack_embedding_grad": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "sse_corpus_upload": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int32]),
    "sse_train_step_rows": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sse_train_grads_rows": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int64]),
    "sse_get_learning_rate": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "sse_set_learning_rate": (C.c_int, [_P, C.c_float]),
    "sse_decay_learning_rate": (C.c_int, [_P]),
    "sse_get_global_step": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "sse_set_global_step": (C.c_int, [_P, C.c_int64]),
    "sse_timer_record": (C.c_int, [_P, C.c_int32, _P]),
    "sse_timer_elapsed_ms": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
    "sse_synchronize": (C.c_int, [_P]),
    "sse_format_rows_stride": (C.c_int64, [C.c_int32]),
    "sse_format_rows_f32": (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P]),
    "sse_parse_rows_f64": (C.c_int, [C.c_char_p, _P, C.c_int64, C.c_int32, _P, C.POINTER(C.c_int64)]),
    "sse_crc32c": (C.c_uint32, [_P, C.c_int64, C.c_uint32]),
}

_lib = None
_host_lib = None
HOST_LIB_PATH = os.path.join(HERE, "libsse_host.so")
HOST_SYMBOLS = ("sse_format_rows_stride", "sse_format_rows_f32", "sse_parse_rows_f64", "sse_crc32c")


def load_host_library():
    """The host-only entry points of include/sse_hip.h (index text I/O, CRC-32C) from libsse_host.so: the same C code
    (csrc/index_io.cpp) linked WITHOUT the HIP runtime, so that reading a TF checkpoint or an index file neither imports
    torch nor touches a GPU.  Falls back to the full library when only that one is built."""
    global _host_lib
    if _host_lib is not None:
        return _host_lib
    if _lib is not None or not os.path.exists(HOST_LIB_PATH):
        _host_lib = load_library()
        return _host_lib
    lib = C.CDLL(HOST_LIB_PATH)
    for name in HOST_SYMBOLS:
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = SYMBOLS[name]
    _host_lib = lib
    return lib


def load_library():
    """dlopen libsse_hip.so (once).  When torch is importable it is imported
    FIRST so that both share one HIP runtime (torch bundles its own
    libamdhip64.so.7; two copies in one process cannot share device memory)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsse_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    if "torch" not in sys.modules and os.environ.get("SSE_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class SSEError(RuntimeError):
    pass


# Handles still alive when the interpreter exits are closed by an atexit hook, i.e. BEFORE module teardown and before the C-level
# exit handlers: Handle.__del__ otherwise runs during interpreter finalisation, when the HIP runtime (and, under rocprofv3, the
# profiler's intercept layer) may already be on its way out.
_live_handles = weakref.WeakSet()


def _close_live_handles():
    for h in list(_live_handles):
        try:
            h.close()
        except Exception:
            pass


atexit.register(_close_live_handles)


class Handle(object):
    """Owner of one `sse_handle*`."""

    def __init__(self, cfg):
        self.lib = load_library()
        self._h = C.c_void_p()
        _live_handles.add(self)
        self.index_gen = 0          # bumped by every index upload: lets holders of "their" index notice a replacement
        rc = self.lib.sse_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise SSEError(self.lib.sse_last_error(None).decode())
        self.cfg = cfg
        # SSE_OPTIONS="name=value,name=value": library options (sse_set_option) for any command line without touching it,
        # e.g. SSE_OPTIONS=lstm_x3=1 python sse_index.py ...; an option the network mode rejects fails loudly
        for item in filter(None, (x.strip() for x in os.environ.get("SSE_OPTIONS", "").split(","))):
            name, _, value = item.partition("=")
            self.set_option(name.strip(), int(value or "1"))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.sse_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise SSEError(self.lib.sse_last_error(self._h).decode())

    # -- variables ---------------------------------------------------------
    def variables(self):
        out = []
        name, cnt, r, c = C.c_char_p(), C.c_int64(), C.c_int32(), C.c_int32()
        for i in range(self.lib.sse_num_variables(self._h)):
            self.check(self.lib.sse_variable_info(self._h, i, C.byref(name), C.byref(cnt), C.byref(r), C.byref(c)))
            out.append((name.value.decode(), int(cnt.value), int(r.value), int(c.value)))
        return out

    def set_variable(self, name, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        self.check(self.lib.sse_set_variable(self._h, name.encode(), _ptr(a), a.size))

    def get_variable(self, name, count):
        a = np.empty(count, np.float32)
        self.check(self.lib.sse_get_variable(self._h, name.encode(), _ptr(a), a.size))
        return a

    # -- encode ------------------------------------------------------------
    def encode(self, side, ids, normalize=True):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if ids.ndim != 2:
            raise ValueError("ids must be [B,T]")
        B, T = ids.shape
        out = np.empty((B, self.cfg.encoding_size), np.float32)
        self.check(self.lib.sse_encode(self._h, side, _ptr(ids), B, T, 1 if normalize else 0, _ptr(out)))
        return out

    def encode_dev(self, side, ids_ptr, B, T, normalize, out_ptr, stream=0):
        self.check(self.lib.sse_encode_dev(self._h, side, ids_ptr, B, T, 1 if normalize else 0, out_ptr, stream))

    def set_option(self, name, value):
        self.check(self.lib.sse_set_option(self._h, name.encode(), int(value)))

    def get_counter(self, name):
        v = C.c_int64()
        self.check(self.lib.sse_get_counter(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def l2_normalize_dev(self, x_ptr, out_ptr, rows, cols, stream=0):
        self.check(self.lib.sse_l2_normalize_dev(self._h, x_ptr, out_ptr, rows, cols, stream))

    # -- index / scoring -----------------------------------------------------
    def index_upload(self, rows, id_base=0):
        rows = np.ascontiguousarray(rows)
        if rows.ndim != 2:
            raise ValueError("index rows must be [N,S]")
        N, S = rows.shape
        if rows.dtype == np.float64:
            self.check(self.lib.sse_index_upload_f64(self._h, _ptr(rows), N, S, id_base))
        else:
            rows = np.ascontiguousarray(rows, dtype=np.float32)
            self.check(self.lib.sse_index_upload(self._h, _ptr(rows), N, S, id_base))
        self.index_gen += 1

    def index_set_dev(self, rows_ptr, N, S, id_base=0, stream=0):
        self.check(self.lib.sse_index_set_dev(self._h, rows_ptr, N, S, id_base, stream))
        self.index_gen += 1

    def score_topk(self, queries, k):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2:
            raise ValueError("queries must be [Q,S]")
        Q = q.shape[0]
        scores = np.empty((Q, k), np.float64)
        ids = np.empty((Q, k), np.int64)
        self.check(self.lib.sse_score_topk(self._h, _ptr(q), Q, k, _ptr(scores), _ptr(ids)))
        return scores, ids

    def encode_score_topk(self, side, ids, normalize, k, want_encodings=False):
        """encode + cosine top-k with the encodings staying on the device (sse_demo.py:121-129,
        sse_evaluator.py:107-111); returns (scores [B,k] f64, ids [B,k] i64[, encodings [B,S] f32])."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if ids.ndim != 2:
            raise ValueError("ids must be [B,T]")
        B, T = ids.shape
        scores = np.empty((B, k), np.float64)
        rows = np.empty((B, k), np.int64)
        enc = np.empty((B, self.cfg.encoding_size), np.float32) if want_encodings else None
        self.check(self.lib.sse_encode_score_topk(self._h, side, _ptr(ids), B, T, 1 if normalize else 0, int(k),
                                                  _ptr(scores), _ptr(rows), _ptr(enc) if want_encodings else None))
        return (scores, rows, enc) if want_encodings else (scores, rows)

    def score_topk_dev(self, q_ptr, Q, k, scores_ptr, ids_ptr, stream=0):
        self.check(self.lib.sse_score_topk_dev(self._h, q_ptr, Q, k, scores_ptr, ids_ptr, stream))

    def merge_topk_strided_dev(self, in_s, in_i, shard_stride, P, Q, k, out_s, out_i, stream=0):
        self.check(self.lib.sse_merge_topk_strided_dev(self._h, in_s, in_i, int(shard_stride), P, Q, k, out_s, out_i, stream))

    def merge_topk_dev(self, in_s, in_i, P, Q, k, out_s, out_i, stream=0):
        self.check(self.lib.sse_merge_topk_dev(self._h, in_s, in_i, P, Q, k, out_s, out_i, stream))

    # -- RCCL exchange without torch ------------------------------
    def rccl_unique_id(self):
        """A fresh 128-byte ncclUniqueId (rank 0 creates it; the other ranks receive the bytes)."""
        buf = C.create_string_buffer(128)
        if self.lib.sse_rccl_get_unique_id(buf) != 0:
            raise SSEError("RCCL (librccl.so.1) is not loadable in this process")
        return buf.raw

    def rccl_library_path(self):
        """The ONE RCCL instance the library bound ($SSE_RCCL_LIB, else one already mapped into the process, else
        librccl.so.1); None when no RCCL is loadable.  Communicators must come from this instance."""
        p = self.lib.sse_rccl_library_path()
        return p.decode() if p else None

    def rccl_group_start(self):
        if self.lib.sse_rccl_group_start() != 0:
            raise SSEError("ncclGroupStart failed (or RCCL is not loadable in this process)")

    def rccl_group_end(self):
        if self.lib.sse_rccl_group_end() != 0:
            raise SSEError("ncclGroupEnd failed (or RCCL is not loadable in this process)")

    def rccl_comm_init_rank(self, world, rank, unique_id):
        """ncclCommInitRank on the handle's device; returns the communicator as an integer handle (void*)."""
        comm = _P()
        self.check(self.lib.sse_rccl_comm_init_rank(self._h, C.byref(comm), int(world), int(rank), C.c_char_p(bytes(unique_id))))
        return comm.value

    def rccl_comm_destroy(self, comm):
        self.check(self.lib.sse_rccl_comm_destroy(self._h, comm))

    def allgather_merge_topk_dev(self, comm, world, loc_s, loc_i, Q, k, out_s, out_i, stream=0):
        self.check(self.lib.sse_allgather_merge_topk_dev(self._h, comm, int(world), loc_s, loc_i, Q, k, out_s, out_i, stream))

    def score_topk_sharded_dev(self, comm, world, q_ptr, Q, k, out_s, out_i, stream=0):
        self.check(self.lib.sse_score_topk_sharded_dev(self._h, comm, int(world), q_ptr, Q, k, out_s, out_i, stream))

    # -- training ------------------------------------------------------------
    @staticmethod
    def _train_batch(src_ids, tgt_ids, labels):
        """src [B,T] int32; tgt [B,T] token ids -- or, for source_only_cnn, [B] rows of the free target matrix
        (the library checks which one the network mode wants); labels float32 [B]."""
        s = np.ascontiguousarray(src_ids, dtype=np.int32)
        t = np.ascontiguousarray(tgt_ids, dtype=np.int32)
        z = np.ascontiguousarray(labels, dtype=np.float32)
        if t.ndim == 2 and t.shape[1] == 1 and s.ndim == 2 and s.shape[1] != 1:
            t = np.ascontiguousarray(t[:, 0])
        if s.ndim != 2 or z.shape != (s.shape[0],) or not (t.shape == s.shape or t.shape == (s.shape[0],)):
            raise ValueError("train batch shapes: src [B,T], tgt [B,T] (or [B] target rows), labels [B]")
        return s, t, z

    def _check_tgt_kind(self, t):
        table = self.cfg.network_mode in (2, 3)                    # source-encoder-only, source_only_cnn
        if table != (t.ndim == 1):
            raise ValueError("source-encoder-only / source_only_cnn train on [B] target-matrix rows, "
                             "dual- and shared-encoder on [B,T] target token ids")

    def embedding_slice(self):
        """(offset, V, E) of the dense word_embedding gradient inside the gradient arena (variable 0 comes first)."""
        return 0, int(self.cfg.vocab_size), int(self.cfg.embedding_size)

    @property
    def max_seq_length(self):
        return int(self.cfg.max_seq_length)

    def set_stream(self, stream=0):
        """The stream (hipStream_t as int; 0 = null stream) the train-step entry points enqueue on."""
        self.check(self.lib.sse_set_stream(self._h, C.c_void_p(stream) if stream else None))
        self._stream = int(stream or 0)

    def train_step(self, src_ids, tgt_ids, labels):
        s, t, z = self._train_batch(src_ids, tgt_ids, labels)
        self._check_tgt_kind(t)
        loss, acc = C.c_float(), C.c_float()
        self.check(self.lib.sse_train_step(self._h, _ptr(s), _ptr(t), _ptr(z), s.shape[0], s.shape[1],
                                           C.byref(loss), C.byref(acc)))
        return float(loss.value), float(acc.value)

    # ---- batches by row number: corpora resident on the device
    def corpus_upload(self, side, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        if ids.ndim != 2:
            raise ValueError("corpus must be [N,T]")
        self.check(self.lib.sse_corpus_upload(self._h, int(side), _ptr(ids), ids.shape[0], ids.shape[1]))

    def _rows_batch(self, src_rows, tgt_rows, labels):
        s = np.ascontiguousarray(src_rows, dtype=np.int32).reshape(-1)
        t = np.ascontiguousarray(tgt_rows, dtype=np.int32).reshape(-1)
        z = np.ascontiguousarray(labels, dtype=np.float32).reshape(-1)
        if not (s.shape == t.shape == z.shape):
            raise ValueError("src_rows, tgt_rows, labels must all be [B]")
        return s, t, z

    def train_step_rows(self, src_rows, tgt_rows, labels):
        s, t, z = self._rows_batch(src_rows, tgt_rows, labels)
        loss, acc = C.c_float(), C.c_float()
        self.check(self.lib.sse_train_step_rows(self._h, _ptr(s), _ptr(t), _ptr(z), s.shape[0], C.byref(loss), C.byref(acc)))
        return float(loss.value), float(acc.value)

    def train_grads_rows(self, src_rows, tgt_rows, labels, rows_global):
        s, t, z = self._rows_batch(src_rows, tgt_rows, labels)
        self.check(self.lib.sse_train_grads_rows(self._h, _ptr(s), _ptr(t), _ptr(z), s.shape[0], int(rows_global)))

    # ---- data-parallel split of the train step: grads -> all-reduce(arena) -> apply
    def train_grad_count(self):
        n = C.c_int64()
        self.check(self.lib.sse_train_grad_count(self._h, C.byref(n)))
        return int(n.value)

    def train_set_grad_arena(self, dev_ptr, count):
        """Hand the library a caller-owned float32 device buffer (e.g. tensor.data_ptr()) of train_grad_count()
        floats for the gradients; dev_ptr=None returns to a library-owned one.  The caller keeps it alive."""
        self.check(self.lib.sse_train_set_grad_arena(self._h, C.c_void_p(dev_ptr) if dev_ptr else None, int(count)))

    def train_bind_arena(self, tensor):
        """tensor: contiguous float32 CUDA tensor of train_grad_count() elements on the handle's device."""
        if not tensor.is_cuda or not tensor.is_contiguous() or str(tensor.dtype) != "torch.float32":
            raise ValueError("gradient arena must be a contiguous float32 CUDA tensor")
        self._arena = tensor                                       # keep it alive
        self.train_set_grad_arena(tensor.data_ptr(), tensor.numel())

    def train_grads(self, src_ids, tgt_ids, labels, rows_global=None):
        s, t, z = self._train_batch(src_ids, tgt_ids, labels)
        self._check_tgt_kind(t)
        self.check(self.lib.sse_train_grads(self._h, _ptr(s), _ptr(t), _ptr(z), s.shape[0], s.shape[1],
                                            int(rows_global if rows_global is not None else s.shape[0])))

    def train_apply(self):
        loss, acc = C.c_float(), C.c_float()
        self.check(self.lib.sse_train_apply(self._h, C.byref(loss), C.byref(acc)))
        return float(loss.value), float(acc.value)

    # ---- (row id, gradient row) exchange of the word-embedding gradient (data_parallel.py)
    def dp_packed_floats(self, cap):
        return int(self.lib.sse_train_packed_embedding_floats(self._h, int(cap)))

    def dp_pack_embedding(self, cap, packed):
        """packed: contiguous float32 CUDA tensor of dp_packed_floats(cap) elements."""
        self.check(self.lib.sse_train_pack_embedding_grad(self._h, int(cap), C.c_void_p(packed.data_ptr())))

    def dp_unpack_embedding(self, gathered, world, cap):
        """gathered: the `world` packed buffers back to back (all_gather_into_tensor), float32 CUDA."""
        self.check(self.lib.sse_train_unpack_embedding_grad(self._h, C.c_void_p(gathered.data_ptr()), int(world), int(cap)))

    @property
    def learning_rate(self):
        v = C.c_float()
        self.check(self.lib.sse_get_learning_rate(self._h, C.byref(v)))
        return float(v.value)

    @learning_rate.setter
    def learning_rate(self, lr):
        self.check(self.lib.sse_set_learning_rate(self._h, float(lr)))

    def decay_learning_rate(self):
        self.check(self.lib.sse_decay_learning_rate(self._h))

    @property
    def global_step(self):
        v = C.c_int64()
        self.check(self.lib.sse_get_global_step(self._h, C.byref(v)))
        return int(v.value)

    @global_step.setter
    def global_step(self, step):
        self.check(self.lib.sse_set_global_step(self._h, int(step)))

    # -- timing --------------------------------------------------------------
    def timer_record(self, slot, stream=0):
        self.check(self.lib.sse_timer_record(self._h, slot, stream))

    def timer_elapsed_ms(self, a, b):
        v = C.c_float()
        self.check(self.lib.sse_timer_elapsed_ms(self._h, a, b, C.byref(v)))
        return float(v.value)

    def synchronize(self):
        self.check(self.lib.sse_synchronize(self._h))
