"""Host-side mirror of the reference's model interface for the hot path.

The reference's boundary is `SSEModel(modelParams)` + `session.run(fetches,
feed_dict)` (sse_model.py:94, sse_train.py:170-172, sse_index.py:90-92,
sse_evaluator.py:107-109, sse_demo.py:121-125).  This module keeps that
contract -- same constructor argument, same feed-dict builders, same fetch
attribute names -- on top of the C ABI in include/sse_hip.h; there is no
TensorFlow and no CPU fallback.
"""
import os

import numpy as np

from . import _lib


class _Sym(object):
    """A graph-node stand-in: something `Session.run` can be asked to fetch or feed."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<sse %s>" % self.name


class _Scalar(_Sym):
    """`model.learning_rate` / `model.global_step`: support `.eval()` like tf.Variable (sse_train.py:181)."""

    def __init__(self, name, getter):
        _Sym.__init__(self, name)
        self._getter = getter

    def eval(self, session=None):
        return self._getter()


class Saver(object):
    """tf.train.Saver stand-in (sse_model.py:138): one .npz per checkpoint holding
    every variable under its TF name, the '<name>/Adagrad' slots, learning_rate
    and global_step; a `checkpoint` text file names the latest one
    (tf.train.get_checkpoint_state, sse_train.py:110-113)."""

    def __init__(self, model, max_to_keep=20):
        self.model = model
        self.max_to_keep = max_to_keep
        self._kept = []
        self.restored_from = None       # the file the last restore() read (.npz of this library, or a TensorFlow .index)

    def save(self, session, path, global_step=None):
        if global_step is not None:
            path = "%s-%d" % (path, int(global_step))
        arrays = self.model.get_variables(with_slots=True)
        arrays["learning_rate"] = np.float32(self.model.handle.learning_rate)
        arrays["global_step"] = np.int64(self.model.handle.global_step)
        tmp = path + ".tmp.npz"
        np.savez(tmp, **arrays)
        os.replace(tmp, path + ".npz")
        d = os.path.dirname(path) or "."
        with open(os.path.join(d, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))
        if path in self._kept:
            self._kept.remove(path)
        self._kept.append(path)
        while len(self._kept) > self.max_to_keep:
            old = self._kept.pop(0)
            if os.path.exists(old + ".npz"):
                os.remove(old + ".npz")
        return path

    def restore(self, session, path):
        """`path`: checkpoint prefix as tf.train.get_checkpoint_state reports it.  A `.npz` written by save() is
        read directly; a TensorFlow V2 checkpoint of the reference (`<prefix>.index` + `.data-*`, same variable
        names) is read by tf_checkpoint.read_bundle -- no TensorFlow needed."""
        from . import tf_checkpoint
        npz = path if path.endswith(".npz") else path + ".npz"
        has_tf = (not path.endswith(".npz")) and tf_checkpoint.is_tf_checkpoint(path)
        if os.path.exists(npz) and has_tf and os.path.getmtime(path + ".index") > os.path.getmtime(npz):
            # a converted .npz left behind by an earlier run must not shadow a checkpoint TensorFlow has re-written since
            # under the same prefix: the newer file wins
            self.restored_from = path + ".index"
            arrays = tf_checkpoint.to_npz_arrays(tf_checkpoint.read_bundle(path))
        elif os.path.exists(npz):
            self.restored_from = npz
            z = np.load(npz)
            arrays = {k: z[k] for k in z.files}
        elif has_tf:
            self.restored_from = path + ".index"
            arrays = tf_checkpoint.to_npz_arrays(tf_checkpoint.read_bundle(path))
        else:
            raise FileNotFoundError("no checkpoint at %s (.npz, or TensorFlow .index/.data-*)" % path)
        known = set(self.model.variable_names())
        weights = {k: v for k, v in arrays.items() if k not in ("learning_rate", "global_step")}
        missing = sorted(known - set(weights))
        if missing:
            raise KeyError("checkpoint %s lacks variables %s (network_mode / sizes differ from modelConfig.param?)"
                           % (path, missing[:4]))
        self.model.set_variables({k: v for k, v in weights.items()
                                  if (k[:-len("/Adagrad")] if k.endswith("/Adagrad") else k) in known})
        if "learning_rate" in arrays:
            self.model.handle.learning_rate = float(arrays["learning_rate"])
        if "global_step" in arrays:
            self.model.handle.global_step = int(arrays["global_step"])


def get_checkpoint_state(model_dir):
    """Latest checkpoint path in model_dir or None (tf.train.get_checkpoint_state)."""
    p = os.path.join(model_dir, "checkpoint")
    if not os.path.exists(p):
        return None
    for line in open(p):
        if line.startswith("model_checkpoint_path:"):
            name = line.split(":", 1)[1].strip().strip('"')
            full = name if os.path.isabs(name) else os.path.join(model_dir, name)
            if os.path.exists(full + ".npz") or os.path.exists(full + ".index"):
                return full                    # ours (.npz) or a TensorFlow V2 checkpoint of the reference (.index)
            # a `checkpoint` file that names something unreadable must not look like "no checkpoint": sse_train would
            # silently start from fresh weights and write next to the real model
            raise FileNotFoundError("%s names checkpoint %r but neither %s.npz nor %s.index exists"
                                    % (p, name, full, full))
    return None


class SSEModel(object):
    """Same constructor contract as the reference (sse_model.py:94-126): a dict
    whose values may be strings (as loaded from modelConfig.param)."""

    def __init__(self, modelParams, device=0):
        self.name = "SSEmodel"
        self.forward_only = bool(modelParams["forward_only"])
        self.network_mode = modelParams["network_mode"]
        self.TOP_N = int(modelParams["predict_nbest"])
        self.MAX_SEQ_LENGTH = int(modelParams["max_seq_length"])
        self.max_gradient_norm = 5.0
        self.vocab_size = int(modelParams["vocab_size"])
        self.word_embed_size = int(modelParams["embedding_size"])
        self.seq_embed_size = int(modelParams["encoding_size"])
        self.src_cell_size = int(modelParams["src_cell_size"])
        self.tgt_cell_size = int(modelParams["tgt_cell_size"])
        self.targetSpaceSize = int(modelParams["targetSpaceSize"])
        if self.network_mode not in _lib.MODE_IDS:
            # sse_model.py:175-177 prints and exit(-1)s; raise instead of killing the host process
            raise ValueError("Error!! Unsupported network mode: %s. Please specify on: source-encoder-only, "
                             "dual-encoder or shared-encoder." % self.network_mode)
        cfg = _lib.SSEConfig(_lib.MODE_IDS[self.network_mode], self.vocab_size, self.word_embed_size,
                             self.seq_embed_size, self.src_cell_size, self.tgt_cell_size, self.MAX_SEQ_LENGTH,
                             self.targetSpaceSize, int(device), float(modelParams["learning_rate"]),
                             float(modelParams["learning_rate_decay_factor"]))
        self.handle = _lib.Handle(cfg)
        self._shapes = {n: (r, c, cnt) for n, cnt, r, c in self.handle.variables()}

        # feedable placeholders (sse_model.py:153-155)
        self._src_input_data = _Sym("source_sequence")
        self._tgt_input_data = _Sym("target_sequence")
        self._labels = _Sym("targetSpace_labels")
        # fetchable tensors / ops (sse_model.py:245,254,282-283,298,302,363)
        self.src_seq_embedding = _Sym("src_seq_embedding")
        self.tgt_seq_embedding = _Sym("tgt_seq_embedding")
        self.norm_src_seq_embedding = _Sym("norm_src_seq_embedding")
        self.norm_tgt_seq_embedding = _Sym("norm_tgt_seq_embedding")
        # in-graph prediction (sse_model.py:344-352): top-N of the all-pairs cosine matrix, scores l2-normalised
        self.predicted_tgts_score = _Sym("predicted_tgts_score")
        self.predicted_labels = _Sym("predicted_labels")
        self.similarity = _Sym("similarity")
        self.loss = _Sym("loss")
        self.train_acc = _Sym("train_acc")
        self.train = _Sym("train")
        self.learning_rate = _Scalar("learning_rate", lambda: self.handle.learning_rate)
        self.global_step = _Scalar("global_step", lambda: self.handle.global_step)
        self.learning_rate_decay_op = _Sym("learning_rate_decay_op")
        self.saver = Saver(self, max_to_keep=20)

    # -- variables -----------------------------------------------------------
    def variable_names(self):
        return list(self._shapes)

    def variable_shape(self, name):
        r, c, cnt = self._shapes[name]
        return (r, c)

    def get_variables(self, with_slots=False):
        out = {}
        for n, (r, c, cnt) in self._shapes.items():
            out[n] = self.handle.get_variable(n, cnt).reshape(r, c)
            if with_slots:
                out[n + "/Adagrad"] = self.handle.get_variable(n + "/Adagrad", cnt).reshape(r, c)
        return out

    def set_variables(self, arrays):
        """arrays: {TF variable name: ndarray}.  Shapes are the TF ones; any
        layout with the right element count in row-major order is accepted."""
        for n, a in arrays.items():
            base = n[:-len("/Adagrad")] if n.endswith("/Adagrad") else n
            if base not in self._shapes:
                raise KeyError("unknown variable %r" % n)
            a = np.asarray(a, np.float32)
            if a.size != self._shapes[base][2]:
                raise ValueError("variable %s: expected %d elements, got %d" % (n, self._shapes[base][2], a.size))
            self.handle.set_variable(n, a.reshape(-1))

    def init_variables(self, seed=None):
        """tf.global_variables_initializer() with the reference initialisers
        (sse_model.py:161-162,243-244,...; BasicLSTMCell: glorot-uniform kernel,
        zero bias; Adagrad slots 0.1)."""
        rng = np.random.RandomState(seed)
        E = self.word_embed_size

        def trunc_normal(shape, std=1.0):
            x = rng.standard_normal(size=shape)
            bad = np.abs(x) > 2.0
            while bad.any():
                x[bad] = rng.standard_normal(size=int(bad.sum()))
                bad = np.abs(x) > 2.0
            return (x * std).astype(np.float32)

        arrays = {}
        for n, (r, c, cnt) in self._shapes.items():
            if n == "word_embedding" or n.endswith("tgt_seq_embedding"):
                arrays[n] = rng.uniform(-0.25, 0.25, size=(r, c)).astype(np.float32)
            elif n.endswith("basic_lstm_cell/kernel"):
                lim = np.sqrt(6.0 / (r + c))
                arrays[n] = rng.uniform(-lim, lim, size=(r, c)).astype(np.float32)
            elif n.endswith("basic_lstm_cell/bias"):
                arrays[n] = np.zeros((r, c), np.float32)
            elif n.endswith("/W"):
                arrays[n] = trunc_normal((r, c), 0.1)
            elif n.endswith("/b"):
                arrays[n] = np.full((r, c), 0.1, np.float32)
            elif n.endswith("_M"):
                arrays[n] = trunc_normal((r, c))
            else:
                raise RuntimeError("no initialiser for %s" % n)
            arrays[n + "/Adagrad"] = np.full((r, c), 0.1, np.float32)
        self.set_variables(arrays)

    # -- direct calls ----------------------------------------------------------
    def encode_source(self, ids, normalize=True):
        return self.handle.encode(_lib.SIDE_SOURCE, ids, normalize)

    def encode_target(self, ids, normalize=True):
        return self.handle.encode(_lib.SIDE_TARGET, ids, normalize)

    def train_step(self, src_ids, tgt_ids, labels):
        return self.handle.train_step(src_ids, tgt_ids, labels)

    def predict(self, src_ids, tgt_ids, top_n=None):
        """`_def_predict` (sse_model.py:344-352): tf.nn.top_k(similarity, TOP_N) over the batch's
        own targets, then the k scores l2-normalised per row.  Returns (scores float32 [Bs,k],
        labels int32 [Bs,k]); the [Bs,Bt] matrix itself is never materialised (fused top-k)."""
        k = min(int(top_n or self.TOP_N), len(tgt_ids))
        ns, nt = self.encode_source(src_ids, True), self.encode_target(tgt_ids, True)
        self.handle.index_upload(nt)
        sc, idx = self.handle.score_topk(ns, k)
        sc = sc.astype(np.float32)
        sc = sc / np.sqrt(np.maximum(np.sum(sc * sc, axis=1, keepdims=True), np.float32(1e-12)))
        return sc.astype(np.float32), idx.astype(np.int32)

    def similarity_matrix(self, src_ids, tgt_ids):
        """`self.similarity` (sse_model.py:286) for callers that really want all pairs: every column
        via the same fused kernel (k = Bt); intended for small Bt."""
        ns, nt = self.encode_source(src_ids, True), self.encode_target(tgt_ids, True)
        self.handle.index_upload(nt)
        sc, idx = self.handle.score_topk(ns, len(nt))
        out = np.empty((len(ns), len(nt)), np.float32)
        np.put_along_axis(out, idx, sc.astype(np.float32), axis=1)
        return out

    # -- reference surface ---------------------------------------------------
    def set_top_n(self, top_n):
        self.TOP_N = top_n

    def set_forward_only(self, forward_only=True):
        self.forward_only = forward_only

    def save(self, session, path, global_step=None):
        return self.saver.save(session, path, global_step)

    def load(self, session, path):
        self.saver.restore(session, path)

    def add_summaries(self):
        return _Sym("summaries")           # TensorBoard is out of scope; fetch yields None

    def get_predict_feed_dict(self, srcSeqs, tgtSeqs):
        return {self._src_input_data: np.array(srcSeqs, dtype=np.int32),
                self._tgt_input_data: np.array(tgtSeqs, dtype=np.int32)}

    def get_train_feed_dict(self, srcSeqs, tgtSeqs, labels):
        return {self._src_input_data: np.array(srcSeqs, dtype=np.int32),
                self._labels: np.array(labels, dtype=np.float32),
                self._tgt_input_data: np.array(tgtSeqs, dtype=np.int32)}

    def get_source_encoding_feed_dict(self, srcSeqs):
        return {self._src_input_data: np.array(srcSeqs, dtype=np.int32)}

    def get_target_encoding_feed_dict(self, tgtSeqs):
        return {self._tgt_input_data: np.array(tgtSeqs, dtype=np.int32)}


class Session(object):
    """`tf.Session` stand-in: `run(fetches, feed_dict)` dispatches to the C ABI.
    One train step is executed at most once per run() however many of
    train/loss/train_acc are fetched, like a TF graph execution."""

    def __init__(self, model=None):
        self.model = model

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        feed = feed_dict or {}
        model = self.model
        if model is None:
            raise RuntimeError("Session has no model bound")
        names = [f.name for f in fl]
        cache = {}
        if any(n in ("train", "loss", "train_acc") for n in names):
            src, tgt, lab = feed[model._src_input_data], feed[model._tgt_input_data], feed[model._labels]
            if "train" in names:
                cache["loss"], cache["train_acc"] = model.train_step(src, tgt, lab)
            else:
                raise NotImplementedError("fetching loss/train_acc without model.train is not used by the reference CLIs")
            cache["train"] = None
        out = []
        for n in names:
            if n in cache:
                out.append(cache[n])
            elif n == "norm_src_seq_embedding":
                out.append(model.encode_source(feed[model._src_input_data], True))
            elif n == "src_seq_embedding":
                out.append(model.encode_source(feed[model._src_input_data], False))
            elif n == "norm_tgt_seq_embedding":
                out.append(model.encode_target(feed[model._tgt_input_data], True))
            elif n == "tgt_seq_embedding":
                out.append(model.encode_target(feed[model._tgt_input_data], False))
            elif n in ("predicted_tgts_score", "predicted_labels"):
                if "predict" not in cache:
                    cache["predict"] = model.predict(feed[model._src_input_data], feed[model._tgt_input_data])
                out.append(cache["predict"][0 if n == "predicted_tgts_score" else 1])
            elif n == "similarity":
                out.append(model.similarity_matrix(feed[model._src_input_data], feed[model._tgt_input_data]))
            elif n == "learning_rate_decay_op":
                model.handle.decay_learning_rate()
                out.append(model.handle.learning_rate)
            elif n == "learning_rate":
                out.append(model.handle.learning_rate)
            elif n == "global_step":
                out.append(model.handle.global_step)
            elif n == "summaries":
                out.append(None)
            else:
                raise KeyError("cannot fetch %r" % n)
        return out[0] if single else out
