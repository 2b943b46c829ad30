"""Corpus preparation and batching with the reference's file formats
(`data_utils.py:82-259`, `data.py:38-115`): DataSet.tar.gz -> TrainPairs /
EvalPairs / targetIDs, *.Corpus, vocabulary.txt, encoded.FullTargetSpace,
modelConfig.param, and the JSON cache `model_dir/compressed`.

Host-side Python above the C ABI (not accelerated); kept so the command lines
run stand-alone and so that `get_train_batch` defines the kernel inputs exactly
as the reference does (interleaved positive / sampled-negative rows).
"""
import codecs
import glob
import json
import os
import tarfile

import numpy as np

from . import sse_text


def save_model_configs(model_dir, configs):
    """modelConfig.param: one key=value per line (data_utils.py:244-249)."""
    with codecs.open(os.path.join(model_dir, "modelConfig.param"), "w", "utf-8") as f:
        for k, v in configs.items():
            f.write("%s=%s\n" % (k, v))


def load_model_configs(model_dir):
    """All values come back as strings, exactly like the reference (data_utils.py:252-259)."""
    cfg = {}
    for line in codecs.open(os.path.join(model_dir, "modelConfig.param"), "r", "utf-8"):
        if "=" not in line.strip():
            continue
        k, v = line.strip().split("=")
        cfg[k] = v
    return cfg


def extract_data_set(raw_dir, work_dir):
    """Untar DataSet.tar.gz and write the lower-cased source/target corpora used to
    build the vocabulary (data_utils.py:82-112)."""
    if os.path.exists(os.path.join(work_dir, "TrainPairs")) and os.path.exists(os.path.join(work_dir, "vocabulary.txt")):
        return
    tar = os.path.join(raw_dir, "DataSet.tar.gz")
    if not os.path.exists(tar):
        raise FileNotFoundError("Error! No corups file found at: %s" % tar)
    with tarfile.open(tar, "r") as t:
        t.extractall(work_dir)
    with codecs.open(os.path.join(work_dir, "sourceSeq.Corpus"), "w", "utf-8") as src, \
            codecs.open(os.path.join(work_dir, "targetSeq.Corpus"), "w", "utf-8") as tgt:
        for name in ("TrainPairs", "EvalPairs"):
            for line in codecs.open(os.path.join(work_dir, name), "r", "utf-8"):
                info = line.strip().split("\t")
                if len(info) >= 2:
                    src.write(info[0].lower() + "\n")
        for line in codecs.open(os.path.join(work_dir, "targetIDs"), "r", "utf-8"):
            info = line.strip().split("\t")
            if len(info) >= 2:
                tgt.write(info[0].lower() + "\n")


def load_or_build_vocab(work_dir, vocab_size, max_lines=1000000, log=print):
    path = os.path.join(work_dir, "vocabulary.txt")
    if os.path.exists(path):
        return sse_text.SubwordVocab(path)
    counts = sse_text.corpus_token_counts(glob.glob(os.path.join(work_dir, "*.Corpus")), max_lines)
    vocab = sse_text.SubwordVocab.build_to_target_size(vocab_size, counts, 2, 1000)
    vocab.store(path)
    log("New vocabulary constructed: %d subtokens" % vocab.vocab_size)
    return vocab


def read_pairs(path, full_target_space, vocab, max_seq_length, log=None):
    """(source token ids, verified target ids) per usable line (data_utils.py:115-157)."""
    corpus = []
    known = set(full_target_space)
    for line in codecs.open(path, "r", "utf-8"):
        info = line.strip().split("\t")
        if len(info) != 2:
            if log:
                log("bad line in %s: %r" % (path, line))
            continue
        text, ids = info
        verified = [t for t in ids.split("|") if t in known]
        if not verified:
            continue
        corpus.append((sse_text.pad_tokens(vocab.encode(text.lower()), max_seq_length), verified))
    return corpus


def prepare_raw_data(raw_dir, work_dir, vocab_size, max_seq_length, log=print):
    """Same products as data_utils.prepare_raw_data (data_utils.py:160-213)."""
    os.makedirs(work_dir, exist_ok=True)
    extract_data_set(raw_dir, work_dir)
    vocab = load_or_build_vocab(work_dir, vocab_size, log=log)
    full, names = {}, {}
    with codecs.open(os.path.join(work_dir, "encoded.FullTargetSpace"), "w", "utf-8") as out:
        for line in codecs.open(os.path.join(work_dir, "targetIDs"), "r", "utf-8"):
            seq, tid = line.strip().split("\t")
            ids = sse_text.pad_tokens(vocab.encode(seq.lower()), max_seq_length)
            full[tid] = ids
            names[tid] = seq
            out.write(tid + "\t" + seq.strip() + "\t" + ",".join(str(i) for i in ids) + "\n")
    eval_corpus = read_pairs(os.path.join(work_dir, "EvalPairs"), full, vocab, max_seq_length)
    train_corpus = read_pairs(os.path.join(work_dir, "TrainPairs"), full, vocab, max_seq_length)
    return vocab, train_corpus, eval_corpus, full, names


class Data(object):
    """data.Data (data.py:38-115): builds or reloads `model_dir/compressed`; batches
    are a random contiguous window of positives, each followed by one uniformly
    sampled negative (labels 1.0 / 0.0)."""

    def __init__(self, work_dir, rawdata_dir, rawvocabsize, max_seq_length, seed=None, log=print):
        self.rng = np.random.RandomState(seed)
        cache = os.path.join(work_dir, "compressed")
        if os.path.exists(cache):
            with open(cache) as f:
                for k, v in json.load(f).items():
                    setattr(self, k, v)
            self.encoder = load_or_build_vocab(work_dir, rawvocabsize, max_lines=2000000, log=log)
            self.max_seq_length = int(self.max_seq_length)
        else:
            enc, train, evalc, full, names = prepare_raw_data(rawdata_dir, work_dir, rawvocabsize, max_seq_length, log)
            self.encoder = enc
            self.rawTrainPosCorpus = train
            self.rawEvalCorpus = evalc
            self.max_seq_length = max_seq_length
            self.encodedFullTargetSpace = full
            self.tgtIdNameMap = names
            self.fullSetTargetIds = list(full.keys())
            self.rawnegSetLen = len(self.fullSetTargetIds)
            with open(cache, "w") as f:
                json.dump({k: v for k, v in self.__dict__.items() if k not in ("encoder", "rng")}, f)
        self.vocab_size = self.encoder.vocab_size
        log("Vocab size: %d unique words; max allowed sequence length: %d" % (self.vocab_size, self.max_seq_length))

    def target_row(self, tgt_id):
        """Row of a target id in the free target matrix of source_only_cnn / source-encoder-only (builder-defined:
        the position of the id in fullSetTargetIds, i.e. the order of the targetIDs file)."""
        if getattr(self, "_row_of", None) is None:
            self._row_of = {t: i for i, t in enumerate(self.fullSetTargetIds)}
        return self._row_of[tgt_id]

    def _corpus_arrays(self):
        """The padded corpora as int32 matrices (built once): sources [n_pos, T], targets [N, T] in
        fullSetTargetIds order; plus the positives of every source as target rows."""
        if getattr(self, "_arr", None) is None:
            src = np.array([tokens for tokens, _ in self.rawTrainPosCorpus], dtype=np.int32)
            tgt = np.array([self.encodedFullTargetSpace[t] for t in self.fullSetTargetIds], dtype=np.int32)
            self._arr = (src, tgt)
        return self._arr

    def get_train_batch_arrays(self, batch_size, target_rows=False):
        """get_train_batch with the same random draws in the same order (same seed -> same batch), returning
        ndarrays gathered from pre-built corpus matrices instead of lists of token lists: the list -> ndarray
        conversion of the feed dict (sse_model.py:419-421) was 0.5 ms of a 3 ms train step."""
        src_arr, tgt_arr = self._corpus_arrays()
        n = len(self.rawTrainPosCorpus)
        start = self.rng.randint(0, n - batch_size) + batch_size     # data.py:97 (window may be cut at the end)
        stop = min(n, start + batch_size)
        rows = np.empty(2 * (stop - start), np.int64)
        for i in range(start, stop):
            verified = self.rawTrainPosCorpus[i][1]
            pos = verified[self.rng.randint(0, len(verified))]
            positives = set(verified)
            neg = self.fullSetTargetIds[self.rng.randint(0, self.rawnegSetLen)]
            while neg in positives:
                neg = self.fullSetTargetIds[self.rng.randint(0, self.rawnegSetLen)]
            rows[2 * (i - start)] = self.target_row(pos)
            rows[2 * (i - start) + 1] = self.target_row(neg)
        src = src_arr[np.repeat(np.arange(start, stop), 2)]
        tgt = rows.astype(np.int32) if target_rows else tgt_arr[rows]
        labels = np.tile(np.array([1.0, 0.0], np.float32), stop - start)
        return src, tgt, labels

    def corpus_matrices(self):
        """(source corpus [n_pos, T], target corpus [N, T] in fullSetTargetIds order) int32: what sse_corpus_upload
        keeps resident on the device."""
        return self._corpus_arrays()

    def _positives_csr(self):
        if getattr(self, "_csr", None) is None:
            cnt = np.array([len(v) for _, v in self.rawTrainPosCorpus], np.int64)
            off = np.concatenate([[0], np.cumsum(cnt)])
            flat = np.array([self.target_row(t) for _, v in self.rawTrainPosCorpus for t in v], np.int64)
            width = int(cnt.max())
            pad = np.full((len(cnt), width), -1, np.int64)
            for i, (_, v) in enumerate(self.rawTrainPosCorpus):
                pad[i, :len(v)] = [self.target_row(t) for t in v]
            self._csr = (cnt, off, flat, pad)
        return self._csr

    def get_train_batch_rows(self, batch_size, vectorized=False):
        """The batch of get_train_batch as ROW NUMBERS: (source-corpus rows [2b], target rows [2b], labels [2b]),
        pos/neg interleaved, each source row twice (data.py:95-115).  vectorized=False draws exactly the random
        numbers of get_train_batch in the same order (same seed -> same batch).  vectorized=True samples the whole
        window with array operations -- the same distribution (positive uniform among the verified targets, negative
        uniform among the targets that are not positives of the source, by rejection) from a different stream; the
        reference itself draws from the unseeded global numpy.random, so no stream is part of its contract."""
        n = len(self.rawTrainPosCorpus)
        start = self.rng.randint(0, n - batch_size) + batch_size     # data.py:97 (window may be cut at the end)
        stop = min(n, start + batch_size)
        b = stop - start
        rows = np.empty(2 * b, np.int64)
        if vectorized:
            cnt, off, flat, pad = self._positives_csr()
            idx = np.arange(start, stop)
            rows[0::2] = flat[off[idx] + (self.rng.random_sample(b) * cnt[idx]).astype(np.int64)]
            neg = self.rng.randint(0, self.rawnegSetLen, size=b)
            bad = (pad[idx] == neg[:, None]).any(axis=1)
            while bad.any():
                neg[bad] = self.rng.randint(0, self.rawnegSetLen, size=int(bad.sum()))
                bad = (pad[idx] == neg[:, None]).any(axis=1)
            rows[1::2] = neg
        else:
            for i in range(start, stop):
                verified = self.rawTrainPosCorpus[i][1]
                pos = verified[self.rng.randint(0, len(verified))]
                positives = set(verified)
                neg = self.fullSetTargetIds[self.rng.randint(0, self.rawnegSetLen)]
                while neg in positives:
                    neg = self.fullSetTargetIds[self.rng.randint(0, self.rawnegSetLen)]
                rows[2 * (i - start)] = self.target_row(pos)
                rows[2 * (i - start) + 1] = self.target_row(neg)
        src_rows = np.repeat(np.arange(start, stop, dtype=np.int32), 2)
        return src_rows, rows.astype(np.int32), np.tile(np.array([1.0, 0.0], np.float32), b)

    def get_train_batch(self, batch_size, target_rows=False):
        """data.py:95-115.  target_rows=True (source_only_cnn): the target side of each pair is the row of the
        target id in the free target matrix instead of its token sequence."""
        tgt_of = self.target_row if target_rows else self.encodedFullTargetSpace.__getitem__
        n = len(self.rawTrainPosCorpus)
        start = self.rng.randint(0, n - batch_size) + batch_size     # data.py:97 (window may be cut at the end)
        src, tgt, labels = [], [], []
        for tokens, verified in self.rawTrainPosCorpus[start:start + batch_size]:
            pos = verified[self.rng.randint(0, len(verified))]
            positives = set(verified)
            src.append(tokens)
            tgt.append(tgt_of(pos))
            labels.append(1.0)
            neg = self.fullSetTargetIds[self.rng.randint(0, self.rawnegSetLen)]
            while neg in positives:
                neg = self.fullSetTargetIds[self.rng.randint(0, self.rawnegSetLen)]
            src.append(tokens)
            tgt.append(tgt_of(neg))
            labels.append(0.0)
        return src, tgt, labels
