"""Serving shell routes (reference webserver.py:124-291) and the TensorFlow-checkpoint reader, without a GPU:
the ranker is a stub (the GPU path is exercised by tests/test_gpu_cli.py)."""
import io
import json
import os
import threading

import numpy as np
import pytest

import sse_amd
from sse_amd import sse_serving, tf_checkpoint


class _StubRanker(object):
    max_seq_length = 6

    def __init__(self):
        self.calls = []

    def tokens(self, text):
        return [0] * (6 - 1 - len(text.split())) + [5 + len(w) for w in text.lower().split()] + [1]

    def rank(self, rows, nbest, normalize):
        self.calls.append((len(rows), nbest, normalize))
        return [[(1.0 - 0.1 * j, "id%d" % (sum(r) + j), "name %d" % j) for j in range(nbest)] for r in rows]


def _get(app, path, query=""):
    out = {}

    def start_response(status, headers):
        out["status"], out["headers"] = status, dict(headers)

    body = b"".join(app({"PATH_INFO": path, "QUERY_STRING": query, "REQUEST_METHOD": "GET"}, start_response))
    return out["status"], out["headers"], body


def test_routes_keys_defaults_and_normalisation_flags():
    r = _StubRanker()
    app = sse_serving.create_app(ranker=r, batch=False)
    st, hd, body = _get(app, "/api/classify", "keywords=Hello+Kitty+sunglasses")
    assert st.startswith("200") and hd["Content-Type"] == "application/json"
    d = json.loads(body)
    assert d["ReqeustKeywords"] == "Hello Kitty sunglasses"                 # (sic) webserver.py:161
    assert len(d["ClassificationResults"]) == 8                             # default nbest 8, webserver.py:131
    assert set(d["ClassificationResults"][0]) == {"targetCategoryId", "targetCategoryName", "confidenceScore"}
    assert r.calls[-1] == (1, 8, True)                                      # normalised fetch, webserver.py:146
    for path, arg, n, qk, rk, keys in [("/api/search", "query", 10, "SearchQuery", "SearchRankingResults",
                                        {"ListingId", "ListingTitle", "rankingScore"}),
                                       ("/api/qna", "question", 5, "Question", "Answers",
                                        {"answerDocId", "answerContent", "confidenceScore"}),
                                       ("/api/crosslingual", "query", 10, "CrossLingualQuery", "SearchResults",
                                        {"documentId", "documentTitle", "confScore"})]:
        st, _, body = _get(app, path, "%s=red+nike+shoes" % arg)
        d = json.loads(body)
        assert st.startswith("200") and d[qk] == "red nike shoes" and len(d[rk]) == n and set(d[rk][0]) == keys
        assert r.calls[-1] == (1, n, False)                                 # raw fetch, webserver.py:184,225,268
    _, _, body = _get(app, "/api/search", "query=x&nbest=3")
    assert len(json.loads(body)["SearchRankingResults"]) == 3
    _, _, body = _get(app, "/api/search", "query=x&?nbest=3")               # the documented '&?nbest=' is not 'nbest'
    assert len(json.loads(body)["SearchRankingResults"]) == 10
    st, _, body = _get(app, "/")
    assert st.startswith("200") and b"/api/crosslingual?query=" in body
    assert _get(app, "/nope")[0].startswith("404") and _get(app, "/api/qna", "q=1")[0].startswith("400")


def test_micro_batcher_gathers_concurrent_requests():
    r = _StubRanker()
    gate = threading.Event()
    orig = r.rank

    def slow_rank(rows, nbest, normalize):
        gate.wait(2.0)                                  # hold the first call so that the others queue up behind it
        return orig(rows, nbest, normalize)

    r.rank = slow_rank
    app = sse_serving.create_app(ranker=r, batch=True)
    results = [None] * 12

    def worker(i):
        results[i] = json.loads(_get(app, "/api/search", "query=%s" % "+".join(["w"] * (1 + i % 4)))[2])

    th = [threading.Thread(target=worker, args=(i,)) for i in range(12)]
    for t in th:
        t.start()
    import time
    time.sleep(0.2)
    gate.set()
    for t in th:
        t.join(5)
    assert all(res is not None and len(res["SearchRankingResults"]) == 10 for res in results)
    assert app.batcher.requests == 12 and app.batcher.batches < 12          # at least one call served several
    assert max(c[0] for c in r.calls) > 1
    # every request got the answer of ITS tokens
    for i, res in enumerate(results):
        want = sum(r.tokens(" ".join(["w"] * (1 + i % 4))))
        assert res["SearchRankingResults"][0]["ListingId"] == "id%d" % want


def test_tf_bundle_roundtrip_and_npz_conversion(tmp_path):
    rng = np.random.RandomState(0)
    var = {"word_embedding": rng.randn(37, 5).astype(np.float32),
           "word_embedding/Adagrad": np.full((37, 5), 0.1, np.float32),
           "source_encoder/rnn/basic_lstm_cell/kernel": rng.randn(9, 16).astype(np.float32),
           "source_encoder/rnn/basic_lstm_cell/bias": np.zeros(16, np.float32),
           "source_encoder/src_M": rng.randn(4, 3).astype(np.float32),
           "learning_rate": np.float32(0.45), "global_step": np.int32(1234)}
    for i in range(20):                                    # enough keys for several table blocks + shared prefixes
        var["target_encoder/extra_%02d" % i] = rng.randn(3, 2).astype(np.float32)
    prefix = str(tmp_path / "SSE-LSTM.ckpt-1234")
    tf_checkpoint.write_bundle(prefix, {k: np.asarray(v) for k, v in var.items()})
    assert tf_checkpoint.is_tf_checkpoint(prefix)
    got = tf_checkpoint.read_bundle(prefix)
    assert set(got) == set(var)
    for k, v in var.items():
        assert np.array_equal(got[k], np.asarray(v)) and got[k].shape == np.asarray(v).shape, k
    path, names = tf_checkpoint.convert(prefix)
    z = np.load(path)
    assert float(z["learning_rate"]) == pytest.approx(0.45) and int(z["global_step"]) == 1234
    assert np.array_equal(z["word_embedding/Adagrad"], var["word_embedding/Adagrad"])
    # get_checkpoint_state: finds a TF checkpoint, refuses a dangling pointer instead of reporting "no checkpoint"
    (tmp_path / "checkpoint").write_text('model_checkpoint_path: "SSE-LSTM.ckpt-1234"\nall_model_checkpoint_paths: "x"\n')
    os.remove(path)
    assert sse_amd.get_checkpoint_state(str(tmp_path)) == prefix
    os.remove(prefix + ".index")
    with pytest.raises(FileNotFoundError):
        sse_amd.get_checkpoint_state(str(tmp_path))
    with pytest.raises(ValueError):
        open(prefix + ".index", "wb").write(b"\x00" * 64)
        tf_checkpoint.read_bundle(prefix)


def test_micro_batcher_survives_a_broken_wakeup_and_never_hangs():
    """an exception outside ranker.rank must not kill the worker thread (every later request would block
    forever), and a request must not wait without bound."""
    r = _StubRanker()
    b = sse_serving.MicroBatcher(r, max_batch=4, max_wait_s=0.001)
    with pytest.raises(Exception):
        b.submit([1, 2, 3], "not-a-number", True)               # int(nbest) fails while the wake-up is grouped
    assert b._t.is_alive()
    assert b.submit(r.tokens("a b"), 3, True)[0][1].startswith("id")      # the worker still serves
    # a wedged ranker: the request times out instead of hanging
    gate = threading.Event()
    r.rank = lambda rows, nbest, normalize: gate.wait(5.0) or []
    with pytest.raises(TimeoutError):
        b.submit(r.tokens("a"), 3, True, timeout_s=0.6)
    gate.set()


def test_crc32c_known_answers_and_library_routine_matches_python():
    """RFC 3720 B.4 vectors pin the checksum the checkpoint reader verifies; the C routine of libsse_hip.so and the
    pure-Python table loop agree, including across the slicing-by-8 tail."""
    assert tf_checkpoint._crc32c(b"123456789") == 0xE3069283
    assert tf_checkpoint._crc32c(bytes(32)) == 0x8A9136AA
    assert tf_checkpoint._crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tf_checkpoint._crc32c(bytes(range(32))) == 0x46DD794E
    lib = sse_amd.load_library()
    rng = np.random.RandomState(0)

    def py_crc(data):
        crc = 0xFFFFFFFF
        for byte in data:
            crc ^= byte
            for _ in range(8):
                crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
        return crc ^ 0xFFFFFFFF

    for n in (0, 1, 7, 8, 9, 63, 64, 65, 1000):
        data = rng.bytes(n)
        assert int(lib.sse_crc32c(data, n, 0)) == py_crc(data), n
    data = rng.bytes(300)                                        # chained calls == one call
    assert int(lib.sse_crc32c(data[100:], 200, lib.sse_crc32c(data[:100], 100, 0))) == py_crc(data)


def test_hand_assembled_two_shard_bundle_fixture():
    """tests/golden/tf_bundle: written byte by byte by tests/golden/make_tf_bundle_fixture.py (its own varint / CRC / block
    code) -- prefix-compressed keys, three data blocks, two data shards, proto3 zero-field omission."""
    prefix = os.path.join(os.path.dirname(__file__), "golden", "tf_bundle", "model.ckpt-7")
    got = tf_checkpoint.read_bundle(prefix)
    z = np.load(os.path.join(os.path.dirname(prefix), "expected.npz"))
    assert sorted(got) == sorted(k.replace("|", "/") for k in z.files)
    for k in z.files:
        a = got[k.replace("|", "/")]
        assert a.dtype == z[k].dtype and a.shape == z[k].shape and np.array_equal(a, z[k]), k
    arrays = tf_checkpoint.to_npz_arrays(got)
    assert int(arrays["global_step"]) == 7 and float(arrays["learning_rate"]) == pytest.approx(0.81)


def test_corrupt_or_truncated_checkpoints_are_rejected(tmp_path):
    import shutil
    src = os.path.join(os.path.dirname(__file__), "golden", "tf_bundle")
    for name in os.listdir(src):
        shutil.copy(os.path.join(src, name), str(tmp_path))
    prefix = str(tmp_path / "model.ckpt-7")
    assert len(tf_checkpoint.read_bundle(prefix)) == 10
    shard = prefix + ".data-00001-of-00002"
    blob = bytearray(open(shard, "rb").read())
    blob[100] ^= 0x01                                            # one flipped bit in a tensor
    open(shard, "wb").write(bytes(blob))
    with pytest.raises(ValueError, match="CRC-32C"):
        tf_checkpoint.read_bundle(prefix)
    blob[100] ^= 0x01
    open(shard, "wb").write(bytes(blob[:-40]))                   # truncated data shard
    with pytest.raises(ValueError, match="truncated|outside data shard"):
        tf_checkpoint.read_bundle(prefix)
    open(shard, "wb").write(bytes(blob))
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[20] ^= 0x40                                              # a flipped bit inside the first table block
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="CRC-32C"):
        tf_checkpoint.read_bundle(prefix)
