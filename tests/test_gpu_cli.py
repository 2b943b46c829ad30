"""End-to-end drop-in check on the GPU: the sse_train / sse_index / sse_demo
command lines on a seeded stand-in for the (missing) rawdata-classification set,
file formats of the reference, and the evaluator numbers against the oracle's
restatement of Evaluator.eval on identical weights; plus real-data parity on a
slice of rawdata-crosslingual (token ids produced by the reference's own
data_utils, tests/golden/crosslingual_ids.npz)."""
import importlib.util
import io
import os

import numpy as np
import pytest

from oracle import sse_oracle as O
from tests.util import make_pair, model_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _standin(tmp, **kw):
    spec = importlib.util.spec_from_file_location("standin", os.path.join(ROOT, "tools", "make_standin_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    raw = os.path.join(tmp, "rawdata-classification")
    mod.write_tar(mod.generate(**kw), raw)
    return raw


def test_train_index_eval_demo_cli(tmp_path):
    import sse_amd
    from sse_amd import sse_data, sse_demo, sse_evaluator, sse_index, sse_train
    tmp = str(tmp_path)
    raw = _standin(tmp, n_targets=37, n_train=600, n_eval=90, n_vocab=400, seed=1)
    mdir = os.path.join(tmp, "models-classification")
    import logging
    sse_train.main(["--task_type=classification", "--data_dir=" + raw, "--model_dir=" + mdir, "--max_epoc=1",
                    "--steps_per_checkpoint=10", "--batch_size=16", "--network_mode=shared-encoder",
                    "--src_cell_size=128", "--vocab_size=600", "--max_seq_length=24", "--seed=0", "--max_steps=25",
                    "--learning_rate=0.3"])
    logging.getLogger("").handlers.clear()
    for name in ("vocabulary.txt", "modelConfig.param", "compressed", "encoded.FullTargetSpace",
                 "targetEncodingIndex.tsv", "TrainingLog.txt", "checkpoint", "SSE-LSTM.ckpt-BestEver.npz",
                 "SSE-LSTM.ckpt-epoch-0.npz"):
        assert os.path.exists(os.path.join(mdir, name)), name
    log = open(os.path.join(mdir, "TrainingLog.txt")).read()
    assert "train_binary_acc" in log and "top 1/3/10 accuracies" in log

    # the trained model, reloaded the way sse_index / sse_demo do
    cfg = sse_data.load_model_configs(mdir)
    model = sse_amd.SSEModel(cfg)
    model.saver.restore(None, sse_amd.get_checkpoint_state(mdir))
    p = model.get_variables()
    ocfg = dict(cfg, vocab_size=int(cfg["vocab_size"]), embedding_size=int(cfg["embedding_size"]),
                encoding_size=int(cfg["encoding_size"]), src_cell_size=int(cfg["src_cell_size"]),
                tgt_cell_size=int(cfg["tgt_cell_size"]))

    # sse_index CLI rewrites the index from the checkpoint; rows == oracle encodings of the padded targets
    sse_index.main(["--idx_model_dir=" + mdir])
    lines = open(os.path.join(mdir, "targetEncodingIndex.tsv"), encoding="utf-8").readlines()
    ids, sents, enc = O.parse_index_lines(lines)
    assert len(ids) == 37 and enc.shape == (37, int(cfg["encoding_size"]))
    data = sse_data.Data(mdir, raw, 600, 24, log=lambda *a: None)
    tgt_rows = np.array([data.encodedFullTargetSpace[t] for t in ids], np.int32)
    want = O.encode(p, ocfg, "tgt", tgt_rows)
    assert np.abs(enc - want).max() < 1e-4
    assert lines[0].split("\t")[1] == open(os.path.join(mdir, "targetIDs"), encoding="utf-8").readline().split("\t")[0]

    # Evaluator numbers == the oracle's restatement of sse_evaluator.py:95-114 on the same index file
    ev = sse_evaluator.Evaluator(model, data.rawEvalCorpus, os.path.join(mdir, "targetEncodingIndex.tsv"),
                                 sse_amd.Session(model))
    got = ev.eval()
    src_rows = np.array([e[0] for e in data.rawEvalCorpus], np.int32)
    gpu_src = model.encode_source(src_rows)
    assert np.abs(gpu_src - O.encode(p, ocfg, "src", src_rows)).max() < 1e-4
    labels = [[ids.index(t) for t in e[1]] for e in data.rawEvalCorpus]
    # ranking parity is judged on identical inputs (the GPU's source encodings): a 1e-7 difference
    # between two encoders may legitimately swap near-tied neighbours of a barely trained model
    want_acc = O.evaluator_accuracy(gpu_src, enc, labels)
    assert got == pytest.approx(want_acc, abs=1e-12)

    # sse_demo: un-normalised source encoding x index, top-N printed (sse_demo.py:121-134)
    out = io.StringIO()
    f = sse_demo.FLAGS.parse(["--model_dir=" + mdir])
    query = open(os.path.join(mdir, "EvalPairs"), encoding="utf-8").readline().split("\t")[0]
    sse_demo.demo(f, 5, stdin=io.StringIO(query + "\nexit\n"), out=out)
    text = out.getvalue()
    assert "Top 5 Prediction results are:" in text and text.count("top") >= 5
    # (the reference tokenises the line WITH its trailing newline, sse_demo.py:113)
    q_ids = np.array([sse_amd.sse_text.pad_tokens(data.encoder.encode((query + "\n").lower()), 24)], np.int32)
    raw_enc = model.encode_source(q_ids, normalize=False)      # same inputs as the demo: only the ranking is compared
    assert np.abs(raw_enc - O.encode(p, ocfg, "src", q_ids, normalize=False)).max() < 1e-4 * max(1.0, np.abs(raw_enc).max())
    wsc, wids = O.topk(O.scores_f64(raw_enc, enc), 5)
    assert ("top1:  %s , %f" % (ids[wids[0][0]], wsc[0][0])) in text

    # serving shell (webserver.py:124-286): /api/classify ranks the NORMALISED encoding, /api/search the raw one;
    # concurrent requests through the micro-batcher give each caller the single-request answer
    import json
    import threading
    from sse_amd import sse_serving
    app = sse_serving.create_app(model_dir=mdir)

    def get(path, qs):
        out = {}
        body = b"".join(app({"PATH_INFO": path, "QUERY_STRING": qs}, lambda st, hd: out.update(status=st)))
        assert out["status"].startswith("200"), (out, body)
        return json.loads(body)

    from urllib.parse import quote_plus
    w_ids = np.array([sse_amd.sse_text.pad_tokens(data.encoder.encode(query.lower()), 24)], np.int32)
    d = get("/api/classify", "keywords=" + quote_plus(query))
    nsc, nids = O.topk(O.scores_f64(model.encode_source(w_ids, normalize=True), enc), 8)
    assert [r["targetCategoryId"] for r in d["ClassificationResults"]] == [ids[j] for j in nids[0]]
    assert np.allclose([r["confidenceScore"] for r in d["ClassificationResults"]], nsc[0], atol=1e-12)
    d = get("/api/search", "query=%s&nbest=4" % quote_plus(query))
    rsc, rids = O.topk(O.scores_f64(model.encode_source(w_ids, normalize=False), enc), 4)
    assert [r["ListingId"] for r in d["SearchRankingResults"]] == [ids[j] for j in rids[0]]
    assert np.allclose([r["rankingScore"] for r in d["SearchRankingResults"]], rsc[0], atol=1e-12)
    queries = [l.split("\t")[0] for l in open(os.path.join(mdir, "EvalPairs"), encoding="utf-8").readlines()[:16]]
    single = [get("/api/qna", "question=" + quote_plus(q)) for q in queries]
    multi = [None] * len(queries)
    th = [threading.Thread(target=lambda i=i: multi.__setitem__(i, get("/api/qna", "question=" + quote_plus(queries[i]))))
          for i in range(len(queries))]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert multi == single


def test_cnn_mode_train_index_eval_cli(tmp_path):
    """source_only_cnn through the same command lines (BUILDER-DEFINED training, configs[4]): pairs carry
    target-matrix rows, the index file holds the l2-normalised target matrix, the evaluator ranks CNN source
    encodings against it; after a short training run the top-10 accuracy is far above chance (10/37)."""
    import logging
    import sse_amd
    from sse_amd import sse_data, sse_evaluator, sse_train
    tmp = str(tmp_path)
    raw = _standin(tmp, n_targets=37, n_train=800, n_eval=120, n_vocab=400, seed=2)
    mdir = os.path.join(tmp, "models-cnn")
    sse_train.main(["--task_type=classification", "--data_dir=" + raw, "--model_dir=" + mdir, "--max_epoc=10",
                    "--steps_per_checkpoint=20", "--batch_size=32", "--network_mode=source_only_cnn",
                    "--vocab_size=600", "--max_seq_length=16", "--embedding_size=24", "--encoding_size=32", "--seed=0",
                    "--max_steps=250", "--learning_rate=0.1"])
    logging.getLogger("").handlers.clear()
    cfg = sse_data.load_model_configs(mdir)
    model = sse_amd.SSEModel(cfg)
    model.saver.restore(None, sse_amd.get_checkpoint_state(mdir))
    p = model.get_variables()
    lines = open(os.path.join(mdir, "targetEncodingIndex.tsv"), encoding="utf-8").readlines()
    ids, sents, enc = O.parse_index_lines(lines)
    data = sse_data.Data(mdir, raw, 600, 16, log=lambda *a: None)
    assert ids == data.fullSetTargetIds and enc.shape == (37, 32)
    assert np.abs(enc - O.l2_normalize(p["target_embedding/tgt_seq_embedding"])).max() < 1e-6
    ev = sse_evaluator.Evaluator(model, data.rawEvalCorpus, os.path.join(mdir, "targetEncodingIndex.tsv"),
                                 sse_amd.Session(model))
    acc1, acc3, acc10 = ev.eval()
    assert acc10 > 0.6 and acc1 > 0.2, (acc1, acc3, acc10)
    log = open(os.path.join(mdir, "TrainingLog.txt")).read()
    assert "train_binary_acc" in log


def test_cnn_configs4_shape_index_eval_demo_serving(tmp_path):
    """BASELINE configs[4] shape (source_only_cnn, T = 64, S = 512, E = 50, bf16 training) through every consumer of
    the index: sse_train (+ its index build) -> Evaluator.eval -> sse_demo -> /api/classify.  The 512-wide index
    runs the scorer with 64-query LDS blocks; rankings are checked against the float64 oracle on the GPU's own
    encodings, the encodings against the oracle's CNN (sse_model.py:179-214,286; sse_evaluator.py:110-111)."""
    import json
    import logging
    import sse_amd
    from urllib.parse import quote_plus
    from sse_amd import sse_data, sse_demo, sse_evaluator, sse_index, sse_serving, sse_train
    tmp = str(tmp_path)
    raw = _standin(tmp, n_targets=37, n_train=700, n_eval=650, n_vocab=400, seed=3)
    mdir = os.path.join(tmp, "models-cnn512")
    sse_train.main(["--task_type=classification", "--data_dir=" + raw, "--model_dir=" + mdir, "--max_epoc=4",
                    "--steps_per_checkpoint=20", "--batch_size=32", "--network_mode=source_only_cnn", "--cnn_bf16=1",
                    "--vocab_size=600", "--max_seq_length=64", "--embedding_size=50", "--encoding_size=512", "--seed=0",
                    "--max_steps=120", "--learning_rate=0.1"])
    logging.getLogger("").handlers.clear()
    cfg = sse_data.load_model_configs(mdir)
    assert int(cfg["encoding_size"]) == 512 and int(cfg["max_seq_length"]) == 64
    model = sse_amd.SSEModel(cfg)
    model.saver.restore(None, sse_amd.get_checkpoint_state(mdir))
    p = model.get_variables()
    ocfg = {k: (int(v) if k.endswith("_size") or k in ("max_seq_length", "targetSpaceSize") else v) for k, v in cfg.items()}
    sse_index.main(["--idx_model_dir=" + mdir])
    index_path = os.path.join(mdir, "targetEncodingIndex.tsv")
    ids, sents, enc = O.parse_index_lines(open(index_path, encoding="utf-8").readlines())
    assert enc.shape == (37, 512)
    assert np.abs(enc - O.l2_normalize(p["target_embedding/tgt_seq_embedding"])).max() < 1e-6
    data = sse_data.Data(mdir, raw, 600, 64, log=lambda *a: None)
    # Evaluator: 650 eval sources = one full 600-batch + a short one (sse_evaluator.py:104-113), index dimension 512
    ev = sse_evaluator.Evaluator(model, data.rawEvalCorpus, index_path, sse_amd.Session(model))
    got = ev.eval()
    src_rows = np.array([e[0] for e in data.rawEvalCorpus], np.int32)
    gpu_src = model.encode_source(src_rows)
    assert np.abs(gpu_src - O.encode(p, ocfg, "src", src_rows)).max() < 1e-4
    labels = [[ids.index(t) for t in e[1]] for e in data.rawEvalCorpus]
    assert got == pytest.approx(O.evaluator_accuracy(gpu_src, enc, labels), abs=1e-12)
    assert got[2] > 0.5, got
    sc, rk = model.handle.score_topk(gpu_src, 10)               # the index the Evaluator uploaded (float64 rows)
    wsc, wids = O.topk(O.scores_f64(gpu_src, enc), 10)
    assert np.array_equal(rk, wids) and np.abs(sc - wsc).max() < 1e-12
    # sse_demo (un-normalised encoding, sse_demo.py:121-129)
    out = io.StringIO()
    query = open(os.path.join(mdir, "EvalPairs"), encoding="utf-8").readline().split("\t")[0]
    sse_demo.demo(sse_demo.FLAGS.parse(["--model_dir=" + mdir]), 5, stdin=io.StringIO(query + "\nexit\n"), out=out)
    q_ids = np.array([sse_amd.sse_text.pad_tokens(data.encoder.encode((query + "\n").lower()), 64)], np.int32)
    raw_enc = model.encode_source(q_ids, normalize=False)
    dsc, dids = O.topk(O.scores_f64(raw_enc, enc), 5)
    assert ("top1:  %s , %f" % (ids[dids[0][0]], dsc[0][0])) in out.getvalue()
    # /api/classify (normalised encoding, webserver.py:144-151)
    app = sse_serving.create_app(model_dir=mdir)
    st = {}
    body = b"".join(app({"PATH_INFO": "/api/classify", "QUERY_STRING": "keywords=" + quote_plus(query)},
                        lambda status, hd: st.update(status=status)))
    assert st["status"].startswith("200"), (st, body)
    d = json.loads(body)
    w_ids = np.array([sse_amd.sse_text.pad_tokens(data.encoder.encode(query.lower()), 64)], np.int32)
    nsc, nids = O.topk(O.scores_f64(model.encode_source(w_ids, normalize=True), enc), 8)
    assert [r["targetCategoryId"] for r in d["ClassificationResults"]] == [ids[j] for j in nids[0]]
    assert np.allclose([r["confidenceScore"] for r in d["ClassificationResults"]], nsc[0], atol=1e-12)


def test_crosslingual_real_data_slice_matches_oracle():
    """SURVEY 8d C3 on real token ids: encode 4868 targets (index) and 600 queries with
    the dual-encoder H=S=256 T=50 model, score, rank: cosine |d| <= 1e-4, top-10 ids and
    the top-1/3/10 accuracies identical to the oracle."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "crosslingual_ids.npz"))
    V = int(z["vocab_size"])
    params = model_params("dual-encoder", V, 50, 256, 256, 256, 50)
    m, p = make_pair(params, seed=8)
    tgt = m.encode_target(z["tgt_ids"])
    src = m.encode_source(z["src_ids"])
    want_t = O.encode(p, params, "tgt", z["tgt_ids"][:256])
    want_s = O.encode(p, params, "src", z["src_ids"][:256])
    assert np.abs(tgt[:256] - want_t).max() < 1e-4 and np.abs(src[:256] - want_s).max() < 1e-4
    # index hand-off through decimal text, as the reference does (sse_index.py:93-95 -> sse_evaluator.py:87)
    _, _, idx64 = O.parse_index_lines([O.format_index_line("t%d" % i, "x", v) for i, v in enumerate(tgt)])
    m.handle.index_upload(idx64)
    sc, ids = m.handle.score_topk(src, 10)
    wsc, wids = O.topk(O.scores_f64(src, idx64), 10)
    assert np.array_equal(ids, wids) and np.abs(sc - wsc).max() < 1e-12
    labels = [[int(v) for v in row if v >= 0] for row in z["labels"]]
    from sse_amd import sse_evaluator
    for n in (1, 3, 10):
        assert sse_evaluator.topk_tight_accuracy(n, labels, ids) == O.topk_tight_accuracy(n, labels, wids)
