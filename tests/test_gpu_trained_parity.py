"""Parity on TRAINED weights, and the reference's own acceptance number.

Every other parity test runs on random-init weights.  The reference's only quality gate is the line
"task specific evaluation: top 1/3/10 accuracies" that sse_train prints per epoch (sse_train.py:223-229), on weights
that hundreds of Adagrad steps have moved: gates saturate, projections grow, scores crowd together.  Here the HIP train
step (sse_train_step through the C ABI) and the CPU oracle (O.train_step: KAT-pinned cell / Adagrad / clip,
tests/test_oracle_tf_kat.py) run the SAME recipe from the SAME initial weights on the SAME batches:

  phase A  free-running: the two loss trajectories, the final variables, and the final top-1/3/10 accuracies (device
           model evaluated on the device, oracle model by the oracle).  Where the recipe's own dynamics are chaotic (below),
           a second device model is re-synchronised to the oracle's variables + Adagrad slots every 5 steps and ITS loss
           trajectory carries the per-step bar;
  phase B  identical trained weights (the oracle's, loaded into the device model): index build + query encode + ranking
           against the oracle -- encodings within the north-star 1e-3 (observed ~1e-6), ids and float64 scores exact on
           identical encodings, top-1/3/10 equal, top-1 ids on the device's own encodings equal wherever the oracle's top-2
           margin exceeds the encoding tolerance -- and ONE more train step at the exact-path tolerance.

Recipes:
  * `standin`: the seeded stand-in for rawdata-classification (tools/make_standin_dataset.py; 37 classes), dual-encoder
    with the reference's default sizes (sse_train.py:60-74: E = 50, H = 96, S = 64, batch 32 -> 64 pair rows), T = 24,
    --learning_rate=0.005.  At the makefile's lr = 0.9 the first Adagrad steps (lr * g / sqrt(0.1 + g^2) ~ 0.9 per weight)
    throw the model into the all-cosines-zero plateau (loss ln 2) where it sits for thousands of steps -- the oracle shows
    the same -- so a run that actually LEARNS within a test's time uses the smaller rate: top-1 rises from chance (1/37) to
    > 0.4.  This is the "trained" case: large accuracy, real margins.
  * `crosslingual`: makefile:42 verbatim (shared-encoder, E = 40, S = 50, T = 50, H = 96, batch 32, lr = 0.9) on the real
    token rows of tests/golden/crosslingual_full_ids.npz, 240 steps: the plateau regime -- clipping engaged on most
    steps, every score of a query within 1e-5 of every other: the near-tie torture case for the ranking path.  The first
    ~40 steps of this recipe are CHAOTIC in the literal sense: the CPU oracle run twice on the same batches with the 64 pair
    rows of each batch permuted (a different fp32 summation order, nothing else) drifts apart to 6 % in the loss at step 28
    and re-converges to 1e-4 once the plateau is reached (measured; profiles/r05_notes.txt).  A free-running comparison
    therefore cannot hold 1e-3 per step there, whatever the kernels do; the per-step bar is carried by the re-synchronised
    model (observed 1e-5), the free-running pair is held to the end state."""
import importlib.util
import os
import time

import numpy as np
import pytest

from oracle import sse_oracle as O
from tests.util import LOSS_REL_EXACT

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(os.path.dirname(__file__), "golden")


def _batch(rng, src_ids, tgt_ids, positives, bs):
    """Data.get_train_batch (data.py:95-115): a random contiguous window of positives (cut at the end of the corpus),
    per source one of its verified targets, then one uniformly drawn target outside its positive set; rows interleaved
    pos, neg; labels 1, 0."""
    n = len(src_ids)
    start = rng.randint(0, n - bs) + bs
    rows = np.arange(start, min(n, start + bs))
    s, t, z = [], [], []
    for r in rows:
        pos = positives[r]
        s += [src_ids[r], src_ids[r]]
        t.append(tgt_ids[pos[rng.randint(len(pos))]])
        neg = rng.randint(len(tgt_ids))
        while neg in pos:
            neg = rng.randint(len(tgt_ids))
        t.append(tgt_ids[neg])
        z += [1.0, 0.0]
    return np.array(s, np.int32), np.array(t, np.int32), np.array(z, np.float32)


def _standin_recipe(tmp):
    from sse_amd import sse_data
    spec = importlib.util.spec_from_file_location("standin", os.path.join(ROOT, "tools", "make_standin_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    raw = os.path.join(tmp, "rawdata-classification")
    mod.write_tar(mod.generate(n_targets=37, n_train=2000, n_eval=300, n_vocab=1000, seed=1), raw)
    mdir = os.path.join(tmp, "models")
    os.makedirs(mdir)
    T = 24
    data = sse_data.Data(mdir, raw, 2000, T, seed=0, log=lambda *a: None)
    src, tgt = data.corpus_matrices()
    row_of = {t: i for i, t in enumerate(data.fullSetTargetIds)}
    positives = [[row_of[t] for t in v] for _, v in data.rawTrainPosCorpus]
    ev_src = np.array([tok for tok, _ in data.rawEvalCorpus], np.int32)
    ev_labels = [[row_of[t] for t in v] for _, v in data.rawEvalCorpus]
    cfg = dict(forward_only=False, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=data.vocab_size,
               embedding_size=50, encoding_size=64, src_cell_size=96, tgt_cell_size=96, learning_rate=0.005,
               learning_rate_decay_factor=0.99, targetSpaceSize=len(tgt))
    return dict(cfg=cfg, lr=0.005, steps=1200, src=src, tgt=tgt, positives=positives, ev_src=ev_src, ev_labels=ev_labels,
                learns=True, chaotic=False)


def _crosslingual_recipe(tmp):
    z = np.load(os.path.join(G, "crosslingual_full_ids.npz"))
    src, tgt = z["src_ids"].astype(np.int32), z["tgt_ids"].astype(np.int32)
    positives = [[int(v) for v in row if v >= 0] for row in z["labels"]]
    cfg = dict(forward_only=False, network_mode="shared-encoder", predict_nbest=10, max_seq_length=50, vocab_size=int(z["vocab_size"]),
               embedding_size=40, encoding_size=50, src_cell_size=96, tgt_cell_size=96, learning_rate=0.9,
               learning_rate_decay_factor=0.99, targetSpaceSize=len(tgt))
    pick = np.linspace(0, len(src) - 1, 2000).astype(np.int64)          # Train == Eval in this data set: a 2,000-query sample
    return dict(cfg=cfg, lr=0.9, steps=240, src=src, tgt=tgt, positives=positives, ev_src=src[pick],
                ev_labels=[positives[i] for i in pick], learns=False, chaotic=True)


def _oracle_rank(se, te, k=10, batch=600):
    sc, ids = [], []
    t64 = te.astype(np.float64)
    for b0 in range(0, len(se), batch):                                  # Evaluator-style batches (sse_evaluator.py:104-112)
        s, i = O.topk_fast(O.scores_f64(se[b0:b0 + batch], t64), k)
        sc.append(s)
        ids.append(i)
    return np.concatenate(sc), np.concatenate(ids)


@pytest.mark.parametrize("recipe", ["standin", "crosslingual"])
def test_training_trajectory_and_trained_weights_parity(recipe, tmp_path, capsys):
    import sse_amd
    r = (_standin_recipe if recipe == "standin" else _crosslingual_recipe)(str(tmp_path))
    cfg, lr, steps = r["cfg"], r["lr"], r["steps"]
    p = O.init_params(cfg, seed=0)
    m = sse_amd.SSEModel(cfg)
    m.set_variables(p)
    m.handle.learning_rate = lr
    st = O.new_optimizer_state(p)
    rng = np.random.RandomState(0)
    got, want, got_acc, want_acc, got_rs = [], [], [], [], []
    t_dev = t_cpu = 0.0
    m_rs, rs_drift = None, 0.0
    if r["chaotic"]:
        m_rs = sse_amd.SSEModel(cfg)
        m_rs.handle.learning_rate = lr
    for step in range(steps):
        s, t, z = _batch(rng, r["src"], r["tgt"], r["positives"], 32)
        if m_rs is not None:
            if step % 5 == 0:                                            # re-synchronise: the oracle's variables and Adagrad slots
                if step:
                    v = m_rs.get_variables()
                    rs_drift = max(rs_drift, max(float(np.abs(v[k].reshape(w.shape) - w).max()) for k, w in p.items()))
                m_rs.set_variables(p)
                for k in p:
                    m_rs.handle.set_variable(k + "/Adagrad", st[k])
            got_rs.append(m_rs.train_step(s, t, z)[0])
        t0 = time.perf_counter()
        wl, wa = O.train_step(p, st, cfg, s, t, z, lr)
        t1 = time.perf_counter()
        gl, ga = m.train_step(s, t, z)
        t_dev += time.perf_counter() - t1
        t_cpu += t1 - t0
        want.append(float(wl))
        want_acc.append(float(wa))
        got.append(gl)
        got_acc.append(ga)
    got, want = np.array(got), np.array(want)
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
    dev_vars = m.get_variables(with_slots=True)
    dv = {k: float(np.abs(dev_vars[k].reshape(w.shape) - w).max()) for k, w in p.items()}
    ds = {k: float(np.abs(dev_vars[k + "/Adagrad"].reshape(w.shape) - st[k]).max() / max(1.0, float(np.abs(st[k]).max()))) for k, w in p.items()}

    # ---- phase A: the free-running device model, evaluated on the device
    tgt_dev, src_dev = m.encode_target(r["tgt"]), m.encode_source(r["ev_src"])
    m.handle.index_upload(tgt_dev.astype(np.float64))
    _, ids_dev = m.handle.score_topk(src_dev, 10)
    acc_dev = [O.topk_tight_accuracy(n, r["ev_labels"], ids_dev) for n in (1, 3, 10)]
    # the oracle-trained model, evaluated by the oracle
    te, se = O.encode(p, cfg, "tgt", r["tgt"]), O.encode(p, cfg, "src", r["ev_src"])
    wsc, wids = _oracle_rank(se, te)
    acc_cpu = [O.topk_tight_accuracy(n, r["ev_labels"], wids) for n in (1, 3, 10)]

    # ---- phase B: identical TRAINED weights
    m.set_variables(p)
    for k in p:                                                          # and the oracle's Adagrad slots: the next step must agree too
        m.handle.set_variable(k + "/Adagrad", st[k])
    tgt_b, src_b = m.encode_target(r["tgt"]), m.encode_source(r["ev_src"])
    enc_err = max(float(np.abs(tgt_b - te).max()), float(np.abs(src_b - se).max()))
    m.handle.index_upload(te.astype(np.float64))                          # identical encodings: ids / scores exact by construction
    sc_x, ids_x = m.handle.score_topk(se, 10)
    m.handle.index_upload(tgt_b.astype(np.float64))                       # the device's own encodings
    sc_o, ids_o = m.handle.score_topk(src_b, 10)
    margin = wsc[:, 0] - wsc[:, 1]
    clear = margin > max(1e-5, 20 * enc_err)
    with capsys.disabled():
        print("\n[%s] %d steps, %d pair rows/step: device %.1f ms/step, oracle %.1f ms/step" % (recipe, steps, 64, t_dev / steps * 1e3, t_cpu / steps * 1e3))
        print("[%s] loss first/last device %.5f / %.5f, oracle %.5f / %.5f; max rel loss diff: steps 0-199 %.2e, all %.2e; "
              "train_acc (mean of the last 100 steps) device %.4f oracle %.4f"
              % (recipe, got[0], got[-1], want[0], want[-1], rel[:200].max(), rel.max(), np.mean(got_acc[-100:]), np.mean(want_acc[-100:])))
        print("[%s] free-running final weights: max |device - oracle| per variable %s; Adagrad slots (relative to the slot's max) %s"
              % (recipe, {k.split("/")[-1] if "/" in k else k: "%.1e" % v for k, v in dv.items()}, "%.1e" % max(ds.values())))
        print("[%s] top 1/3/10 accuracies: device-trained model on the device %s | oracle-trained model by the oracle %s"
              % (recipe, ["%.4f" % a for a in acc_dev], ["%.4f" % a for a in acc_cpu]))
        print("[%s] trained weights, identical on both sides: max |encoding diff| %.2e; top-1 ids equal on the device's own encodings: "
              "%d of %d (queries whose oracle top-2 margin exceeds %.1e: %d, all equal: %s); median top-2 margin %.2e"
              % (recipe, enc_err, int(np.sum(ids_o[:, 0] == wids[:, 0])), len(wids), max(1e-5, 20 * enc_err), int(clear.sum()),
                 bool(np.array_equal(ids_o[clear, 0], wids[clear, 0])), float(np.median(margin))))

    # phase A assertions.  Two fp32 implementations with different summation orders drift apart step by step; the bar
    # (VERDICT r04): the loss trajectory within 1e-3 relative over the first 200 steps, and the whole run within 1e-2.
    if r["chaotic"]:
        rel_rs = np.abs(np.array(got_rs) - want) / np.maximum(np.abs(want), 1e-3)
        with capsys.disabled():
            print("[%s] chaotic recipe: device model re-synchronised to the oracle every 5 steps: max rel loss diff %.2e, max variable drift "
                  "within 5 steps %.2e; free-running max rel loss diff per 20-step window %s"
                  % (recipe, rel_rs.max(), rs_drift, ["%.1e" % rel[i:i + 20].max() for i in range(0, steps, 20)]))
        assert rel_rs.max() < 1e-3, rel_rs.max()
        assert rs_drift < 2e-3, rs_drift
        assert rel[-20:].max() < 2e-2, rel[-20:].max()                  # the two free runs end on the same plateau
    else:
        assert rel[:200].max() < 1e-3, rel[:200].max()
        assert rel.max() < 1e-2, rel.max()
        for k, v in dv.items():
            assert v < 2e-2, (k, v)
    for a, b in zip(acc_dev, acc_cpu):
        assert abs(a - b) <= 0.03, (acc_dev, acc_cpu)
    if r["learns"]:
        assert acc_cpu[0] > 0.4 and acc_cpu[2] > 0.9, acc_cpu               # chance: 1/37, 10/37
        assert acc_dev[0] > 0.4 and acc_dev[2] > 0.9, acc_dev
        assert np.mean(want_acc[-100:]) > 0.5 and np.mean(got_acc[-100:]) > 0.5

    # phase B assertions
    assert enc_err < 1e-3                                                    # north_star tolerance
    assert enc_err < 5e-5                                                    # what fp32 actually gives on trained weights
    assert np.array_equal(ids_x, wids) and np.abs(sc_x - wsc).max() < 1e-12
    for n in (1, 3, 10):
        assert O.topk_tight_accuracy(n, r["ev_labels"], ids_x) == O.topk_tight_accuracy(n, r["ev_labels"], wids)
    assert np.abs(sc_o[:, 0] - wsc[:, 0]).max() < 1e-3
    assert np.array_equal(ids_o[clear, 0], wids[clear, 0])
    if r["learns"]:
        assert clear.mean() > 0.9
        for n, a in zip((1, 3, 10), acc_cpu):
            assert abs(O.topk_tight_accuracy(n, r["ev_labels"], ids_o) - a) <= 1.0 / len(wids) + 1e-12
    # one more train step on the trained weights + trained Adagrad slots, exact-path tolerance
    s, t, z = _batch(rng, r["src"], r["tgt"], r["positives"], 32)
    wl, wa = O.train_step(p, st, cfg, s, t, z, lr)
    gl, ga = m.train_step(s, t, z)
    assert gl == pytest.approx(float(wl), rel=10 * LOSS_REL_EXACT, abs=1e-6)
    assert ga == pytest.approx(float(wa), abs=1e-6)
    v = m.get_variables(with_slots=True)
    for name, w in p.items():
        assert np.abs(v[name].reshape(w.shape) - w).max() < 2e-4, name
        sl = st[name]
        assert np.abs(v[name + "/Adagrad"].reshape(w.shape) - sl).max() < 2e-4 * max(1.0, float(np.abs(sl).max())), name


QNA_EPOCHS = int(os.environ.get("SSE_QNA_EPOCHS", "100"))
QNA_LR = float(os.environ.get("SSE_QNA_LR", "0.01"))


def test_qna_recipe_learns_on_real_data_and_the_trained_model_matches_the_oracle(capsys):
    """a LEARNED model on REAL data inside the GPU suite.  makefile:17 (rawdata-qna, dual-encoder, the
    defaults of sse_train.py:60-74: E = 50, H = 96, S = 64, batch 32, T = 1000, vocabulary 8000) on the token rows the
    reference's own prepare_raw_data produced (tests/golden/qna_full_ids.npz), trained on the DEVICE with the reference's
    loop (sse_train.py:166-229: windows of 10 steps, learning-rate decay after 5 windows without improvement) through
    sse_train_step_rows.  --learning_rate 0.01: at the makefile's 0.9 this recipe sits on the all-cosines-zero plateau for
    hundreds of epochs (device and oracle alike, profiles/r05r_recipe_qna_lr0.9_200ep.txt), 0.02 .. 0.1 do not leave it within 40
    epochs either; 0.01 reaches a pooled top-1 of 0.44 in 120 epochs, 0.005 0.30 (profiles/r06_recipe_qna_*.txt).
    Then the CPU oracle on the TRAINED weights -- real token statistics, targets of up to 999 tokens, gates that training
    has moved -- against the device: encodings, ranking, the reference's acceptance numbers (sse_train.py:223-229)."""
    import sse_amd
    z = np.load(os.path.join(G, "qna_full_ids.npz"))
    src, tgt = z["src_ids"].astype(np.int32), z["tgt_ids"].astype(np.int32)
    positives = [[int(v) for v in row if v >= 0] for row in z["labels"]]
    V, T = int(z["vocab_size"]), src.shape[1]
    assert T == 1000 and len(tgt) == 93 and len(src) == 602
    cfg = dict(forward_only=False, network_mode="dual-encoder", predict_nbest=10, max_seq_length=T, vocab_size=V,
               embedding_size=50, encoding_size=64, src_cell_size=96, tgt_cell_size=96, learning_rate=QNA_LR,
               learning_rate_decay_factor=0.99, targetSpaceSize=len(tgt))
    m = sse_amd.SSEModel(cfg)
    m.init_variables(seed=0)
    h = m.handle
    h.learning_rate = QNA_LR
    h.corpus_upload(0, src)
    h.corpus_upload(1, tgt)
    rng = np.random.RandomState(0)
    batch, spc = 32, 10
    # diagnostic (printed, not asserted): are the two encoders of a step running side by side?  Inside the whole suite this test has
    # been seen at 52 ms / step against 26 ms alone (profiles/r06_notes.txt); a scratch model, 15 steps each way
    probe = sse_amd.SSEModel(cfg)
    probe.init_variables(seed=1)
    probe.handle.corpus_upload(0, src)
    probe.handle.corpus_upload(1, tgt)
    prow = np.repeat(np.arange(32, dtype=np.int32), 2)
    ptgt = (np.arange(64, dtype=np.int32) * 7) % len(tgt)
    pz = np.tile(np.array([1.0, 0.0], np.float32), 32)
    probe_ms = {}
    for serial in (0, 1, 0):
        probe.handle.set_option("train_serial", serial)
        probe.handle.train_step_rows(prow, ptgt, pz)
        t0 = time.perf_counter()
        for _ in range(15):
            probe.handle.train_step_rows(prow, ptgt, pz)
        probe_ms[serial] = (time.perf_counter() - t0) / 15 * 1e3
    probe.handle.close()
    steps = QNA_EPOCHS * (len(src) // batch)
    previous, window_acc, first_window, last_window = [], 0.0, None, None
    t0 = time.perf_counter()
    for step in range(1, steps + 1):
        n = len(src)
        start = rng.randint(0, n - batch) + batch                         # Data.get_train_batch (data.py:95-115) as row numbers
        rows = np.arange(start, min(n, start + batch))
        trow = np.empty(2 * len(rows), np.int32)
        for i, r in enumerate(rows):
            pos = positives[r]
            trow[2 * i] = pos[rng.randint(len(pos))]
            neg = rng.randint(len(tgt))
            while neg in pos:
                neg = rng.randint(len(tgt))
            trow[2 * i + 1] = neg
        _, acc = h.train_step_rows(np.repeat(rows.astype(np.int32), 2), trow, np.tile(np.array([1.0, 0.0], np.float32), len(rows)))
        window_acc += acc / spc
        if step % spc == 0:                                               # sse_train.py:196-203
            if len(previous) > 6 and window_acc < min(previous[-5:]):
                h.decay_learning_rate()
            previous.append(window_acc)
            first_window = window_acc if first_window is None else first_window
            last_window = window_acc
            window_acc = 0.0
    t_train = time.perf_counter() - t0

    p = m.get_variables()
    t0 = time.perf_counter()
    te_o, se_o = O.encode(p, cfg, "tgt", tgt), O.encode(p, cfg, "src", src)
    wsc, wids = O.topk_fast(O.scores_f64(se_o, te_o.astype(np.float64)), 10)
    t_oracle = time.perf_counter() - t0
    te_d, se_d = m.encode_target(tgt), m.encode_source(src)
    enc_err = max(float(np.abs(te_d - te_o).max()), float(np.abs(se_d - se_o).max()))
    h.index_upload(te_o.astype(np.float64))                               # identical encodings: ids / scores exact by construction
    sc_x, ids_x = h.score_topk(se_o, 10)
    h.index_upload(te_d.astype(np.float64))                               # the device's own encodings of the trained model
    sc_d, ids_d = h.score_topk(se_d, 10)
    acc_o = [O.topk_tight_accuracy(n, positives, wids) for n in (1, 3, 10)]
    acc_d = [O.topk_tight_accuracy(n, positives, ids_d) for n in (1, 3, 10)]
    margin = wsc[:, 0] - wsc[:, 1]
    clear = margin > max(1e-5, 20 * enc_err)
    K = [k for k in p if k.endswith("/kernel")]
    with capsys.disabled():
        print("\n[qna] makefile:17 shapes, T = %d, lr %.4g, %d epochs = %d steps on the device in %.1f s (%.1f ms/step incl. batch sampling); "
              "train_binary_acc first / last window %.3f / %.3f; learning rate at the end %.5f; probe: encoders side by side %.1f ms/step, "
              "option train_serial %.1f ms/step"
              % (T, QNA_LR, QNA_EPOCHS, steps, t_train, t_train / steps * 1e3, first_window, last_window, h.learning_rate, probe_ms[0], probe_ms[1]))
        print("[qna] oracle on the TRAINED weights (%.1f s of CPU): max |encoding diff| %.2e; top 1/3/10 oracle %s device %s; top-1 ids equal "
              "%d of %d (oracle top-2 margin > %.1e: %d queries, all equal there: %s); median top-2 margin %.2e; max |LSTM kernel| %.3f, max |projection| %.3f"
              % (t_oracle, enc_err, ["%.4f" % a for a in acc_o], ["%.4f" % a for a in acc_d], int(np.sum(ids_d[:, 0] == wids[:, 0])), len(wids),
                 max(1e-5, 20 * enc_err), int(clear.sum()), bool(np.array_equal(ids_d[clear, 0], wids[clear, 0])), float(np.median(margin)),
                 max(float(np.abs(p[k]).max()) for k in K), max(float(np.abs(p[k]).max()) for k in p if k.endswith("_M"))))
    assert acc_o[0] > 0.2 and acc_d[0] > 0.2, (acc_o, acc_d)              # chance is 1/93: the model has learned
    assert acc_o[2] > 0.5
    assert last_window > first_window + 0.1                               # and train_binary_acc moved with it
    assert enc_err < 1e-3                                                 # north_star tolerance
    assert enc_err < 5e-5                                                 # what fp32 gives
    assert np.array_equal(ids_x, wids) and np.abs(sc_x - wsc).max() < 1e-12
    assert np.array_equal(ids_d[clear, 0], wids[clear, 0])
    assert clear.mean() > 0.9
    assert np.abs(sc_d[:, 0] - wsc[:, 0]).max() < 1e-3
    for a, b in zip(acc_d, acc_o):
        assert abs(a - b) <= 2.0 / len(wids) + 1e-12, (acc_d, acc_o)
