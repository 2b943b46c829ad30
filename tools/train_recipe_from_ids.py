#!/usr/bin/env python
"""A makefile recipe of the reference, trained on the GPU from the token rows the reference's own prepare_raw_data
produced (tests/golden/{qna,crosslingual}_full_ids.npz -- the raw text is not on the GPU box), with the reference's
training loop (sse_train.py:166-229: windows of steps_per_checkpoint steps, "train_binary_acc" log line, learning-rate
decay after 5 windows without improvement, per-epoch "top 1/3/10 accuracies") and -- at the end -- the CPU oracle run on
the TRAINED weights: its encodings, its ranking, its top-1/3/10 against the device's.

    python tools/train_recipe_from_ids.py qna          [--epochs 40] [--lr 0.9]     makefile:17  (T = 1000, vocab 8000, batch 32)
    python tools/train_recipe_from_ids.py crosslingual [--epochs 2]  [--lr 0.9]     makefile:42  (shared-encoder, E 40, S 50, T 50)

Output is meant to be recorded under profiles/ (VERDICT r04 item 1c)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sse_amd  # noqa: E402
from oracle import sse_oracle as O  # noqa: E402  (the checker, after the run)

ap = argparse.ArgumentParser()
ap.add_argument("recipe", choices=["qna", "crosslingual"])
ap.add_argument("--epochs", type=int, default=40)
ap.add_argument("--lr", type=float, default=0.9)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--eval-every", type=int, default=10)
ap.add_argument("--oracle-queries", type=int, default=2000)
a = ap.parse_args()

z = np.load(os.path.join(ROOT, "tests", "golden", "%s_full_ids.npz" % a.recipe))
src, tgt = z["src_ids"].astype(np.int32), z["tgt_ids"].astype(np.int32)
positives = [[int(v) for v in row if v >= 0] for row in z["labels"]]
V, T = int(z["vocab_size"]), src.shape[1]
if a.recipe == "qna":        # makefile:17 (defaults of sse_train.py:60-74 otherwise)
    cfg = dict(network_mode="dual-encoder", embedding_size=50, encoding_size=64, src_cell_size=96, tgt_cell_size=96)
    batch, spc = 32, 10
else:                        # makefile:42
    cfg = dict(network_mode="shared-encoder", embedding_size=40, encoding_size=50, src_cell_size=96, tgt_cell_size=96)
    batch, spc = 32, 200
cfg.update(forward_only=False, predict_nbest=10, max_seq_length=T, vocab_size=V, learning_rate=a.lr,
           learning_rate_decay_factor=0.99, targetSpaceSize=len(tgt))
m = sse_amd.SSEModel(cfg)
m.init_variables(seed=a.seed)
m.handle.learning_rate = a.lr
h = m.handle
h.corpus_upload(0, src)
h.corpus_upload(1, tgt)
rng = np.random.RandomState(a.seed)
epoc_steps = len(src) // batch
print("recipe %s: %s, V=%d T=%d, %d positives, %d targets, batch %d -> %d steps per epoch, lr %.3g, %d epochs"
      % (a.recipe, {k: cfg[k] for k in ("network_mode", "embedding_size", "encoding_size", "src_cell_size")}, V, T, len(src), len(tgt),
         batch, epoc_steps, a.lr, a.epochs), flush=True)


def batch_rows():
    """Data.get_train_batch (data.py:95-115) as row numbers."""
    n = len(src)
    start = rng.randint(0, n - batch) + batch
    rows = np.arange(start, min(n, start + batch))
    trow = np.empty(2 * len(rows), np.int32)
    for i, r in enumerate(rows):
        pos = positives[r]
        trow[2 * i] = pos[rng.randint(len(pos))]
        neg = rng.randint(len(tgt))
        while neg in pos:
            neg = rng.randint(len(tgt))
        trow[2 * i + 1] = neg
    return np.repeat(rows.astype(np.int32), 2), trow, np.tile(np.array([1.0, 0.0], np.float32), len(rows))


def evaluate():
    """createIndexFile + Evaluator.eval on the device (sse_index.py:66-92, sse_evaluator.py:95-114; batches of 600,
    batch accuracies averaged unweighted)."""
    te = m.encode_target(tgt)
    h.index_upload(te.astype(np.float64))
    accs = {1: [], 3: [], 10: []}
    for b0 in range(0, len(src), 600):
        se = m.encode_source(src[b0:b0 + 600])
        _, ids = h.score_topk(se, 10)
        for n in accs:
            accs[n].append(O.topk_tight_accuracy(n, positives[b0:b0 + 600], ids))
    return [float(np.mean(accs[n])) for n in (1, 3, 10)]


step_time = loss = train_acc = 0.0
current_step, previous = 0, []
t_run = time.time()
for epoch in range(a.epochs):
    for _ in range(epoc_steps):
        t0 = time.time()
        s_rows, t_rows, lab = batch_rows()
        step_loss, step_acc = h.train_step_rows(s_rows, t_rows, lab)
        step_time += (time.time() - t0) / spc
        loss += step_loss / spc
        train_acc += step_acc / spc
        current_step += 1
        if current_step % spc == 0:
            if current_step % (spc * 10) == 0 or a.recipe != "qna":
                print("global epoc: %.3f, global step %d, learning rate %.4f step-time:%.4f loss:%.4f train_binary_acc:%.4f "
                      % (h.global_step / float(epoc_steps), h.global_step, h.learning_rate, step_time, step_loss, train_acc), flush=True)
            if len(previous) > 6 and train_acc < min(previous[-5:]):
                h.decay_learning_rate()
            previous.append(train_acc)
            step_time = loss = train_acc = 0.0
    if (epoch + 1) % a.eval_every == 0 or epoch == a.epochs - 1:
        a1, a3, a10 = evaluate()
        print("epoc#%d, task specific evaluation: top 1/3/10 accuracies: %f / %f / %f" % (epoch, a1, a3, a10), flush=True)
print("trained %d steps in %.1f s wall (%.2f ms/step incl. batch sampling)" % (current_step, time.time() - t_run, (time.time() - t_run) / current_step * 1e3))

# ---- the oracle on the trained weights
p = m.get_variables()
nq = min(a.oracle_queries, len(src))
pick = np.linspace(0, len(src) - 1, nq).astype(np.int64)
t0 = time.time()
te_o = O.encode(p, cfg, "tgt", tgt)
se_o = O.encode(p, cfg, "src", src[pick])
wsc, wids = O.topk_fast(O.scores_f64(se_o, te_o.astype(np.float64)), 10)
te_d, se_d = m.encode_target(tgt), m.encode_source(src[pick])
h.index_upload(te_d.astype(np.float64))
sc_d, ids_d = h.score_topk(se_d, 10)
lab = [positives[i] for i in pick]
margin = wsc[:, 0] - wsc[:, 1]
enc_err = max(float(np.abs(te_d - te_o).max()), float(np.abs(se_d - se_o).max()))
clear = margin > max(1e-5, 20 * enc_err)
print("oracle on the trained weights (%d queries x %d targets, %.1f s of CPU): max |encoding diff| %.2e; top 1/3/10 oracle %s device %s; "
      "top-1 ids equal %d of %d (oracle top-2 margin > %.1e: %d queries, equal there: %s); max |top-1 score diff| %.2e"
      % (nq, len(tgt), time.time() - t0, enc_err,
         ["%.4f" % O.topk_tight_accuracy(n, lab, wids) for n in (1, 3, 10)], ["%.4f" % O.topk_tight_accuracy(n, lab, ids_d) for n in (1, 3, 10)],
         int(np.sum(ids_d[:, 0] == wids[:, 0])), nq, max(1e-5, 20 * enc_err), int(clear.sum()),
         bool(np.array_equal(ids_d[clear, 0], wids[clear, 0])), float(np.abs(sc_d[:, 0] - wsc[:, 0]).max())))
K = [k for k in p if k.endswith("/kernel")]
print("trained weights: max |LSTM kernel| %.3f, max |projection| %.3f, max |embedding| %.3f"
      % (max(float(np.abs(p[k]).max()) for k in K), max(float(np.abs(p[k]).max()) for k in p if k.endswith("_M")), float(np.abs(p["word_embedding"]).max())))
