"""Known-answer pins of the oracle's MODEL arithmetic against TensorFlow's own unit tests.

The reference's encoder / optimizer arithmetic is TensorFlow 1.x (`/root/reference/sse_model.py:240-242,248-249,
298,359-363`; `requirements.txt:1`), which is not installable here.  TensorFlow's unit tests, however, PUBLISH input /
output vectors for exactly the ops the reference calls; the constants below are those vectors (TensorFlow r1.0 - r1.4
source tree, the era `sse_model.py`'s `tf.contrib.rnn` / `targets=` keyword fix), and every test runs the ORACLE'S OWN
function -- the one `O.lstm_forward` / `O.train_step` are built from -- on them:

  * `tensorflow/contrib/rnn/python/kernel_tests/core_rnn_cell_test.py::RNNCellTest.testBasicLSTMCell`
  * `tensorflow/python/training/adagrad_test.py::AdagradOptimizerTest.{testBasic, testSparseBasic, testSparseRepeatedIndices}`
  * `tensorflow/python/kernel_tests/clip_ops_test.py::ClipTest.testClipByGlobalNorm{Clipped, WithIndexedSlicesClipped, NotClipped}`
  * `tensorflow/python/ops/nn_test.py::L2NormalizeTest`, `nn_xent_test.py::SigmoidCrossEntropyWithLogitsTest`,
    `nn_test.py::WeightedCrossEntropyTest` (these compare the op with a plain numpy expression on fixed inputs: the
    inputs and the expression are restated here)

What stays unpinned after this file: nothing of the per-op arithmetic the LSTM modes use; the COMPOSITION (static_rnn
unrolling, tf.gradients through it) is pinned by the independent torch-autograd restatement in
tests/test_oracle_torch_step.py.  The CNN mode has no reference behaviour to pin (its graph does not build,
sse_model.py:206)."""
import numpy as np

from oracle import sse_oracle as O

F32 = np.float32


# --------------------------------------------------------------------------
# BasicLSTMCell  (sse_model.py:240,248,262)
# --------------------------------------------------------------------------

def test_basic_lstm_cell_matches_tensorflow_testBasicLSTMCell():
    """core_rnn_cell_test.testBasicLSTMCell: MultiRNNCell of two BasicLSTMCell(2, state_is_tuple=False), every weight
    0.5 (variable_scope initializer constant 0.5; the cell's bias is created with its own zeros initializer),
    x = [[1, 1]], state m = 0.1 * ones([1, 8]) laid out [c1 | h1 | c2 | h2].  TF asserts
        output  == [[0.24024698, 0.24024698]]
        state   == [[0.68967271, 0.68967271, 0.44848421, 0.44848421, 0.39897051, 0.39897051, 0.24024698, 0.24024698]]"""
    K = np.full((2 + 2, 4 * 2), 0.5, F32)           # [(input 2 + units 2), 4 * units]
    b = np.zeros(8, F32)
    x = np.array([[1.0, 1.0]], F32)
    c1 = h1 = c2 = h2 = np.full((1, 2), 0.1, F32)
    c1n, h1n, _ = O.lstm_cell_step(x, c1, h1, K, b)          # layer 1
    c2n, h2n, _ = O.lstm_cell_step(h1n, c2, h2, K, b)        # layer 2 reads layer 1's new h
    state = np.concatenate([c1n, h1n, c2n, h2n], axis=1)
    want = np.array([[0.68967271, 0.68967271, 0.44848421, 0.44848421, 0.39897051, 0.39897051, 0.24024698, 0.24024698]])
    assert np.abs(state - want).max() < 5e-7                 # float32 arithmetic against 8 published digits
    assert np.abs(h2n - np.array([[0.24024698, 0.24024698]])).max() < 5e-7


def test_basic_lstm_cell_gate_order_and_forget_bias_are_what_the_kat_needs():
    """The KAT above has all gates equal, so it cannot see a permutation of i, j, f, o.  This one can: with distinct gate
    columns the published TF formula (rnn_cell_impl.BasicLSTMCell.call: `i, j, f, o = split(..., 4, axis=1)`;
    `new_c = c * sigmoid(f + forget_bias) + sigmoid(i) * tanh(j)`; `new_h = tanh(new_c) * sigmoid(o)`) is evaluated in
    float64 straight from that text and compared with the oracle's step; any other gate order or a forget bias folded the
    wrong way changes the numbers by > 1e-2."""
    rng = np.random.RandomState(0)
    E, H = 3, 2
    K = rng.uniform(-1, 1, size=(E + H, 4 * H)).astype(F32)
    b = rng.uniform(-1, 1, size=4 * H).astype(F32)
    x = rng.uniform(-1, 1, size=(4, E)).astype(F32)
    c = rng.uniform(-1, 1, size=(4, H)).astype(F32)
    h = rng.uniform(-1, 1, size=(4, H)).astype(F32)
    g = np.concatenate([x, h], 1).astype(np.float64) @ K.astype(np.float64) + b
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    i, j, f, o = np.split(g, 4, axis=1)
    new_c = c * sig(f + 1.0) + sig(i) * np.tanh(j)
    new_h = np.tanh(new_c) * sig(o)
    cn, hn, _ = O.lstm_cell_step(x, c, h, K, b)
    assert np.abs(cn - new_c).max() < 1e-6 and np.abs(hn - new_h).max() < 1e-6
    for perm in ([0, 2, 1, 3], [2, 0, 1, 3], [0, 1, 3, 2]):          # a permuted gate layout is NOT the same function
        gp = np.concatenate([np.split(g, 4, axis=1)[q] for q in perm], axis=1)
        i2, j2, f2, o2 = np.split(gp, 4, axis=1)
        assert np.abs((c * sig(f2 + 1.0) + sig(i2) * np.tanh(j2)) - cn).max() > 1e-2


def test_lstm_forward_is_the_cell_step_unrolled_from_zero_state():
    """static_rnn (sse_model.py:241-242): zero initial state, the cell called T times, the LAST output taken."""
    rng = np.random.RandomState(1)
    V, E, H, T, B = 11, 3, 4, 5, 6
    emb = rng.uniform(-1, 1, size=(V, E)).astype(F32)
    K = rng.uniform(-0.5, 0.5, size=(E + H, 4 * H)).astype(F32)
    b = rng.uniform(-0.5, 0.5, size=4 * H).astype(F32)
    ids = rng.randint(0, V, size=(B, T)).astype(np.int32)
    c = h = np.zeros((B, H), F32)
    for t in range(T):
        c, h, _ = O.lstm_cell_step(emb[ids[:, t]], c, h, K, b)
    assert np.array_equal(O.lstm_forward(emb, K, b, ids), h)


# --------------------------------------------------------------------------
# Adagrad  (sse_model.py:359,363)
# --------------------------------------------------------------------------

def test_adagrad_dense_matches_tensorflow_testBasic():
    """adagrad_test.testBasic: AdagradOptimizer(3.0, initial_accumulator_value=0.1), var0 = [1, 2], var1 = [3, 4],
    grads 0.1 / 0.01, three steps -> [-1.6026098728179932, -0.6026098728179932], [2.715679168701172, 3.715679168701172]."""
    var0, var1 = np.array([1.0, 2.0], F32), np.array([3.0, 4.0], F32)
    acc0, acc1 = np.full(2, 0.1, F32), np.full(2, 0.1, F32)
    for _ in range(3):
        O.adagrad_apply_dense(var0, acc0, np.array([0.1, 0.1], F32), 3.0)
        O.adagrad_apply_dense(var1, acc1, np.array([0.01, 0.01], F32), 3.0)
    assert np.allclose(var0, [-1.6026098728179932, -0.6026098728179932], rtol=1e-6, atol=1e-6)   # TF: assertAllCloseAccordingToType
    assert np.allclose(var1, [2.715679168701172, 3.715679168701172], rtol=1e-6, atol=1e-6)
    assert O.ADAGRAD_INIT_ACC == F32(0.1)                    # the optimizer default the reference relies on (sse_model.py:359)
    st = O.new_optimizer_state({"w": np.zeros((2, 2), F32)})
    assert np.all(st["w"] == F32(0.1))


def test_adagrad_sparse_matches_tensorflow_testSparseBasic():
    """adagrad_test.testSparseBasic: var0 = [[1], [2]] with IndexedSlices([[0.1]], indices [0]); var1 = [[3], [4]] with
    IndexedSlices([[0.01]], indices [1]); three steps -> [[-1.6026098728179932], [2.0]] and [[3.0], [3.715679168701172]]
    (untouched rows AND their accumulators stay)."""
    var0, var1 = np.array([[1.0], [2.0]], F32), np.array([[3.0], [4.0]], F32)
    acc0, acc1 = np.full((2, 1), 0.1, F32), np.full((2, 1), 0.1, F32)
    for _ in range(3):
        O.adagrad_apply_sparse(var0, acc0, np.array([0]), np.array([[0.1]], F32), 3.0)
        O.adagrad_apply_sparse(var1, acc1, np.array([1]), np.array([[0.01]], F32), 3.0)
    assert np.allclose(var0, [[-1.6026098728179932], [2.0]], rtol=1e-6, atol=1e-6)
    assert np.allclose(var1, [[3.0], [3.715679168701172]], rtol=1e-6, atol=1e-6)
    assert acc0[1, 0] == F32(0.1) and acc1[0, 0] == F32(0.1)


def test_adagrad_sparse_repeated_indices_equal_the_aggregated_update():
    """adagrad_test.testSparseRepeatedIndices: IndexedSlices([[0.1], [0.1]], indices [1, 1]) must update exactly like
    IndexedSlices([[0.2]], indices [1]) -- duplicates are SUMMED before the update (not applied one after the other, which
    would square them separately).  This is what makes the word_embedding update of sse_model.py:163-164,363 well defined
    for a token that occurs in several rows."""
    rep, agg = np.array([[1.0], [2.0]], F32), np.array([[1.0], [2.0]], F32)
    acc_r, acc_a = np.full((2, 1), 0.1, F32), np.full((2, 1), 0.1, F32)
    for _ in range(3):
        O.adagrad_apply_sparse(rep, acc_r, np.array([1, 1]), np.array([[0.1], [0.1]], F32), 3.0)
        O.adagrad_apply_sparse(agg, acc_a, np.array([1]), np.array([[0.2]], F32), 3.0)
    assert np.array_equal(rep, agg) and np.array_equal(acc_r, acc_a)
    seq = np.array([[1.0], [2.0]], F32)                                  # the WRONG reading, for contrast
    acc_s = np.full((2, 1), 0.1, F32)
    for g in (0.1, 0.1):
        O.adagrad_apply_sparse(seq, acc_s, np.array([1]), np.array([[g]], F32), 3.0)
    O.adagrad_apply_sparse(agg2 := np.array([[1.0], [2.0]], F32), np.full((2, 1), 0.1, F32), np.array([1]), np.array([[0.2]], F32), 3.0)
    assert abs(float(seq[1, 0] - agg2[1, 0])) > 1e-2


# --------------------------------------------------------------------------
# clip_by_global_norm  (sse_model.py:362)
# --------------------------------------------------------------------------

X0 = np.array([[-2.0, 0.0, 0.0], [4.0, 0.0, 0.0]], F32)
X1 = np.array([1.0, -2.0], F32)


def test_clip_by_global_norm_matches_tensorflow_clipped():
    """clip_ops_test.testClipByGlobalNormClipped: x0 = [[-2,0,0],[4,0,0]], x1 = [1,-2]; global norm
    sqrt(1 + 4^2 + 2^2 + 2^2) = 5; clip_norm 4 -> [[-1.6,0,0],[3.2,0,0]], [0.8,-1.6]."""
    out, gn = O.clip_by_global_norm({"x0": X0, "x1": X1}, 4.0)
    assert gn == F32(5.0)
    assert np.allclose(out["x0"], [[-1.6, 0.0, 0.0], [3.2, 0.0, 0.0]], rtol=1e-6)
    assert np.allclose(out["x1"], [0.8, -1.6], rtol=1e-6)


def test_clip_by_global_norm_matches_tensorflow_indexed_slices():
    """clip_ops_test.testClipByGlobalNormWithIndexedSlicesClipped: the same numbers with x1 an IndexedSlices(values
    [1, -2], indices [3, 4]) -- the norm is taken over the slice VALUES (un-deduplicated), the values are scaled."""
    x1 = (np.array([3, 4]), X1.reshape(2, 1))
    out, gn = O.clip_by_global_norm({"x0": X0, "x1": x1}, 4.0)
    assert gn == F32(5.0)
    assert np.allclose(out["x0"], [[-1.6, 0.0, 0.0], [3.2, 0.0, 0.0]], rtol=1e-6)
    assert np.array_equal(out["x1"][0], [3, 4]) and np.allclose(out["x1"][1].ravel(), [0.8, -1.6], rtol=1e-6)
    # duplicates are NOT merged before the norm (an IndexedSlices with a repeated index keeps both value rows)
    dup = (np.array([7, 7]), np.array([[3.0], [4.0]], F32))
    assert O.global_norm({"d": dup}) == F32(5.0)             # sqrt(9 + 16), not |3 + 4| = 7


def test_clip_by_global_norm_matches_tensorflow_not_clipped():
    """clip_ops_test.testClipByGlobalNormNotClipped: clip_norm 6 > norm 5 -> tensors unchanged, norm still reported 5."""
    out, gn = O.clip_by_global_norm({"x0": X0, "x1": X1}, 6.0)
    assert gn == F32(5.0)
    assert np.allclose(out["x0"], X0, rtol=1e-6) and np.allclose(out["x1"], X1, rtol=1e-6)


# --------------------------------------------------------------------------
# l2_normalize, weighted_cross_entropy_with_logits  (sse_model.py:282-283,298)
# --------------------------------------------------------------------------

def test_l2_normalize_matches_tensorflow_definition():
    """nn_impl.l2_normalize: x * rsqrt(maximum(reduce_sum(square(x), dim), epsilon)), epsilon = 1e-12;
    nn_test.L2NormalizeTest compares with x / sqrt(sum(x^2)) on random x (tolerance 1e-6 there)."""
    rng = np.random.RandomState(0)
    x = rng.rand(20, 7).astype(F32)
    assert np.abs(O.l2_normalize(x) - x / np.sqrt((x.astype(np.float64) ** 2).sum(-1, keepdims=True))).max() < 1e-6
    tiny = np.array([[1e-8, 0.0, -1e-8]], F32)               # sum of squares 2e-16 < epsilon: divided by sqrt(1e-12), not by the norm
    assert np.allclose(O.l2_normalize(tiny), tiny / 1e-6, rtol=1e-6)
    assert np.array_equal(O.l2_normalize(np.zeros((1, 3), F32)), np.zeros((1, 3), F32))


XENT_X = np.array([-100.0, -2.0, -2.0, 0.0, 2.0, 2.0, 2.0, 100.0], F32)
XENT_Y = np.array([0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.5, 1.0], F32)


def test_weighted_cross_entropy_matches_tensorflow_tests_expression():
    """nn_xent_test.SigmoidCrossEntropyWithLogitsTest / nn_test.WeightedCrossEntropyTest: logits
    [-100, -2, -2, 0, 2, 2, 2, 100], targets [0, 0, 1, 0, 0, 1, 0.5, 1]; expected
    -(q * y * log(pred) + (1 - y) * log(1 - pred)) with pred = sigmoid(x) clipped to [eps, 1 - eps], eps = 1e-4 (the
    clip only matters at +-100, where both sides are 0 up to e^-100)."""
    pred = 1.0 / (1.0 + np.exp(-XENT_X.astype(np.float64)))
    eps = 0.0001
    pred = np.minimum(np.maximum(pred, eps), 1 - eps)
    for q in (1.0, 2.0):
        want = -(q * XENT_Y * np.log(pred) + (1 - XENT_Y) * np.log(1 - pred))
        got = O.weighted_cross_entropy_with_logits(XENT_Y, XENT_X, q)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-3), q       # TF's own tolerance there: assertAllClose(np_loss, tf_loss, atol=0.001)
        mid = slice(1, 7)                                            # away from the +-100 ends the clip is idle: tight
        assert np.allclose(got[mid], want[mid], rtol=1e-6, atol=1e-6), q
    # the reference's loss: pos_weight = 1.0, logits = 64 * cosine (sse_model.py:298) -- through loss_and_acc
    ns = O.l2_normalize(np.random.RandomState(1).randn(8, 5).astype(F32))
    nt = O.l2_normalize(np.random.RandomState(2).randn(8, 5).astype(F32))
    loss, _, cos = O.loss_and_acc(ns, nt, XENT_Y)
    assert np.isclose(loss, np.mean(O.weighted_cross_entropy_with_logits(XENT_Y, F32(64.0) * cos, 1.0)), rtol=1e-6)


def test_train_step_is_built_from_the_pinned_pieces():
    """O.train_step == gradients -> the KAT-pinned clip_by_global_norm -> the KAT-pinned Adagrad updates, nothing else."""
    cfg = dict(network_mode="dual-encoder", vocab_size=30, embedding_size=4, encoding_size=6, src_cell_size=5, tgt_cell_size=7,
               max_seq_length=4, targetSpaceSize=3)
    p = O.init_params(cfg, seed=0)
    for k in p:
        if k.endswith("_M"):
            p[k] = (p[k] * 20).astype(F32)                   # clipping engaged
    q = {k: v.copy() for k, v in p.items()}
    st_p, st_q = O.new_optimizer_state(p), O.new_optimizer_state(q)
    rng = np.random.RandomState(3)
    src, tgt = rng.randint(0, 30, size=(6, 4)), rng.randint(0, 30, size=(6, 4))
    z = np.array([1, 0, 1, 0, 1, 0], F32)
    O.train_step(p, st_p, cfg, src, tgt, z, 0.9)
    _, _, grads = O.gradients(q, cfg, src, tgt, z)
    assert O.global_norm(grads) > 5.0
    grads, _ = O.clip_by_global_norm(grads, 5.0)
    for name, g in grads.items():
        if isinstance(g, tuple):
            O.adagrad_apply_sparse(q[name], st_q[name], g[0], g[1], 0.9)
        else:
            O.adagrad_apply_dense(q[name], st_q[name], g, 0.9)
    for k in p:
        assert np.array_equal(p[k], q[k]) and np.array_equal(st_p[k], st_q[k]), k
