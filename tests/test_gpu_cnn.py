"""Parity of the HIP text-CNN encoder ('source_only_cnn', sse_model.py:179-211)
and of the free target matrix modes against the CPU oracle (the reference
itself cannot run this mode: tf.concat(3, ...) at :206; the oracle follows the
evident intent, concat on axis 3)."""
import numpy as np
import pytest

from oracle import sse_oracle as O
from tests.util import make_pair, model_params, random_ids

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,E,S,T,B", [(300, 50, 512, 64, 37), (100, 30, 64, 80, 9), (60, 8, 16, 5, 3), (200, 50, 64, 33, 70)])
def test_cnn_encode_matches_oracle(V, E, S, T, B):
    params = model_params("source_only_cnn", V, E, 96, 96, S, T, N=11)
    m, p = make_pair(params, seed=2)
    ids = random_ids(np.random.RandomState(4), B, T, V, pad_frac=0.5)
    for normalize in (True, False):
        want = O.encode(p, params, "src", ids, normalize=normalize)
        got = m.encode_source(ids, normalize=normalize)
        scale = 1.0 if normalize else max(1.0, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 1e-4 * scale


@pytest.mark.parametrize("mode", ["source_only_cnn", "source-encoder-only"])
def test_free_target_matrix_is_the_target_encoding(mode):
    """norm_tgt_seq_embedding in these modes = l2_normalize(tgt_seq_embedding) [N,S], whatever is fed
    (sse_model.py:214,233,283)."""
    import sse_amd
    params = model_params(mode, 80, 16, 32, 32, 24, 8, N=13)
    m, p = make_pair(params, seed=1)
    dummy = np.zeros((13, 8), np.int32)
    got = m.encode_target(dummy)
    assert np.abs(got - O.l2_normalize(p["target_embedding/tgt_seq_embedding"])).max() < 1e-6
    assert np.array_equal(m.encode_target(dummy, normalize=False), p["target_embedding/tgt_seq_embedding"])
    with pytest.raises(sse_amd.SSEError):
        m.encode_target(np.zeros((5, 8), np.int32))
