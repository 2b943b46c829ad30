"""Oracle vs the reference's OWN scoring code (fixtures from oracle/make_golden.py,
which imported data_utils.py from /root/reference)."""
import os

import numpy as np
import pytest

from oracle import sse_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["small", "eval"])
def test_oracle_scoring_matches_reference(name):
    z = np.load(os.path.join(G, "scoring_%s.npz" % name))
    labels = [[int(v) for v in row if v >= 0] for row in z["labels"]]
    scores = O.scores_f64(z["src"], z["tgt64"])
    sc, idx = O.sorted_results(scores)
    assert np.array_equal(idx[:, :16], z["ranked_idx"])          # no exact ties in the fixture
    assert np.array_equal(sc[:, :16], z["ranked_score"])
    for j, k in enumerate((1, 3, 10)):
        assert O.topk_tight_accuracy(k, labels, idx) == pytest.approx(z["accs_tight"][j], abs=1e-15)
        assert O.topk_accuracy(k, labels, idx) == pytest.approx(z["accs_loose"][j], abs=1e-15)
    assert O.evaluator_accuracy(z["src"], z["tgt64"], labels) == pytest.approx(z["eval_acc"].tolist(), abs=1e-15)


def test_index_text_roundtrip_matches_reference_parse():
    z = np.load(os.path.join(G, "scoring_small.npz"))
    lines = [O.format_index_line("t%d" % i, "Sentence %d" % i, v) for i, v in enumerate(z["tgt32"])]
    _, _, enc = O.parse_index_lines(lines)
    assert np.array_equal(enc, z["tgt64"])        # float(str(np.float32)) exactly as sse_evaluator.py:87


def test_oracle_scoring_matches_reference_at_encoding_size_512():
    """configs[4] (encoding_size 512): one evaluator batch, 600 queries x 571 targets, ranked by the reference's own
    np.dot + getSortedResults (fixture scoring_wide512.npz holds its outputs; the seeded inputs are regenerated)."""
    from oracle.make_golden import wide_inputs
    z = np.load(os.path.join(G, "scoring_wide512.npz"))
    src, _, tgt64, labels = wide_inputs()
    assert float(src.astype(np.float64).sum()) == float(z["src_sum"]) and float(tgt64.sum()) == float(z["tgt_sum"])
    sc, idx = O.sorted_results(O.scores_f64(src, tgt64))
    assert np.array_equal(idx[:, :16], z["ranked_idx"]) and np.array_equal(sc[:, :16], z["ranked_score"])
    for j, k in enumerate((1, 3, 10)):
        assert O.topk_tight_accuracy(k, labels, idx) == pytest.approx(z["accs_tight"][j], abs=1e-15)
