"""targetEncodingIndex.tsv text I/O (csrc/index_io.cpp through sequence-semantic-embedding_amd/index_io.py) against
the reference's own Python loops: `",".join([str(n) for n in vec])` on numpy.float32 (sse_index.py:95) and
`[float(f) for f in field.split(",")]` (sse_evaluator.py:87).  Host-only: runs without a GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sse_amd  # noqa: E402
from sse_amd import index_io  # noqa: E402


def _ref_format(rows):
    return [",".join([str(n) for n in row]) for row in rows]


def _ref_parse(fields):
    return np.array([[float(f) for f in field.strip().split(",")] for field in fields])


EDGE = [0.0, -0.0, 1.0, -1.0, 1e-4, 9.9999997e-5, 1.0000001e-4, 1e16, 9.9999998e15, 1.0000001e16, 0.1, 0.5, 123456.7,
        1e8, 16777216.0, 3.4028235e38, 1e-45, 1.17549435e-38, np.inf, -np.inf, np.nan, 1e15, 2.5e-4, 100.0, 1e-3,
        0.001953125, 0.3, 1 / 3, 2 / 3, 8.5e-5, -7.2e-7]


def test_format_matches_numpy_str_on_edges_encodings_and_random_bits():
    rng = np.random.RandomState(0)
    enc = rng.standard_normal((3000, 64)).astype(np.float32)
    enc /= np.linalg.norm(enc, axis=1, keepdims=True)                         # what sse_index writes
    tiny = (rng.standard_normal((500, 64)) * rng.choice([1e-3, 1e-4, 1e-5, 1e-7], size=(500, 64))).astype(np.float32)
    bits = rng.randint(0, 2 ** 32, size=(3000, 64), dtype=np.uint64).astype(np.uint32).view(np.float32)
    edge = np.array([EDGE + [0.25] * (64 - len(EDGE))], np.float32)
    for arr in (edge, enc, tiny, bits):
        assert index_io.format_rows(arr) == _ref_format(arr)
    assert index_io.format_rows(np.zeros((0, 8), np.float32)) == []


def test_parse_matches_python_float_and_roundtrips():
    rng = np.random.RandomState(1)
    enc = rng.standard_normal((2000, 48)).astype(np.float32) / 7
    fields = _ref_format(enc)
    got = index_io.parse_rows(fields)
    want = _ref_parse(fields)
    assert got.dtype == np.float64 and np.array_equal(got, want)
    assert np.array_equal(got.astype(np.float32), enc)                       # shortest repr round-trips the float32
    # hand-written spellings float() accepts
    odd = ["1,2.5,-3e-2,+4,.5,6.,1E3,inf,-inf, 7 ", "0.1,0.2,0.3,1e-320,1e400,-0.0,12345678901234567890,-1e-400,1,2"]
    assert np.array_equal(index_io.parse_rows(odd), _ref_parse(odd))
    assert index_io.parse_rows([], 5).shape == (0, 5)


def test_parse_errors_like_the_reference():
    with pytest.raises(ValueError):
        index_io.parse_rows(["1,2,x"])                    # float('x') raises in the reference
    with pytest.raises(ValueError):
        index_io.parse_rows(["1,2,3", "1,2"])             # ragged rows cannot form the [N,S] matrix
    with pytest.raises(ValueError):
        index_io.parse_rows(["1,2,3", "1,2,3,4"])


def test_index_file_written_and_loaded_identically(tmp_path):
    """format -> file -> load_index_file == the reference's writer/reader pair on the same encodings."""
    from sse_amd import sse_evaluator
    rng = np.random.RandomState(2)
    enc = rng.standard_normal((300, 32)).astype(np.float32)
    enc /= np.linalg.norm(enc, axis=1, keepdims=True)
    path = str(tmp_path / "targetEncodingIndex.tsv")
    vecs = index_io.format_rows(enc)
    with open(path, "w", encoding="utf-8") as f:
        for i, v in enumerate(vecs):
            f.write("id%d\tSentence %d\t%s\n" % (i, i, v))
        f.write("broken line without fields\n")
    ids, names, got, id_map = sse_evaluator.load_index_file(path)
    assert ids == ["id%d" % i for i in range(300)] and id_map["id7"] == 7 and names[3] == "Sentence 3"
    assert np.array_equal(got, _ref_parse(_ref_format(enc)))


def test_format_and_parse_property_based():
    """hypothesis: any float32 bit pattern formats exactly like numpy's str() and, when finite, parses back to itself."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.integers(min_value=0, max_value=2 ** 32 - 1), min_size=1, max_size=40))
    def check(bits):
        row = np.array(bits, np.uint32).view(np.float32)[None, :]
        got = index_io.format_rows(row)
        assert got == _ref_format(row)
        back = index_io.parse_rows(got, row.shape[1])
        want = _ref_parse(got)
        assert np.array_equal(back, want, equal_nan=True)
        fin = np.isfinite(row[0])
        assert np.array_equal(back[0][fin].astype(np.float32), row[0][fin])

    check()
