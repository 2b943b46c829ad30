"""CPU-side checks of the drop-in boundary: libsse_hip.so builds/loads and
exports every function include/sse_hip.h declares (no compute without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sse_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sse_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    names = _declared()
    for must in ("sse_create", "sse_destroy", "sse_encode", "sse_encode_dev", "sse_index_upload",
                 "sse_index_upload_f64", "sse_score_topk", "sse_score_topk_dev", "sse_merge_topk_dev",
                 "sse_train_step", "sse_last_error"):
        assert must in names


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    import sse_amd
    lib = sse_amd.load_library()
    names = _declared()
    assert set(names) == set(sse_amd.SYMBOLS), set(names) ^ set(sse_amd.SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when no HIP device is present."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sse_amd
    params = dict(forward_only=True, network_mode="dual-encoder", predict_nbest=10, max_seq_length=8,
                  vocab_size=50, embedding_size=8, encoding_size=8, src_cell_size=16, tgt_cell_size=16,
                  learning_rate=0.9, learning_rate_decay_factor=0.99, targetSpaceSize=5)
    with pytest.raises(sse_amd.SSEError):
        sse_amd.SSEModel(params)


def test_product_package_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under the product package may import it."""
    pkg = os.path.join(ROOT, "sequence-semantic-embedding_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "sse_oracle" not in txt, f
