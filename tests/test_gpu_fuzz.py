"""A fixed-seed slice of tools/fuzz_parity.py in the suite: random model shapes (cell sizes 5..256, encodings 2..256,
T 5..80, batch 1..200, all four network modes, heavy padding) -- encode, score/top-k and one train step against the
oracle.  The sweep found the CNN LDS-tile limit and a tie-order assumption before they reached a user."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_slice():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    failures, worst = mod.main(n_cases=40, seed=5, verbose=False)
    assert failures == 0, worst
    assert worst["enc"] < 1e-4 and worst["score"] < 1e-12


def test_fuzz_large_batches_slice():
    """A fixed-seed slice of tools/fuzz_large_batches.py: the large-batch inference paths of round 6 (gate-split small-cell kernel,
    device-side PAD-prefix bucketing) in every option combination -- same bits in the caller's row order -- and against the oracle."""
    spec = importlib.util.spec_from_file_location("fuzz_large_batches", os.path.join(ROOT, "tools", "fuzz_large_batches.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    failures, worst = mod.main(n_cases=8, seed=3, verbose=False)
    assert failures == 0 and worst < 1e-4, (failures, worst)
