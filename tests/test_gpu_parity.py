"""Parity of the HIP path (through the C ABI) against the oracle.  Bit-exact: integer work."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,seed,G,pairs", [(48, 11, 60000, 3000), (48, 12, 200000, 40000), (40, 13, 80000, 8000),
                                            (60, 14, 80000, 10000)])
def test_parity_synthetic(oracle, K, seed, G, pairs):
    rs = util.make_set(seed, G, pairs)
    ref, d = util.run_both(oracle, rs, K=K)
    st = util.check_parity(ref, d)
    assert st["n_solid"] > 0
    assert d.spectrum_json() == oracle.spectrum_json(ref["hist"])
