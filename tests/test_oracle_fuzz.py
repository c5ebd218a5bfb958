"""Bounded run of the randomised oracle-vs-reference campaign (tools/fuzz_oracle.py): random single-op graphs through
the REAL reference CPU backend and through the oracle restatement, every byte equal.  The long campaign of the round
(1e9 outputs) is documented in DESIGN.md; this keeps a few seconds of it in the CPU suite."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from oracle import ref_capi  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_capi.available(), reason="reference library not built (oracle/build_ref.py)")


@pytest.mark.parametrize("dtype,seed", [("uint8", 11), ("int8", 12)])
def test_random_graphs_oracle_is_the_reference(dtype, seed):
    import fuzz_oracle
    graphs, outputs, bad = fuzz_oracle.campaign(dtype, 6.0, seed)
    assert graphs > 20 and outputs > 100000
    assert bad == 0, "%d of %d outputs differ from the reference" % (bad, outputs)
