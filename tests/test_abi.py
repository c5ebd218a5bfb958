"""The C-ABI library loads here (no GPU) and exports every symbol include/tengine_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "tengine_amd.h")).read()
    return sorted(set(re.findall(r"TAMD_API\s+[\w\s\*]+?\b(tamd_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tengine_amd import capi
    assert os.path.exists(capi.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    assert sorted(capi.EXPORTS) == names


def test_no_device_fails_loudly():
    """Without a GPU prerun must fail with an error, never fall back to a CPU path."""
    from helpers import conv_graph
    from tengine_amd import capi, tm2
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    g, _ = conv_graph(0, 1, 16, 8, 8, 16, 1)
    with pytest.raises(capi.TamdError):
        capi.Graph(tm2.write_tm2(g))


def test_tm2_loader_rejects_garbage():
    from tengine_amd import capi
    L = capi.lib()
    assert not L.tamd_graph_load_tm2(b"\x02\x00" + b"\xff" * 64, 66)
    assert b"tm2" in L.tamd_last_error()


def test_product_never_imports_oracle():
    """tengine_amd/ must not reference oracle/ (the oracle is test infrastructure only)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "tengine_amd")):
        for f in fs:
            if f.endswith((".py", ".cc", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "tg_oracle" not in src, f
