"""CPU: the C-ABI library loads and exports every symbol include/recommendit_hip.h declares."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "recommendit_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rihip_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from recommendit_amd import _lib
    so = _lib.LIB_PATH
    if not so.exists():
        import __graft_entry__ as g
        g.build()
    l = ctypes.CDLL(str(so))
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(l, n)]
    assert not missing, missing


def test_python_binding_covers_header():
    from recommendit_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    l = _lib.lib()
    assert l.rihip_abi_version() == 1
    assert l.rihip_target_arch() == b"gfx950"
    assert l.rihip_tower_supported(64, 128) == 1 and l.rihip_tower_supported(48, 128) == 0


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from recommendit_amd import TwoTowerModel, FAISSIndex
    m = TwoTowerModel(10, 20, 32, 64)
    with pytest.raises(RuntimeError, match="no HIP device"):
        m.user_tower(torch.tensor([1, 2]))
    with pytest.raises(RuntimeError, match="no HIP device"):
        import numpy as np
        FAISSIndex(embed_dim=32).build_ivf_index(np.zeros((4, 32), np.float32), [1, 2, 3, 4])


def test_product_never_imports_oracle():
    for f in (ROOT / "recommendit_amd").glob("*.py"):
        assert "oracle" not in f.read_text().replace("SURVEY", ""), f


def test_integration_guard_is_the_device_not_the_import():
    """the package imports on a GPU-less host, so the swap must test have_gpu()"""
    import recommendit_amd
    import torch
    assert recommendit_amd.have_gpu() == torch.cuda.is_available()
