"""The C-ABI library loads and exports every symbol include/dfk.h declares (no GPU needed)."""
import ctypes
import os
import re

from superplus_amd import dfk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "dfk.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(dfk_[a-z_0-9]+)\s*\(", h)))


def test_library_exports_header():
    lib = dfk.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"libdfk.so does not export {n}"
    assert sorted(dfk.EXPORTS) == names
    lib.dfk_abi_version.restype = ctypes.c_int
    assert lib.dfk_abi_version() == dfk.ABI_VERSION


def test_struct_sizes_match_header():
    assert ctypes.sizeof(dfk.Config) == 88
    assert dfk.ENTRY_DTYPE.itemsize == 32
    assert ctypes.sizeof(dfk.Stats) == 9 * 8 + 8 * 4 + 8 + 8 * 8


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device dfk_create must fail, not degrade."""
    import torch
    if torch.cuda.is_available():
        return
    try:
        dfk.Dfk(K=48)
    except dfk.DfkError as e:
        assert e.code in (-2, -3)
    else:
        raise AssertionError("dfk_create succeeded without a GPU")


def test_create_rejects_bad_config():
    import pytest
    for kw in (dict(K=47), dict(K=48, min_bc=9), dict(K=48, min_freq=0), dict(K=48, minimizer_len=20)):
        with pytest.raises(dfk.DfkError):
            dfk.Dfk(**kw)
