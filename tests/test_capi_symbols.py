"""CPU: the C-ABI library loads and exports every symbol include/otter_hip.h declares; the ctypes table matches the
header one-to-one (no compute calls -- there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "otter_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(otter_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from otter_amd import _capi, build

    build.build(verbose=False)  # hipcc cross-compiles gfx950 without a GPU
    return _capi.lib()


def test_header_matches_ctypes_table(lib):
    from otter_amd import _capi

    assert header_symbols() == sorted(_capi.SIGNATURES)


def test_every_declared_symbol_is_exported(lib):
    from otter_amd import _capi

    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.otter_abi_version() == _capi.ABI_VERSION == _capi.ABI_VERSION_MIRROR == 3


def test_abi_version_mirror_matches_header():
    """_capi falls back to ABI_VERSION_MIRROR when the package is deployed without its sibling include/ directory."""
    import re

    from otter_amd import _capi

    m = re.search(r"^#define\s+OTTER_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "otter_hip.h")).read(), re.M)
    assert int(m.group(1)) == _capi.ABI_VERSION_MIRROR == _capi.ABI_VERSION


def test_argument_validation_without_gpu(lib):
    """Error convention: negative status + message, no exception across the ABI, nothing launched for bad arguments."""
    import ctypes as C

    from otter_amd import _capi

    e = _capi.EpilogueArgs()
    rc = lib.otter_gemm_nt(None, 8, None, 8, None, 8, 4, 4, 8, _capi.BF16, _capi.F32, C.byref(e), None)
    assert rc == -1 and b"null" in lib.otter_last_error()
    assert lib.otter_layernorm_bwd_workspace_bytes(4096, 4096) > 0
    assert lib.otter_gemm_num_partials(4096, 16384, _capi.BF16) == 16 * 64
    assert lib.otter_attn_bwd_workspace_bytes(8, 8, 512, 64) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from otter_amd import _capi

    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_capi.OtterHipError, match="no PyTorch/CPU fallback"):
        _capi.lib()
