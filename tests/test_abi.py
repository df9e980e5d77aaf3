"""The C-ABI library loads on a CPU-only host and exports every symbol include/dmnerf_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dmnerf_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dmnerf_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("dmnerf_render_rays_fwd", "dmnerf_mlp_fwd_rays", "dmnerf_mlp_fwd_embedded", "dmnerf_composite_fwd",
                 "dmnerf_sample_pdf", "dmnerf_importance_resample", "dmnerf_raygen", "dmnerf_pack_weights",
                 "dmnerf_composite_bwd", "dmnerf_mlp_bwd_data", "dmnerf_mlp_fwd_rays_train", "dmnerf_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol_and_binding_matches():
    from dm_nerf_amd import _lib
    lib = _lib.load()                                   # raises if the .so is missing: no fallback
    syms = declared_symbols()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dmnerf_hip.h but not exported"
    assert set(_lib.SIGNATURES) == set(syms), set(_lib.SIGNATURES) ^ set(syms)
    assert lib.dmnerf_abi_version() == 8
    assert isinstance(lib.dmnerf_device_count(), int)   # 0 on a CPU-only host, never an error


def test_host_only_calls_validate_arguments():
    from dm_nerf_amd import _lib
    lib = _lib.load()
    assert lib.dmnerf_param_count(13) == 696338
    assert lib.dmnerf_param_count(0) == -1 and lib.dmnerf_blob_floats(500) == -1
    assert lib.dmnerf_train_save_floats(100) == 2466 * 128      # 2394 feature rows + 72 mask words per sample, rows padded to 32 samples
    assert lib.dmnerf_blob_t_floats(13) == 1024 + (1 + 2 + 7 * 4 + 2) * 16384          # table + 31 quarters + 2 landing quarters
    # argument errors are reported before anything touches a device
    rc = lib.dmnerf_composite_fwd(None, None, None, 4, 64, 14, None, None, None, None, None)
    assert rc == -1 and "null" in _lib.last_error()
    rc = lib.dmnerf_raygen(480, 640, None, None, 0, 10, None, None, None)
    assert rc == -1


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    from dm_nerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU / eager fallback"):
        _lib.load()
