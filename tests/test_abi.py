"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports exactly the symbols include/mvf_hotpath.h declares; the Python binding covers
them all; the product has no CPU fallback.  (No compute calls: there is no GPU here.)"""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from mono_vifi_amd import _native
    return _native


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mvf_hotpath.h")).read()
    return sorted(set(re.findall(r"^MVF_API[^;(]*?\b(mvf_[a-z0-9_]+)\s*\(", text, flags=re.M)))


def test_header_declares_the_path():
    syms = header_symbols()
    for must in ("mvf_warp_fwd", "mvf_warp_bwd", "mvf_photo_fwd", "mvf_photo_bwd", "mvf_unit_fwd",
                 "mvf_unit_bwd", "mvf_ssim_fwd", "mvf_smooth_fwd", "mvf_backproject_fwd",
                 "mvf_project_fwd", "mvf_grid_sample_fwd", "mvf_pose_fwd", "mvf_disp_to_depth_fwd"):
        assert must in syms
    assert len(syms) >= 25


def test_library_exports_every_declared_symbol(built):
    handle = ctypes.CDLL(built.LIB_PATH)
    for name in header_symbols():
        assert hasattr(handle, name), f"{name} declared in include/mvf_hotpath.h but not exported"
    assert handle.mvf_abi_version() == built.ABI_VERSION


def test_fast_build_is_a_separate_library_with_the_same_abi(built):
    """Round 6: the opt-in fast mode (`make fast`, -DMVF_FAST_SSIM) is a SEPARATE library with the same entry points and
    ABI version; the product loads the exact-mode library unless a process names another one with MVF_HOTPATH_LIB, and the
    parity suite never does."""
    fast = os.path.join(os.path.dirname(built.LIB_PATH), "libmvf_hotpath_fast.so")
    assert os.path.exists(fast), fast
    assert os.path.basename(built.LIB_PATH) == "libmvf_hotpath.so" or os.environ.get("MVF_HOTPATH_LIB")
    handle = ctypes.CDLL(fast)
    for name in header_symbols():
        assert hasattr(handle, name), f"{name} missing from the fast build"
    assert handle.mvf_abi_version() == built.ABI_VERSION
    for f in os.listdir(os.path.join(ROOT, "tests")):
        if f.endswith(".py") and f != "test_abi.py":
            assert "libmvf_hotpath_fast" not in open(os.path.join(ROOT, "tests", f)).read(), f


def test_binding_covers_the_header(built):
    assert sorted(built.EXPORTS) == header_symbols()
    lib = built.lib()
    lib.mvf_workspace_floats.restype = ctypes.c_size_t
    assert lib.mvf_workspace_floats(12, 192, 640) >= 12 * 32 + 12 * 120 * 4


def test_no_cpu_fallback(built):
    from mono_vifi_amd import layers
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        layers.SSIM()(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        layers.disp_to_depth(torch.rand(1, 1, 8, 8), 0.1, 100.0)


def test_product_never_imports_the_oracle():
    """The product path must not import, include or link anything under oracle/ (comments
    may cite the proof program by name)."""
    pkg = os.path.join(ROOT, "mono-vifi_amd")
    py_dep = re.compile(r"^\s*(from|import)\s+oracle\b|importlib[^\n]*oracle|CDLL\([^\n]*oracle", re.M)
    c_dep = re.compile(r"#\s*include[^\n]*oracle|dlopen\([^\n]*oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            text = None
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not py_dep.search(text), f"{f} depends on the oracle"
            elif f.endswith((".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not c_dep.search(text), f"{f} includes the oracle"
            elif f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, "the product build links the oracle"


def test_units_entry_points_reject_bad_arguments(built):
    """mvf_units_fwdbwd / mvf_units_fwdbwd_scale validate their arguments BEFORE the first HIP call and
    return hipErrorInvalidValue (1) -- checked here without a GPU: unit counts outside 1..MVF_MAX_UNITS, more
    than one source pair, missing workspace / tickets / descriptors / required pointers, a plane whose byte
    offsets do not fit 28 bits (the tap stash of the unit kernel carries four flag bits above them);
    an empty batch is a no-op (0)."""
    import ctypes as C
    from mono_vifi_amd import _native as nat
    lib = nat.lib()
    INVALID = 1
    n = nat.MAX_UNITS
    descs = (nat.UnitDesc * n)()
    ws = (C.c_float * 64)()
    tk = (C.c_int32 * 16)()
    P = C.cast(descs, C.c_void_p)
    W, T = C.cast(ws, C.c_void_p), C.cast(tk, C.c_void_p)

    def call(units=P, n_units=1, S=2, flags=0, ws=W, tickets=T, B=1, H=32, Wd=64):
        return lib.mvf_units_fwdbwd(units, n_units, S, flags, 1e-3, 0.01, 9.99, 1e-7, ws, tickets, B, H, Wd, None)

    assert call(n_units=0) == INVALID and call(n_units=n + 1) == INVALID
    assert call(units=None) == INVALID
    assert call(S=0) == INVALID and call(S=3) == INVALID
    assert call(ws=None) == INVALID and call(tickets=None) == INVALID
    assert call(B=0) == 0 and call(H=0) == 0                         # nothing to do
    assert call(H=8192, Wd=8192) == INVALID                          # 2^28 bytes per plane
    assert call(H=1 << 22, Wd=8) == INVALID
    assert call() == INVALID                                          # descriptor without its required pointers
    sd = (nat.UnitScaleDesc * n)()
    SP = C.cast(sd, C.c_void_p)
    assert lib.mvf_units_fwdbwd_scale(None, 1, 1e-3, 1, 2, 32, 64, None) == INVALID
    assert lib.mvf_units_fwdbwd_scale(SP, 0, 1e-3, 1, 2, 32, 64, None) == INVALID
    assert lib.mvf_units_fwdbwd_scale(SP, n + 1, 1e-3, 1, 2, 32, 64, None) == INVALID
    assert lib.mvf_units_fwdbwd_scale(SP, 1, 1e-3, 0, 2, 32, 64, None) == 0
    assert lib.mvf_units_fwdbwd_scale(SP, 1, 1e-3, 1, 2, 32, 64, None) == INVALID     # missing pointers
    assert lib.mvf_units_workspace_floats(3, 12, 192, 640) > 0 and lib.mvf_units_ticket_ints(3, 12) >= 3
