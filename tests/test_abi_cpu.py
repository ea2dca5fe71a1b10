"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/cheetah_hip.h declares, and fails loudly (no fallback)
without a GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from cheetah_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cheetah_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cheetah_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_capi.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(_capi.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    # the binding covers the header exactly
    assert sorted(_capi.SIGNATURES) == names


def test_binding_loads_and_reports_version():
    lib = _capi.load()
    assert b"gfx950" in lib.cheetah_version()


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _capi.load()
    h = C.c_void_p()
    rc = lib.cheetah_ctx_create((C.c_int * 1)(0), 1, C.byref(h))
    assert rc in (_capi.ERR_NO_DEVICE, _capi.ERR_HIP) and not h.value
    assert lib.cheetah_last_error(None)
    from cheetah_amd import CheetahError, Engine
    with pytest.raises(CheetahError):
        Engine(0)
    # NULL context is rejected, not dereferenced
    assert lib.cheetah_fp6_mul_batch(None, None, None, None, 4) == _capi.ERR_INVALID
    assert lib.cheetah_ctx_destroy(None) == _capi.ERR_INVALID


def test_product_does_not_import_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "cheetah_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in src.replace("no oracle", ""), fn


def test_cpp_host_api_compiles_and_fails_loudly_without_gpu():
    """cheetah_amd/host/cheetah.hpp (the C++ mirror of the reference API) builds
    against the C ABI; without a GPU the program reports the error and exits 2."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, "tests", "cpp", "host_api_test")
    assert os.path.exists(exe), "build first (__graft_entry__.build())"
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no HIP device" in r.stdout
