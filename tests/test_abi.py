"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads without a
GPU, exports every symbol include/reevr_amd/rvc.h declares, and fails loudly (no CPU
fallback) when there is no device."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "reevr_amd", "rvc.h")


@pytest.fixture(scope="module")
def lib():
    from reevr_amd import build
    build.build_lib()
    from reevr_amd import _lib
    return _lib.lib()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rvc_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    from reevr_amd import _lib
    names = declared_symbols()
    assert len(names) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in rvc.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_no_cpu_fallback_without_gpu(lib):
    import reevr_amd
    from reevr_amd import _lib
    if lib.rvc_device_count() > 0:
        pytest.skip("a GPU is present")
    c = reevr_amd.TwoStageFFTConvolver()
    assert c.init(512, 8192, np.ones(100, np.float32)) is False
    assert c.last_error == _lib.RVC_ERR_NO_DEVICE
    assert np.all(c.process(np.ones(16, np.float32)) == 0)


def test_reference_error_conventions_without_gpu(lib):
    """bool init: False iff a block size is 0; empty IR -> True, zeros out
    (TwoStageFFTConvolver.cpp:94-115, FFTConvolver.cpp:97-111). These paths need no device."""
    import reevr_amd
    c = reevr_amd.TwoStageFFTConvolver()
    ir = np.ones(8, np.float32)
    assert c.init(0, 8, ir) is False and c.init(8, 0, ir) is False
    assert c.init(4, 8, np.zeros(0, np.float32)) is True
    assert c.init(4, 8, np.zeros(9, np.float32)) is True
    assert np.all(c.process(np.ones(11, np.float32)) == 0)
    f = reevr_amd.FFTConvolver()
    assert np.all(f.process(np.ones(3, np.float32)) == 0)     # process before init
    assert f.init(0, ir) is False
    assert f.init(1 << 20, ir) is (lib.rvc_device_count() > 0)    # huge block: clamped, so it only fails for want of a device


def test_removed_persistent_flag_is_refused_not_ignored(lib):
    """RVC_FLAG_PERSISTENT (the resident-kernel mode of rounds 2-3) was removed: a set created with the bit reports
    RVC_ERR_UNSUPPORTED at once and every init on it fails with the same code -- with or without a device."""
    from reevr_amd import _lib
    h = lib.rvc_set_create(2, 0, _lib.RVC_FLAG_PERSISTENT)
    assert h
    try:
        assert lib.rvc_last_error(h) == _lib.RVC_ERR_UNSUPPORTED
        ir = np.ones(64, np.float32)
        irs = (_lib.F32P * 2)(ir.ctypes.data_as(_lib.F32P), ir.ctypes.data_as(_lib.F32P))
        lens = (ctypes.c_size_t * 2)(64, 64)
        assert lib.rvc_set_init(h, 512, 8192, irs, lens, 512) == 0
        assert lib.rvc_last_error(h) == _lib.RVC_ERR_UNSUPPORTED
        assert b"RVC_FLAG_PERSISTENT" in lib.rvc_last_error_string(h)
    finally:
        lib.rvc_set_destroy(h)


def test_cpp_shim_headers_compile():
    """The C++ drop-in classes (include/reevr_amd/Convolver.h, StereoConvolver.h) compile and
    link against the C ABI with plain g++ (no HIP headers needed on the host side)."""
    import shutil
    import subprocess
    import tempfile
    from reevr_amd import _lib, build
    build.build_lib()
    src = os.path.join(ROOT, "tests", "shim_smoke.cpp")
    if not os.path.exists(src) or shutil.which("g++") is None:
        pytest.skip("shim smoke source or g++ missing")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "shim_smoke")
        libdir = os.path.dirname(_lib.LIB_PATH)
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                        "-L", libdir, "-lreevr_amd", f"-Wl,-rpath,{libdir}"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
