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
DEBUG_HEADER = os.path.join(ROOT, "include", "reevr_amd", "rvc_debug.h")     # measurement / development entries


@pytest.fixture(scope="module")
def lib():
    from reevr_amd import build
    build.build_lib()
    from reevr_amd import _lib
    return _lib.lib()


def declared_symbols(header=None):
    text = open(header).read() if header else open(HEADER).read() + open(DEBUG_HEADER).read()
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
    # rvc.h is the reference's surface plus the set API: no measurement / development entry in it
    dbg = lambda n: n.startswith("rvc_debug_") or n.endswith("_timed") or n.endswith("_tuned") or n.endswith("_stamped")
    assert not [n for n in declared_symbols(HEADER) if dbg(n)]
    assert all(dbg(n) for n in declared_symbols(DEBUG_HEADER))


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


def test_tuning_knobs_and_context_manager(lib):
    """Every knob reevr_amd.tuning can restore is known to the library (rvc_debug_set_tuning returns 1), unknown ones are
    refused on both sides, and the context manager puts the defaults back when its block raises."""
    import reevr_amd
    from reevr_amd.convolver import TUNING_DEFAULTS
    for k, v in TUNING_DEFAULTS.items():
        assert reevr_amd.set_tuning(k, v), k
    assert not reevr_amd.set_tuning("no_such_knob", 1)
    with pytest.raises(KeyError):
        reevr_amd.tuning(no_such_knob=1)
    with pytest.raises(RuntimeError):
        with reevr_amd.tuning(subsets=4, sweep_lds=0):
            raise RuntimeError("boom")


def test_plan_and_per_set_knobs_without_gpu(lib):
    """rvc_set_plan before any init (all zero but the channel count), after an init with empty impulses (initialised, not live);
    rvc_set_create_tuned takes knobs of the set's own and refuses unknown keys / malformed items; the knob defaults come from
    the library; the ABI number is the header's."""
    import reevr_amd
    from reevr_amd import _lib
    assert lib.rvc_abi_version() == _lib.RVC_ABI_VERSION == 2
    hdr = open(HEADER).read()
    assert "#define RVC_ABI_VERSION 2" in hdr
    s = reevr_amd.ConvolverSet(6)
    p = s.plan()
    assert p["channels"] == 6 and p["subsets"] == 1 and not p["initialised"] and not p["live"] and p["head_block"] == 0
    assert s.init(64, 256, [np.zeros(5, np.float32)] * 6)
    p = s.plan()
    assert p["initialised"] == 1 and p["live"] == 0 and (p["head_block"], p["tail_block"]) == (64, 256)
    s.close()
    # the struct the header declares and the ctypes mirror have the same fields in the same order
    body = re.sub(r"/\*.*?\*/", "", hdr[hdr.index("typedef struct rvc_plan {"):hdr.index("} rvc_plan;")], flags=re.S)
    fields = [n for decl in re.findall(r"(?:int|size_t)\s+([a-z_0-9, ]+);", body) for n in re.split(r",\s*", decl.strip())]
    assert fields == [n for n, _ in _lib.Plan._fields_], fields
    assert lib.rvc_set_plan(None, None, 0) == 0
    # a caller compiled against a shorter (earlier) struct gets the fields it knows; a longer buffer is zero-filled behind the struct
    short = (ctypes.c_byte * _lib.Plan.head_partitions.offset)()
    h = lib.rvc_set_create(3, 0, 0)
    assert lib.rvc_set_plan(h, ctypes.byref(short), ctypes.sizeof(short)) == 1
    assert ctypes.cast(short, ctypes.POINTER(ctypes.c_int))[0] == 3
    long_buf = (ctypes.c_byte * (ctypes.sizeof(_lib.Plan) + 64))(*([0x55] * (ctypes.sizeof(_lib.Plan) + 64)))
    assert lib.rvc_set_plan(h, ctypes.byref(long_buf), ctypes.sizeof(long_buf)) == 1 and not any(long_buf[ctypes.sizeof(_lib.Plan):])
    assert lib.rvc_set_plan(h, ctypes.byref(short), 4) == 0
    lib.rvc_set_destroy(h)
    # knobs
    keys = lib.rvc_debug_tuning_keys().decode().split(",")
    assert "k1" in keys and "sweep_lds" in keys and "mac3" in keys and sorted(keys) == sorted(reevr_amd.TUNING_DEFAULTS.keys())
    assert reevr_amd.TUNING_DEFAULTS["subsets"] == -1 and reevr_amd.TUNING_DEFAULTS["kid_fence"] == 1
    t = reevr_amd.ConvolverSet(4, tune={"k1": 32, "subsets": 2})
    t.close()
    for bad in (b"no_such=1", b"k1", b"k1=", b"=3", b"k1=3x",
                #  only plain decimal ints inside int range: no sign prefix but '-', no blanks, no other bases, no overflow
                b"k1=+5", b"k1= 7", b"k1=0x10", b"subsets=99999999999", b"subsets=-99999999999", b"k1=-", b"k1=--1"):
        assert not lib.rvc_set_create_tuned(2, 0, 0, bad), bad
    for good in (b"k1=-1", b"subsets=2147483647", b"tail_phases=8,tail_spread=0,host_zero_copy=-1,block_lanex=1,kid_stagger=0"):
        h = lib.rvc_set_create_tuned(2, 0, 0, good)
        assert h, good
        lib.rvc_set_destroy(h)
    # the staging rows of a set without device state: nothing to hand out (rvc_set_host_buffers returns 0, the arrays hold NULL)
    h = lib.rvc_set_create(3, 0, 0)
    ins, outs = (ctypes.c_void_p * 3)(1, 1, 1), (ctypes.c_void_p * 3)(1, 1, 1)
    assert lib.rvc_set_host_buffers(h, ins, outs) == 0 and not any(ins) and not any(outs)
    assert lib.rvc_set_host_buffers(None, ins, outs) == 0
    p = _lib.Plan()
    assert lib.rvc_set_plan(h, ctypes.byref(p), ctypes.sizeof(p)) == 1 and p.tail_phase_groups == 0 and p.tail_spread == 0
    lib.rvc_set_destroy(h)
    h = lib.rvc_set_create_tuned(2, 0, 0, None)
    assert h
    lib.rvc_set_destroy(h)
    # rvc_debug_set_tuning changes the DEFAULT of later sets only, and the shipped value stays readable
    assert reevr_amd.set_tuning("k1", 8)
    try:
        v = ctypes.c_int(-5)
        assert lib.rvc_debug_tuning_default(b"k1", ctypes.byref(v)) == 1 and v.value == 0
        assert lib.rvc_debug_tuning_default(b"nope", ctypes.byref(v)) == 0
    finally:
        reevr_amd.set_tuning("k1", 0)


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
        # the plain-C++ host examples build against the same headers and library (host_many_channels: rvc_set_host_buffers, rvc_set_plan)
        for ex in ("host_block_loop", "host_many_channels"):
            subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", ex + ".cpp"),
                            "-o", os.path.join(d, ex), "-L", libdir, "-lreevr_amd", f"-Wl,-rpath,{libdir}"], check=True)
        r = subprocess.run([os.path.join(d, "host_many_channels"), "4", "2", "1"], capture_output=True, text=True, timeout=120)
        assert r.returncode in (0, 2), r.stdout + r.stderr          # (2: no GPU here -- "this engine has no CPU fallback")


from oracle.ref_glue import GLUE_EXE, REF_SC, build_reference_glue  # noqa: E402  (the recipe lives with the checker, not with the tests)


def glue_operands(path, block, nblocks, ir_len, quad):
    from reevr_amd import synth
    irs = synth.synth_ir(ir_len, 4, 3)
    x = np.stack([synth.synth_input(block * nblocks, 70 + c) for c in range(2)])
    with open(path, "wb") as f:
        np.array([block, nblocks, ir_len, int(quad)], np.int32).tofile(f)
        irs.astype(np.float32).tofile(f)          # LL RR LR RL
        x.astype(np.float32).tofile(f)
    return irs, x


def test_reference_stereo_convolver_compiles_against_the_shim(tmp_path):
    """CPU: the reference's own StereoConvolver.cpp compiles and links against the drop-in Convolver (C ABI underneath); without
    a device every init fails loudly and the outputs are zeros -- no CPU fallback behind the reference's glue either."""
    import shutil
    import subprocess
    if not os.path.exists(REF_SC) or shutil.which("g++") is None:
        pytest.skip("/root/reference or g++ absent")
    exe = build_reference_glue(str(tmp_path))
    from reevr_amd import _lib
    if _lib.lib().rvc_device_count() > 0:
        pytest.skip("a GPU is present: the -m gpu test runs the binary against the oracle")
    pin, pout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    glue_operands(pin, 64, 6, 500, True)
    r = subprocess.run([exe, pin, pout], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    out = np.fromfile(pout, np.float32)
    assert out.size == 6 * 4 * 64 and np.all(out == 0)
