"""The C-ABI library loads and exports exactly what include/bhr.h declares.  CPU only: no
compute entry point is called (there is no GPU here); bhr_create must refuse to run without one."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    inc = os.path.join(ROOT, "include")
    text = ""
    for f in sorted(os.listdir(inc)):
        if f.endswith(".h"):
            with open(os.path.join(inc, f)) as fh:
                text += fh.read()
    return sorted(set(re.findall(r"BHR_API\s+[\w\s\*]+?\b(bhr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(hip_lib):
    from bhr_amd import _lib
    from bhr_amd.build import library_path
    declared = _declared()
    assert len(declared) >= 26
    assert sorted(_lib.SYMBOLS) == declared                  # the binding covers the whole header
    out = subprocess.check_output(["nm", "-D", "--defined-only", library_path()], text=True)
    exported = sorted(set(re.findall(r"\sT\s+(bhr_[a-z0-9_]+)", out)))
    assert exported == declared                              # nothing missing, nothing extra
    for name in declared:
        assert hasattr(hip_lib, name)


def test_version_and_struct_sizes(hip_lib, tmp_path):
    """ctypes mirrors of the structs have the sizes a C compiler gives the header's."""
    from bhr_amd import _lib
    assert hip_lib.bhr_abi_version() == 1
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "bhr.h"\nint main(void){printf("%zu %zu %zu %d\\n",'
                   'sizeof(bhr_config),sizeof(bhr_camera),sizeof(bhr_counters),BHR_TIMING_RING);return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    cfg, cam, cnt, ring = (int(v) for v in subprocess.check_output([str(exe)], text=True).split())
    assert (C.sizeof(_lib.Config), C.sizeof(_lib.Camera), C.sizeof(_lib.Counters)) == (cfg, cam, cnt)
    assert ring >= 256


def test_no_cpu_fallback_without_device(hip_lib):
    """On a box without a GPU the product refuses to create a context -- it never computes on the CPU."""
    from bhr_amd import _lib
    if hip_lib.bhr_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = _lib.Config(64, 36, 0, 36, 0.1, 10.0, 2.0, 15.0, 0.0, 0, 1.0, 0.1, 0, _lib.MATH_STRICT)
    h = C.c_void_p()
    rc = hip_lib.bhr_create(C.byref(cfg), C.byref(h))
    assert rc == _lib.BHR_ERR_NO_DEVICE and not h.value
    assert b"no HIP device" in hip_lib.bhr_last_error()
    with pytest.raises(_lib.BhrError):
        _lib.check(rc)


def test_argument_validation_needs_no_device(hip_lib):
    from bhr_amd import _lib
    h = C.c_void_p()
    bad = _lib.Config(0, 36, 0, 36, 0.1, 10.0, 2.0, 15.0, 0.0, 0, 1.0, 0.1, 0, _lib.MATH_STRICT)
    assert hip_lib.bhr_create(C.byref(bad), C.byref(h)) == _lib.BHR_ERR_INVALID
    bad = _lib.Config(64, 36, 10, 5, 0.1, 10.0, 2.0, 15.0, 0.0, 0, 1.0, 0.1, 0, _lib.MATH_STRICT)
    assert hip_lib.bhr_create(C.byref(bad), C.byref(h)) == _lib.BHR_ERR_INVALID
    bad = _lib.Config(64, 36, 0, 36, 0.1, 10.0, 5.0, 3.0, 0.0, 0, 1.0, 0.1, 0, _lib.MATH_STRICT)
    assert hip_lib.bhr_create(C.byref(bad), C.byref(h)) == _lib.BHR_ERR_INVALID
    with pytest.raises(ValueError):
        _lib.check(_lib.BHR_ERR_INVALID)


def test_product_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "black-hole-renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                with open(os.path.join(dirpath, f), errors="replace") as fh:
                    text = fh.read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"
    for f in ("render.py",):
        p = os.path.join(ROOT, f)
        if os.path.isfile(p):
            with open(p) as fh:
                assert "oracle" not in fh.read().lower()
