"""The C-ABI library on a box without a GPU: it loads, exports every symbol the header
declares, its structs match the ctypes mirror, and it fails cleanly (no compute, no crash)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rslf_hip.h")


@pytest.fixture(scope="module")
def L():
    from remotesensingproject_amd import _lib
    return _lib.lib()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rslf_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(L):
    from remotesensingproject_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "librslf_hip.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names, "binding and header disagree on the symbol list"


def test_abi_version_and_strings(L):
    from remotesensingproject_amd import _lib
    assert L.rslf_abi_version() == _lib.ABI_VERSION == 6
    assert L.rslf_status_string(0) == b"ok"
    assert b"invalid" in L.rslf_status_string(-1)


def test_default_params_match_reference_defaults(L, oracle_mod):
    """core.hpp:16-31, :74-99 -- and the oracle's mirror of the same defaults."""
    from remotesensingproject_amd import _lib
    from remotesensingproject_amd.depth import Depth1DParameters
    p = _lib.default_params()
    o = oracle_mod.default_params()
    assert p.edge_score_threshold == C.c_float(0.02).value
    assert p.raw_score_threshold == 0.0
    assert p.mean_shift_max_iter == 10.0
    assert p.edge_confidence_filter_size == 9 and p.median_filter_size == 5
    assert p.median_filter_epsilon == C.c_float(0.1).value
    assert p.slope_factor == 1.0 and p.cut_shadows == 1
    assert p.shadow_level == C.c_float(0.05 * 1.73205080757).value
    assert p.kernel_bandwidth == C.c_float(0.2).value
    assert p.interpolation == 0 == o.interpolation   # Interpolation1DLinear, core.hpp:76
    for f in ("edge_score_threshold", "raw_score_threshold", "mean_shift_max_iter", "edge_confidence_filter_size",
              "median_filter_size", "median_filter_epsilon", "slope_factor", "cut_shadows", "shadow_level", "kernel_bandwidth"):
        assert getattr(p, f) == getattr(o, f), f
    q = Depth1DParameters().to_c()
    for f, _ in _lib.RslfParams._fields_:
        assert getattr(p, f) == getattr(q, f), f


def test_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof as gcc sees include/rslf_hip.h == the ctypes mirror."""
    from remotesensingproject_amd import _lib
    src = tmp_path / "layout.c"
    fields_p = [f for f, _ in _lib.RslfParams._fields_]
    fields_d = [f for f, _ in _lib.RslfVolumeDesc._fields_]
    fields_s = [f for f, _ in _lib.RslfStats._fields_]
    body = ['#include <stdio.h>', '#include <stddef.h>', '#include "rslf_hip.h"', 'int main(void){',
            'printf("%zu %zu %zu\\n", sizeof(rslf_params), sizeof(rslf_volume_desc), sizeof(rslf_stats));']
    for f in fields_p:
        body.append('printf("%%zu\\n", offsetof(rslf_params, %s));' % f)
    for f in fields_d:
        body.append('printf("%%zu\\n", offsetof(rslf_volume_desc, %s));' % f)
    for f in fields_s:
        body.append('printf("%%zu\\n", offsetof(rslf_stats, %s));' % f)
    body.append("return 0;}")
    src.write_text("\n".join(body))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    sizes = list(map(int, out[:3]))
    assert sizes == [C.sizeof(_lib.RslfParams), C.sizeof(_lib.RslfVolumeDesc), C.sizeof(_lib.RslfStats)]
    offs = list(map(int, out[3:]))
    want = [getattr(_lib.RslfParams, f).offset for f in fields_p] + \
           [getattr(_lib.RslfVolumeDesc, f).offset for f in fields_d] + \
           [getattr(_lib.RslfStats, f).offset for f in fields_s]
    assert offs == want


def test_header_is_plain_c(tmp_path):
    """The boundary must be consumable from C and from the reference's C++11."""
    for comp, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "cpp")):
        src = tmp_path / ("inc." + ext)
        src.write_text('#include "rslf_hip.h"\nint main(void){ rslf_params p; (void)p; return 0; }\n')
        subprocess.run([comp, std, "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                        "-o", str(tmp_path / ("inc_%s.o" % ext))], check=True)


def test_no_device_is_an_error_not_a_fallback(L):
    """On a box without a GPU the product must refuse, loudly -- never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = L.rslf_ctx_create(0, C.byref(h))
    assert rc in (-4, -3), rc
    assert not h.value
    assert L.rslf_last_error()
    # NULL handles are rejected, not dereferenced
    assert L.rslf_volume_create(None, 1, 1, 1, 1, C.byref(h)) == -1
    assert L.rslf_ctx_synchronize(None) == -1


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under the package or include/ may reference it."""
    bad = []
    for base in (os.path.join(ROOT, "remotesensingproject_amd"), os.path.join(ROOT, "include")):
        for dp, _, fs in os.walk(base):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M) or "rslf_oracle" in txt or "liboracle" in txt:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_opencv_block_compiles():
    """The cv::Mat constructors and getters of include/rslf_hip.hpp (north_star's literal 'cv::Mat in / cv::Mat out') are
    compiled -- syntax and types only -- against a declaration-only mock of the few cv::Mat members they touch: this image
    has no OpenCV, and without this the block had never met a compiler."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I", os.path.join(root, "include"),
                        "-I", os.path.join(root, "tests", "cpp", "opencv_mock"), os.path.join(root, "tests", "cpp", "opencv_block.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
