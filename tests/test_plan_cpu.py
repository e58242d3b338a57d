"""Host-side planning (remotesensingproject_amd/csrc/rslf_plan.hpp) on the CPU, under AddressSanitizer + UBSan, and the
engineering contracts of the C boundary that can be checked without a GPU: every `extern "C" int rslf_*` definition is
a function-try-block closed by the library's handler list, worker threads are owned by a JoinGuard, and no source file
of the library has grown past 1 200 lines."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "remotesensingproject_amd", "csrc")


def test_plan_unit_tests_under_sanitizers(tmp_path):
    exe = tmp_path / "test_plan"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
                    "-Werror", "-I", CSRC, os.path.join(ROOT, "tests", "cpp", "test_plan.cpp"), "-o", str(exe)], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "plan tests ok" in r.stdout


def test_plan_header_is_host_only():
    """rslf_plan.hpp must stay compilable by g++ alone: no HIP include, no device qualifier."""
    txt = open(os.path.join(CSRC, "rslf_plan.hpp")).read()
    assert "hip/hip_runtime" not in txt and "__device__" not in txt and "__global__" not in txt


def _units():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def test_every_int_entry_point_is_guarded():
    """include/rslf_hip.h: 'never throws across the boundary'.  Every definition `extern "C" int rslf_*(...)` in the library
    opens with RSLF_API_TRY and its body is followed by RSLF_API_CATCH; every int-returning symbol the header declares
    has such a definition."""
    guarded, bare = set(), []
    for f in _units():
        src = open(os.path.join(CSRC, f)).read()
        for m in re.finditer(r'^extern "C" int (rslf_\w+)\(', src, flags=re.M):
            tail = src[m.start():]
            head_end = tail.index("{")
            head = tail[:head_end]
            body_end = tail.index("\n}\n")
            after = tail[body_end + 3:body_end + 3 + 20]
            if "RSLF_API_TRY" in head and after.startswith("RSLF_API_CATCH"):
                guarded.add(m.group(1))
            else:
                bare.append("%s:%s" % (f, m.group(1)))
    assert not bare, "entry points without the exception barrier: %s" % bare
    hdr = open(os.path.join(ROOT, "include", "rslf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"^\s*int\s+(rslf_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 50
    assert declared <= guarded, "declared but not guarded: %s" % sorted(declared - guarded)


def test_threads_are_owned_by_a_join_guard():
    """No bare std::thread object outside JoinGuard: a throwing emplace_back with earlier threads joinable is std::terminate
    (ADVICE r2), and so is an exception that unwinds past a joinable thread."""
    for f in _units() + ["rslf_internal.hpp"]:
        src = open(os.path.join(CSRC, f)).read()
        src = re.sub(r"//[^\n]*", "", src)
        uses = [m.start() for m in re.finditer(r"std::thread\b(?!::hardware_concurrency)", src)]
        if f == "rslf_internal.hpp":
            guard = src[src.index("class JoinGuard"):src.index("template <typename F>\nint guarded_status")]
            assert all(src.index("class JoinGuard") <= u < src.index("class JoinGuard") + len(guard) for u in uses), f
        else:
            assert not uses, "%s creates std::thread objects outside JoinGuard" % f


def test_no_library_source_exceeds_1200_lines():
    big = []
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".hpp")):
            n = sum(1 for _ in open(os.path.join(CSRC, f)))
            if n > 1200:
                big.append((f, n))
    assert not big, big


def test_status_strings_cover_the_new_codes():
    from remotesensingproject_amd import _lib
    L = _lib.lib()
    assert b"internal" in L.rslf_status_string(-6)
    assert b"allocation" in L.rslf_status_string(-5)
    assert L.rslf_debug_inject(b"no_such_site", 1) == -1
    assert L.rslf_debug_inject(b"worker", 0) == 0
