"""Compile librslf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m remotesensingproject_amd.build [--force] [--report]

The flags are part of the numerics contract: -ffp-contract=off (no FMA
contraction) and no fast-math, so every device float op is the single IEEE
binary32 operation the reference performs ("Numerics").
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
SO = os.path.join(CSRC, "librslf_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
COMMON = ["rslf_internal.hpp", "rslf_plan.hpp", "rslf_device.hpp", os.path.join(INCLUDE, "rslf_hip.h")]
# translation unit -> the kernel headers it alone includes (device code is per unit; rslf_internal.hpp lists the units)
UNITS = {
    "rslf_core.hip": ["k0_pack.hpp"],
    "rslf_pile.hip": ["k1_edge.hpp", "k_compact.hpp", "k2_scan.hpp", "k2_reg.hpp", "k2_stream.hpp", "k3_median.hpp"],
    "rslf_chip_a.hip": ["k2_scan.hpp", "k2_chip.hpp"],
    "rslf_chip_b.hip": ["k2_scan.hpp", "k2_chip.hpp"],
    "rslf_chip_c.hip": ["k2_scan.hpp", "k2_chip.hpp"],
    "rslf_sweep.hip": ["k3_median.hpp", "k4_propagate.hpp", "k_compact.hpp"],
    "rslf_f2c.hip": ["k5_f2c.hpp"],
    "rslf_multi.hip": [],
    "rslf_multi_sweep.hip": [],
}
SOURCES = list(UNITS)

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    # packed fp32 (v_pk_*) issues at half the rate of the scalar forms on gfx950
    # (tools/ubench_valu.hip), so SLP packing only adds v_mov shuffles
    "-fno-slp-vectorize",
]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps(unit: str) -> list[str]:
    out = []
    for d in [unit] + UNITS[unit] + COMMON:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.exists(p):
            out.append(p)
    return out


def _obj(unit: str) -> str:
    return os.path.join(OBJ, unit.replace(".hip", ".o"))


def _stale(unit: str) -> bool:
    o = _obj(unit)
    return not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in _deps(unit))


def needs_build() -> bool:
    return not os.path.exists(SO) or any(_stale(u) or os.path.getmtime(_obj(u)) > os.path.getmtime(SO) for u in SOURCES)


def source_hash() -> str:
    """sha256 over the sources the library is built from (sorted by name): stamps profiles/k2_traffic.json entries, so a
    counter figure measured on other code is never quoted beside today's kernel time (bench.py)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(set(p for u in SOURCES for p in _deps(u)))
    for p in files:
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def build_variant(name: str, defines: list[str]) -> str:
    """A/B builds (developer tool): the library with extra -D flags, as ab/librslf_<name>.so (git-ignored; it travels to
    the GPU box).  EVERY translation unit is compiled with the defines (the first version passed them to the scan unit
    only: an A/B of a macro that lives in the sweep's kernels then compared a build with itself)."""
    import concurrent.futures
    out_dir = os.path.join(os.path.dirname(HERE), "ab")
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)

    def compile_unit(u: str) -> str:
        obj = os.path.join(OBJ, "%s_%s.o" % (os.path.splitext(u)[0], name))
        r = subprocess.run([_hipcc()] + HIPCC_FLAGS + ["-D" + d for d in defines] + ["-I", INCLUDE, "-c", u, "-o", obj],
                           cwd=CSRC, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (u, r.stderr[-4000:]))
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_unit, SOURCES))
    so = os.path.join(out_dir, "librslf_%s.so" % name)
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return so


def build(force: bool = False, report: bool = False) -> str:
    """Build the shared library in-tree (one object per translation unit, compiled in parallel); returns its path."""
    if not force and not report and not needs_build():
        return SO
    os.makedirs(OBJ, exist_ok=True)
    todo = [u for u in SOURCES if force or report or _stale(u)]
    logs: dict[str, str] = {}

    def compile_unit(u: str) -> None:
        cmd = [_hipcc()] + HIPCC_FLAGS + ["-I", INCLUDE, "-c", u, "-o", _obj(u)]
        if report:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (u, r.stderr[-4000:]))
        logs[u] = r.stderr

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_unit, todo))
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + [_obj(u) for u in SOURCES],
                       cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    if report:
        print("# source hash %s" % source_hash())
        print(resource_table("\n".join(logs[u] for u in SOURCES)))
    return SO


def resource_table(stderr: str) -> str:
    rows, cur = [], {}
    for line in stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    out = ["%-64s %5s %5s %8s %4s %6s" % ("kernel", "vgpr", "sgpr", "scratch", "occ", "lds")]
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
        name = re.sub(r"\(.*", "", name).replace("void rslf::", "")
        out.append("%-64s %5d %5d %8d %4d %6d" % (name[:64], r.get("vgpr", -1), r.get("sgpr", -1), r.get("scratch", -1),
                                                   r.get("occ", -1), r.get("lds", -1)))
    return "\n".join(out)


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m remotesensingproject_amd.build --variant NAME -DFOO=1 ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a[2:] for a in sys.argv[i + 2:] if a.startswith("-D")]))
        sys.exit(0)
    p = build(force="--force" in sys.argv, report="--report" in sys.argv)
    print(p)
