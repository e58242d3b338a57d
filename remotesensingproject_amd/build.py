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
SO = os.path.join(CSRC, "librslf_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SOURCES = ["rslf_abi.hip"]
DEPS = SOURCES + ["rslf_device.hpp", "k1_edge.hpp", "k2_scan.hpp", "k3_median.hpp", "k4_propagate.hpp", "k5_f2c.hpp", os.path.join(INCLUDE, "rslf_hip.h")]

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
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


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    for d in DEPS:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.getmtime(p) > t:
            return True
    return False


def build(force: bool = False, report: bool = False) -> str:
    """Build the shared library in-tree; returns its path."""
    if not force and not report and not needs_build():
        return SO
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-I", INCLUDE, "-o", SO] + SOURCES
    if report:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])
    if report:
        print(resource_table(r.stderr))
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
    p = build(force="--force" in sys.argv, report="--report" in sys.argv)
    print(p)
