#!/usr/bin/env python3
"""Per-kernel metadata (registers, scratch, LDS) read from the SHIPPED librslf_hip.so -- not from a compile log: the gfx950 code
objects are unbundled from the library's .hip_fatbin section (one clang offload bundle per translation unit) and their
amdhsa notes parsed.   python tools/kernel_metadata.py [library] [substring]   prints name, vgpr, agpr, sgpr, scratch, lds."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, d):
    """The gfx950 code objects of the library (one per translation unit), unbundled into directory d."""
    fat = os.path.join(d, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    objs = []
    for i, a in enumerate(starts):
        part = os.path.join(d, "b%d.bin" % i)
        open(part, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        obj = os.path.join(d, "b%d.o" % i)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + obj], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(obj) and os.path.getsize(obj) > 0:
            objs.append(obj)
    return objs


def disassemble(lib, substring):
    """{demangled kernel name: [instruction text, ...]} for the kernels of the shipped library whose name contains
    `substring` (llvm-objdump -d of the unbundled gfx950 code objects)."""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for obj in code_objects(lib, d):
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--demangle", obj],
                                 capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    name = re.sub(r"\(.*", "", m.group(1)).replace("void ", "")
                    cur = out.setdefault(name, []) if substring in name else None
                    continue
                if cur is not None and line.startswith("\t"):
                    cur.append(line.split("//")[0].strip())
    return out


def kernels(lib):
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for obj in code_objects(lib, d):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "agpr_count" and cur.get("name"):
                    out.pop(cur["name"], None)
                    cur = {}
                if k in ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count", "name", "symbol"):
                    cur[k] = v
                if k == "vgpr_count" and "name" in cur:     # the last key of a kernel's map (keys are sorted)
                    out[cur["name"]] = dict(cur)
                    cur = {}
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r"\(.*", "", d_).replace("void ", ""): {k: int(v) for k, v in out[n].items() if k not in ("name", "symbol")} for n, d_ in zip(names, dem)}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "remotesensingproject_amd", "csrc", "librslf_hip.so")
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    ks = kernels(lib)
    for n in sorted(ks):
        if sub in n:
            k = ks[n]
            print("%-70s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d" % (n[:70], k.get("vgpr_count", -1), k.get("agpr_count", -1), k.get("sgpr_count", -1),
                                                                        k.get("private_segment_fixed_size", -1), k.get("group_segment_fixed_size", -1)))
    print(len(ks), "kernels")
