#!/usr/bin/env python3
"""Scratch (private segment) bytes and spilled registers of every kernel in a gfx950 assembly listing or in libseggroup_hip.so.

    python3 tools/scratch_report.py [file.s | lib.so]        (default: seggroup_amd/libseggroup_hip.so)

Why it matters: a spilled register is HBM traffic per lane and tile (round 5: the MLP3 EdgeConv's 17 MB of writes per scene-launch were
spills); tests/test_build.py keeps every kernel of the inference path at zero.
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of_notes(text):
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count", text):
        n = re.search(r"\.name:\s+(\S+)", blk)
        p = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        v = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        s = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
        if n and p:
            out.append((n.group(1), int(p.group(1)), int(v.group(1)) if v else -1, int(s.group(1)) if s else -1))
    return out


def kernels_of_library(path):
    """(mangled name, scratch bytes, VGPRs, spilled VGPRs) of every kernel in the library's gfx950 code objects"""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        with open(path, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, capture_output=True, check=True)
        for name in sorted(os.listdir(tmp)):
            if "gfx950" not in name:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, name)], capture_output=True, text=True).stdout
            out += kernels_of_notes(notes)
    return out


def packed_fp32_of_library(path):
    """{mangled kernel name: number of v_pk_add / mul / fma_f32 instructions} over the library's gfx950 code objects (kernels with none are left out)"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        with open(path, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, capture_output=True, check=True)
        for name in sorted(os.listdir(tmp)):
            if "gfx950" not in name:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", os.path.join(tmp, name)], capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    cur = m.group(1)
                elif cur and re.search(r"\bv_pk_(add|mul|fma)_f32\b", line):
                    out[cur] = out.get(cur, 0) + 1
    return out


def demangle(names):
    for tool in (f"{LLVM}/llvm-cxxfilt", "c++filt"):
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
        except FileNotFoundError:
            continue
        if r.returncode == 0:
            return r.stdout.splitlines()
    return names


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "seggroup_amd", "libseggroup_hip.so")
    ks = kernels_of_notes(open(path).read()) if path.endswith(".s") else kernels_of_library(path)
    names = demangle([k[0] for k in ks])
    bad = 0
    for (raw, scratch, vgpr, spill), name in zip(ks, names):
        if scratch:
            bad += 1
            print(f"{scratch:5d} B scratch  {spill:3d} spilled  {vgpr:3d} VGPRs  {name[:140]}")
    print(f"{len(ks)} kernels, {bad} with scratch")
    if not path.endswith(".s"):
        pk = packed_fp32_of_library(path)
        for (raw, n), name in zip(sorted(pk.items()), demangle(sorted(pk))):
            print(f"{n:5d} packed fp32 instructions  {name[:140]}")
        print(f"{len(pk)} kernels with packed fp32 arithmetic")


if __name__ == "__main__":
    main()
