#!/usr/bin/env python3
"""Registers, scratch and spills of every kernel in gan-aug-pfa_amd/lib/libpfa_hip.so, read from the code objects' notes
(no GPU needed).  `python tools/kernel_resources.py [substring]` prints a table; kernel_resources() returns
{demangled-ish name: dict(vgpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds)}.

The patch kernels read LDS through inline asm into registers they name ("+v" operands): a spilled register that is the
target of an in-flight asm LDS read is saved before the data lands - silent corruption.  tests/test_abi.py therefore
asserts ScratchSize 0 for them on every build."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gan-aug-pfa_amd", "lib", "libpfa_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib=LIB):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "hipv4-amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True,
                                   stdout=subprocess.PIPE, text=True).stdout
            # one YAML list item per kernel ("  - .agpr_count: ..." opens it: the keys are in alphabetical order, so some of a
            # kernel's keys precede its .name)
            entry = None
            entries = []
            for line in notes.splitlines():
                m = re.match(r"(\s+)(- )?\.(\w+):\s+(\S+)", line)
                if not m:
                    continue
                _, opens, k, v = m.groups()
                if opens and k == "agpr_count":
                    entry = {}
                    entries.append(entry)
                if entry is None:
                    continue
                if k == "name":
                    entry.setdefault("name", v)
                elif k in ("private_segment_fixed_size", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                           "group_segment_fixed_size", "agpr_count"):
                    entry.setdefault(k, int(v))
            for e in entries:
                if e.get("name", "").startswith("_Z"):
                    out[e.pop("name")] = e
    names = list(out)
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    dem = subprocess.run([filt] + names, stdout=subprocess.PIPE, text=True).stdout.splitlines() if filt else names
    return {d.replace("(anonymous namespace)::", ""): out[n] for n, d in zip(names, dem)}


if __name__ == "__main__":
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, r in sorted(kernel_resources().items()):
        if want in name:
            print(f"{r.get('vgpr_count', 0):4d} v {r.get('agpr_count', 0):4d} a {r.get('sgpr_count', 0):4d} s  scratch {r.get('private_segment_fixed_size', 0):5d}  "
                  f"spill v{r.get('vgpr_spill_count', 0)} s{r.get('sgpr_spill_count', 0)}  lds {r.get('group_segment_fixed_size', 0):6d}  {name[:110]}")
