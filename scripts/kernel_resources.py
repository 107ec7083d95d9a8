"""Per-kernel resource usage of the gfx950 code objects inside a built library (VGPRs, SGPRs, scratch, spills, LDS), read from the
AMDGPU metadata notes -- no GPU needed.

    python scripts/kernel_resources.py [lib.so] [--grep REGEX] [--json]

A HIP shared library carries one clang offload bundle per translation unit in its `.hip_fatbin` section; every bundle holds the gfx950 ELF whose
`NT_AMDGPU_METADATA` note (msgpack, printed as YAML by `llvm-readelf --notes`) lists `.vgpr_count`, `.vgpr_spill_count`,
`.private_segment_fixed_size` (scratch bytes per lane) ... per kernel.  tests/test_kernel_resources.py holds the decode kernels to
"no scratch, no spills" with this.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = os.environ.get("MRS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(HERE), "mistral.rs_amd", "lib", "libmrs_hip_ext.so")


def code_objects(lib: str):
    """yield the gfx950 ELF images bundled in `lib`"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            raise RuntimeError(f"no .hip_fatbin in {lib}: {r.stderr.strip()}")
        data = open(fat, "rb").read()
    pos = data.find(MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield data[pos + off:pos + off + size]
        pos = data.find(MAGIC, pos + 1)


_FIELDS = ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
           "group_segment_fixed_size", "max_flat_workgroup_size", "uses_dynamic_stack")


def kernels_of(elf: bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        r = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True)
    cur = None
    for line in r.stdout.splitlines():
        m = re.match(r"\s*(-\s+)?\.([a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        dash, key, val = m.groups()
        if key == "agpr_count" and dash:  # first key of a kernel entry (keys are sorted; `.agpr_count` leads, `.args` may come first)
            pass
        if dash and key in ("agpr_count", "args"):
            if cur and "name" in cur:
                yield cur
            cur = {}
        if cur is not None and key in _FIELDS:
            v = val.strip().strip("'\"")
            cur[key] = int(v) if re.fullmatch(r"-?\d+", v) else v
    if cur and "name" in cur:
        yield cur


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or os.path.join(LLVM, "llvm-cxxfilt")
    try:
        r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    except OSError:
        return names
    out = r.stdout.splitlines()
    return out if len(out) == len(names) else names


def resources(lib: str = DEFAULT_LIB):
    ks = []
    for co in code_objects(lib):
        ks.extend(kernels_of(co))
    for k, d in zip(ks, demangle([k["name"] for k in ks])):
        k["demangled"] = re.sub(r"^void ", "", d)
    return ks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib", nargs="?", default=DEFAULT_LIB)
    ap.add_argument("--grep", default=None)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    ks = resources(a.lib)
    if a.grep:
        ks = [k for k in ks if re.search(a.grep, k["demangled"])]
    if a.json:
        json.dump(ks, sys.stdout, indent=1)
        return
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'scratch':>7} {'lds':>7}  kernel")
    for k in sorted(ks, key=lambda k: k["demangled"]):
        print(f"{k.get('vgpr_count', 0):5d} {k.get('agpr_count', 0):5d} {k.get('sgpr_count', 0):5d} {k.get('vgpr_spill_count', 0):6d} "
              f"{k.get('private_segment_fixed_size', 0):7d} {k.get('group_segment_fixed_size', 0):7d}  {k['demangled'][:150]}")


if __name__ == "__main__":
    main()
