#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of the gfx950 code object inside a HIP shared library.

    python scripts/kernel_resources.py [lib.so] [name-filter]

Pulls the `.hip_fatbin` section out with objcopy, unbundles the gfx950 entry of the clang offload bundle (decompressing a
CCOB wrapper with the bundler itself when present) and prints the AMDGPU metadata notes (llvm-readelf --notes)."""

import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def extract_code_objects(lib: str, out: str) -> list:
    """Every gfx950 code object of the library (one per translation unit: the `.hip_fatbin` section holds their offload bundles
    one after the other) -> files out, out.1, out.2, ...; returns the paths."""
    paths = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        starts = sorted(m.start() for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__|CCOB", data))
        # a bundle magic inside another bundle's payload would be a false start: keep only starts at or past the end of the previous one
        end = 0
        for i, st in enumerate(starts):
            if st < end:
                continue
            dst = out if not paths else f"{out}.{len(paths)}"
            chunk = data[st:]
            if chunk[:4] == b"CCOB":
                # compressed bundle: let the bundler unpack it (header: magic, version, method, total size ...)
                one = os.path.join(td, f"b{i}.bin")
                nxt = next((s2 for s2 in starts if s2 > st), len(data))
                open(one, "wb").write(data[st:nxt])
                subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={one}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={dst}"], check=True)  # fmt: skip
                paths.append(dst)
                end = nxt
                continue
            (n,) = struct.unpack_from("<Q", chunk, 24)
            pos, found = 32, False
            for _ in range(n):
                off, size, tl = struct.unpack_from("<QQQ", chunk, pos)
                triple = chunk[pos + 24 : pos + 24 + tl].decode()
                pos += 24 + tl
                end = max(end, st + off + size)
                if "gfx950" in triple:
                    open(dst, "wb").write(chunk[off : off + size])
                    found = True
            if found:
                paths.append(dst)
    if not paths:
        raise SystemExit("no gfx950 entry in the library")
    return paths


def extract_code_object(lib: str, out: str) -> None:
    """the first code object only (kept for callers that look at one translation unit)"""
    extract_code_objects(lib, out)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(here, "..", "cppflow_amd", "csrc", "libcppflow_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    co = os.environ.get("CPPF_CODE_OBJECT_OUT", "/tmp/cppflow_gfx950.co")
    notes = ""
    for path in extract_code_objects(lib, co):
        notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", path], capture_output=True, text=True, check=True).stdout
    demangle = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
    for blk in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        get = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
        dn = demangle(name)
        if flt and flt not in dn:
            continue
        print(f"vgpr {get('vgpr_count'):4d} sgpr {get('sgpr_count'):4d} lds {get('group_segment_fixed_size'):6d} "
              f"scratch {get('private_segment_fixed_size'):5d} spill {get('vgpr_spill_count'):3d}  {dn[:150]}")  # fmt: skip


if __name__ == "__main__":
    main()
