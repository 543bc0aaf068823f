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


def extract_code_object(lib: str, out: str) -> None:
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        if data[:4] == b"CCOB":
            # compressed bundle: let the bundler unpack it
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], check=True)  # fmt: skip
            return
        assert data[:24] == b"__CLANG_OFFLOAD_BUNDLE__", data[:24]
        (n,) = struct.unpack_from("<Q", data, 24)
        pos = 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24 : pos + 24 + tl].decode()
            pos += 24 + tl
            if "gfx950" in triple:
                open(out, "wb").write(data[off : off + size])
                return
        raise SystemExit("no gfx950 entry in the bundle")


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(here, "..", "cppflow_amd", "csrc", "libcppflow_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    co = os.environ.get("CPPF_CODE_OBJECT_OUT", "/tmp/cppflow_gfx950.co")
    extract_code_object(lib, co)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
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
