#!/usr/bin/env python3
"""An A/B build of the library that differs from the in-tree one in preprocessor defines of the HEADLINE translation unit only
(csrc/fused_static.hip: the row-shape fused kernel of the shipped robots): that unit is recompiled with the extra defines and linked
with the in-tree object of the other unit (csrc/obj/cppflow_hip.o), i.e. ~1 minute instead of the library's 3.5.

    python scripts/make_variant_build.py build_var/lib_canon.so -DCPPF_LEAD_HW_SINCOS=0

Use with CPPFLOW_HIP_LIB=<that file> (scripts/lib_ab.sh alternates libraries on one box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cppflow_amd import build  # noqa: E402

out, defines = sys.argv[1], sys.argv[2:]
build.build()  # the in-tree library (and its objects) must be current
objdir = os.path.join(ROOT, "build_var", "obj_variant")
os.makedirs(objdir, exist_ok=True)
obj = os.path.join(objdir, os.path.basename(out) + ".fused_static.o")
src = "fused_static.hip"
cmd = ([os.environ.get("HIPCC", "hipcc")] + build.HIPCC_FLAGS + build.EXTRA_FLAGS.get(src, []) + defines +
       [f'-DCPPF_BUILD_ID="{build.source_hash()}"', "-c", "-o", obj, os.path.join(build.CSRC, src)])
print(" ".join(cmd))
subprocess.run(cmd, check=True, cwd=build.CSRC)
cmd = [os.environ.get("HIPCC", "hipcc")] + build.link_flags() + ["-o", out, os.path.join(build.CSRC, "obj", "cppflow_hip.o"), obj]
print(" ".join(cmd))
subprocess.run(cmd, check=True)
print(out)
