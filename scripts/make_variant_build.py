#!/usr/bin/env python3
"""An A/B build of the library that differs from the in-tree one in preprocessor defines of ONE translation unit: by default the
headline unit (csrc/fused_static.hip: the row-shape fused kernel of the shipped robots, ~1 minute), with --unit cppflow_hip.hip the
rest of the library (~3.5 minutes); that unit is recompiled with the extra defines and linked with the in-tree object of the other.

    python scripts/make_variant_build.py build_var/lib_canon.so -DCPPF_LEAD_SINCOS=0
    python scripts/make_variant_build.py --unit cppflow_hip.hip build_var/lib_x.so '-DCPPF_PCR_FENCE()=((void)0)'

Use with CPPFLOW_HIP_LIB=<that file> (scripts/lib_ab.sh alternates libraries on one box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cppflow_amd import build  # noqa: E402

argv = sys.argv[1:]
src = "fused_static.hip"
if argv[0] == "--unit":
    src, argv = argv[1], argv[2:]
out, defines = argv[0], argv[1:]
build.build()  # the in-tree library (and its objects) must be current
objdir = os.path.join(ROOT, "build_var", "obj_variant")
os.makedirs(objdir, exist_ok=True)
obj = os.path.join(objdir, os.path.basename(out) + "." + src.replace(".hip", ".o"))
cmd = ([os.environ.get("HIPCC", "hipcc")] + build.HIPCC_FLAGS + build.EXTRA_FLAGS.get(src, []) + defines +
       [f'-DCPPF_BUILD_ID="{build.source_hash()}"', "-c", "-o", obj, os.path.join(build.CSRC, src)])
print(" ".join(cmd))
subprocess.run(cmd, check=True, cwd=build.CSRC)
others = [os.path.join(build.CSRC, "obj", u.replace(".hip", ".o")) for u in build.SOURCES if u != src]
cmd = [os.environ.get("HIPCC", "hipcc")] + build.link_flags() + ["-o", out] + others + [obj]
print(" ".join(cmd))
subprocess.run(cmd, check=True)
print(out)
