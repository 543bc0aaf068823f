#!/usr/bin/env python3
"""Diagnostic build for scripts/wave_timeline.py: a COPY of the library sources in which the fused kernel stamps s_memrealtime at
wavefront start, at the end of the LM loop and at wavefront end (plus HW_REG_HW_ID / XCC_ID) into the `n_iters` output, compiled to
build_var/lib_stamp.so.  Never shipped, never loaded unless CPPFLOW_HIP_LIB points at it.

    python scripts/make_stamp_build.py [--prio none|iteration|phase] [--shift 8]

--prio adds an s_setprio experiment (MI355X_MICROARCH.md "static priority"): `iteration` rotates each wavefront's priority with
its own iteration count, `phase` with the global 100 MHz counter (period 2^shift x 10 ns) so that the four wavefronts of a SIMD
hold four different priorities at any instant.  Measured: neither changes the age-ordered finish of an isolated launch."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cppflow_amd import build  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prio", default="none", choices=["none", "iteration", "phase"])
ap.add_argument("--shift", type=int, default=8)
ap.add_argument("--out", default=os.path.join(ROOT, "build_var", "lib_stamp.so"))
args = ap.parse_args()

tree = tempfile.mkdtemp(prefix="cppf_stamp_")
os.makedirs(os.path.join(tree, "cppflow_amd"))
shutil.copytree(os.path.join(ROOT, "cppflow_amd", "csrc"), os.path.join(tree, "cppflow_amd", "csrc"), ignore=shutil.ignore_patterns("*.so"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tree, "include"))
p = os.path.join(tree, "cppflow_amd", "csrc", "kernels_fused.h")
s = open(p).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, old[:70]
    s = s.replace(old, new)


rep("    const bool active = row < (size_t)prm.n;\n    float q[D];",
    "    const bool active = row < (size_t)prm.n;\n    const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();\n    float q[D];")
rep("    RowSummary rs;\n    if (active) {\n        float Rt[9], tt[3];",
    "    RowSummary rs;\n    unsigned long long stamp_mid = 0;\n    if (active) {\n        float Rt[9], tt[3];")
rep("        if (out.n_iters) out.n_iters[row_b] = iters;", "        stamp_mid = __builtin_amdgcn_s_memrealtime();\n        (void)iters;")
rep("""            block_seed_summary<RB>(rb, prm.W, (size_t)(blockIdx.x * (unsigned)kBlock + (unsigned)tid_c), active, q, rs, out.seed_summary);
        }
    }
}""", """            block_seed_summary<RB>(rb, prm.W, (size_t)(blockIdx.x * (unsigned)kBlock + (unsigned)tid_c), active, q, rs, out.seed_summary);
        }
    }
    if (out.n_iters && active) {  // lane 0: start | end << 16 (10 ns ticks mod 65536), lane 1: start | loop end << 16, lanes 2, 3: HW_ID, XCC_ID
        const unsigned long long stamp1 = __builtin_amdgcn_s_memrealtime();
        int tid_d = threadIdx.x;
        asm volatile("" : "+v"(tid_d));
        const size_t row_d = (size_t)(blockIdx.x * (unsigned)kBlock + (unsigned)tid_d);
        const unsigned long long e = ((tid_d & 63) == 1) ? stamp_mid : stamp1;
        int word = (int)((stamp0 & 0xFFFFull) | ((e & 0xFFFFull) << 16));
        if ((tid_d & 63) == 2) word = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        if ((tid_d & 63) == 3) word = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);
        out.n_iters[row_d] = word;
    }
}""")
if args.prio != "none":
    rep("// ---- one row of the fused launch, in three pieces", """__device__ __forceinline__ void rotate_prio(int k) {
    switch (k & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// ---- one row of the fused launch, in three pieces""")
    key = "it" if args.prio == "iteration" else f"(int)((__builtin_amdgcn_s_memrealtime() >> {args.shift}) & 3)"
    rep("        for (int it = 0; it < prm.n_steps; ++it) {\n            const bool conv = lm_row_iterate",
        f"        const int age_rank = (int)(blockIdx.x >> 8);\n        for (int it = 0; it < prm.n_steps; ++it) {{\n            rotate_prio(age_rank + {key});\n            const bool conv = lm_row_iterate")
open(p, "w").write(s)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
# (one hipcc call over both translation units; the diagnostic build does not need fused_static.hip's own scheduler flags)
cmd = ["hipcc"] + build.HIPCC_FLAGS + ["-shared", '-DCPPF_BUILD_ID="stamp"', "-o", args.out] + [os.path.join(tree, "cppflow_amd", "csrc", f) for f in build.SOURCES]
print(" ".join(cmd))
subprocess.check_call(cmd)
shutil.rmtree(tree)
print(args.out)
