#!/usr/bin/env python3
"""Diagnostic build for scripts/dp_step_timeline.py: a patched COPY of csrc/kernels_dp.h + csrc/cppflow_hip.hip in which the resident
dp_search kernel of 65 .. 256 candidates (dp_persistent4_kernel<D, 256>) stamps the 100 MHz s_memrealtime counter (one clock for the
whole chip) at four points of every step -- cost-independent part done / this wavefront's cost words arrived / workgroup barrier
passed / this wavefront's destination published -- into a device array, and an extra entry point copies the array out.  That unit is
compiled alone (~3.5 minutes) and linked with the in-tree object of the fused kernel into build_var/lib_dpstamp.so; never shipped."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cppflow_amd import build  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build_var", "lib_dpstamp.so")
build.build()
tree = tempfile.mkdtemp(prefix="cppf_dpstamp_")
os.makedirs(os.path.join(tree, "cppflow_amd"))
shutil.copytree(build.CSRC, os.path.join(tree, "cppflow_amd", "csrc"), ignore=shutil.ignore_patterns("*.so", "obj"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tree, "include"))
csrc = os.path.join(tree, "cppflow_amd", "csrc")


def patch(name, pairs):
    p = os.path.join(csrc, name)
    s = open(p).read()
    for old, new in pairs:
        assert s.count(old) == 1, (name, old[:80], s.count(old))
        s = s.replace(old, new)
    open(p, "w").write(s)


STAMP = "__builtin_amdgcn_s_memrealtime()"
patch("kernels_dp.h", [
    ("template <int D, int SRC, int NS = 1>\n__global__ __launch_bounds__(2 * SRC) void dp_persistent4_kernel(",
     "// [step][workgroup][16]: 0 cost-independent part done (wave 0), 1..8 cost words of wave w arrived, 9 barrier passed (wave 0), 10..13 destination i published\n"
     "__device__ unsigned long long g_dp_stamps[256 * 64 * 16];\n"
     "template <int D, int SRC, int NS = 1>\n__global__ __launch_bounds__(2 * SRC) void dp_persistent4_kernel("),
    ("        unsigned long long (*img)[SRC] = keys[t & 1];\n        {\n            // lanes / sources beyond k carry (+inf, 0), like the idle lanes of the per-waypoint kernel",
     "        unsigned long long* const stamps = g_dp_stamps + ((size_t)(t & 255) * 64 + (blockIdx.x & 63)) * 16;\n"
     f"        if (tid == 0) stamps[0] = {STAMP};\n"
     "        unsigned long long (*img)[SRC] = keys[t & 1];\n        {\n            // lanes / sources beyond k carry (+inf, 0), like the idle lanes of the per-waypoint kernel"),
    ("            for (int u = 0; u < 2; ++u) img[2 * h + u][a] = best[u];\n        }\n        __syncthreads();\n        if (wave < BP) {\n            const int i = wave;\n            unsigned long long key = img[i][lane];\n#pragma unroll\n            for (int w = 1; w < SRC / 64; ++w) {",
     "            for (int u = 0; u < 2; ++u) img[2 * h + u][a] = best[u];\n        }\n"
     f"        if (lane == 0 && wave < 8) stamps[1 + wave] = {STAMP};\n"
     "        __syncthreads();\n"
     f"        if (tid == 0) stamps[9] = {STAMP};\n"
     "        if (wave < BP) {\n            const int i = wave;\n            unsigned long long key = img[i][lane];\n#pragma unroll\n            for (int w = 1; w < SRC / 64; ++w) {"),
    ("                memoT[(size_t)t * k + b0 + i] = (int32_t)(uint32_t)key;  // read only by the back-trace launch\n                dp_publish_cost(costsT + (size_t)t * k + b0 + i, dp_key_value(key));\n            }\n        }\n    }\n}\n\n// The resident form for 257 .. 1024 candidates",
     "                memoT[(size_t)t * k + b0 + i] = (int32_t)(uint32_t)key;  // read only by the back-trace launch\n                dp_publish_cost(costsT + (size_t)t * k + b0 + i, dp_key_value(key));\n"
     f"                stamps[10 + i] = {STAMP};\n"
     "            }\n        }\n    }\n}\n\n// The resident form for 257 .. 1024 candidates"),
])
patch("cppflow_hip.hip", [
    ("const char* cppf_build_id(void) { return kBuildIdMarker + 14; }",
     "const char* cppf_build_id(void) { return kBuildIdMarker + 14; }\n"
     "int cppf_debug_dp_stamps(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dp_stamps), bytes); }"),
])
src = "cppflow_hip.hip"
objdir = os.path.join(ROOT, "build_var", "obj_variant")
os.makedirs(objdir, exist_ok=True)
obj = os.path.join(objdir, os.path.basename(out_path) + ".cppflow_hip.o")
cmd = ([os.environ.get("HIPCC", "hipcc")] + build.HIPCC_FLAGS + build.EXTRA_FLAGS.get(src, []) +
       [f'-DCPPF_BUILD_ID="{build.source_hash()}"', "-c", "-o", obj, os.path.join(csrc, src)])
print(" ".join(cmd))
subprocess.run(cmd, check=True, cwd=csrc)
others = [os.path.join(build.CSRC, "obj", u.replace(".hip", ".o")) for u in build.SOURCES if u != src]
subprocess.run([os.environ.get("HIPCC", "hipcc")] + build.link_flags() + ["-o", out_path] + others + [obj], check=True)
shutil.rmtree(tree)
print(out_path)
