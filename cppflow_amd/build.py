"""Build libcppflow_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  `python -m cppflow_amd.build`."""

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["cppflow_hip.hip"]
HEADERS = ["lmik_device.h", "robots_gen.h", "kernels_chain.h", "kernels_collision.h", "kernels_fused.h", "kernels_quad.h", "kernels_eval.h",
           "kernels_coupled.h", "kernels_dp.h", os.path.join("..", "..", "include", "cppflow_hip.h")]  # fmt: skip
OUT = os.path.join(CSRC, "libcppflow_hip.so")

# -ffp-contract=off: the only fused multiply-adds are the explicit fmaf() of the canonical operation order, so FK and the
# capsule distances agree bit for bit with the fp32 CPU oracle; correctly rounded fp32 divide / sqrt for the same reason.
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    # no SLP vectorisation: v_pk_fma_f32 / v_pk_mul_f32 issue no faster than two scalar ops on gfx950 for this code and
    # cost register-pair shuffles (v_mov); measured -13 % kernel time on the fused launch
    "-fno-slp-vectorize",
    "-Wno-comment",
]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    from cppflow_amd import gen_robots

    if gen_robots.write() and verbose:
        print("regenerated", gen_robots.OUT)
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
