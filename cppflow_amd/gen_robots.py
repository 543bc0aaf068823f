"""Emit csrc/robots_gen.h: compile-time tables of the shipped robots' canonical chains.

The heavy kernels are instantiated once per shipped robot with these tables as `static constexpr` data.  After full
unrolling every chain constant is a literal, so (a) the 0 / +-1 entries of the fixed transforms fold away (Panda's
joint frames differ by +-90 degree rolls: R*F becomes a signed column permutation, no arithmetic), (b) no constant
competes for SGPRs or is re-fetched through the scalar cache every LM iteration, (c) capsule end points are indexed by
compile-time ids and stay in VGPRs instead of LDS.  `cppf_robot_create` picks a table by exact comparison with the
description it is handed; anything else runs the generic (kernel-argument-driven) instantiation.

Run by cppflow_amd.build before hipcc; the output is committed so that a bare `hipcc` of csrc/ also works.
"""

import os

import numpy as np

from cppflow_amd.robot_model import CanonicalChain, canonicalize
from cppflow_amd.robot_zoo import ROBOT_SPECS

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "robots_gen.h")
ORDER = ["panda", "fetch", "fetch_arm", "chain12"]


def _f(v) -> str:
    """Exact fp32 literal (C99 hex float)."""
    v = float(np.float32(v))
    if v == 0.0:
        return "0.0f"
    if v == 1.0:
        return "1.0f"
    if v == -1.0:
        return "-1.0f"
    return v.hex() + "f"


def _arr(vals) -> str:
    return "{" + ", ".join(_f(v) for v in vals) + "}"


def sqrt_threshold(r) -> np.float32:
    """Smallest fp32 y with sqrt_rn(y) >= r: then for every fp32 d2 >= 0,  sqrtf(d2) - r < 0  <=>  d2 < y  exactly
    (correctly rounded sqrt is monotone).  Lets the mask-only kernels skip the square root without changing a single bit
    of the masks the sqrt-based definition (collision_detection.py:66-68: `min_dists < 0`) produces."""
    r = np.float32(r)
    if r <= 0:
        return np.float32(0.0)
    y = np.float32(r * r)
    while np.sqrt(y) >= r:
        y = np.nextafter(y, np.float32(0.0))
    while np.sqrt(y) < r:
        y = np.nextafter(y, np.float32(np.inf))
    return y


def cull_threshold(reach: float) -> np.float32:
    """Broad-phase threshold on a squared mid-point distance: (reach + 1 cm)^2 (1 + 1e-4), rounded up to fp32 (reach = half
    lengths + radii).  Conservative by construction -- see cull_far in csrc/kernels_collision.h."""
    y = (float(reach) + 0.01) ** 2 * (1.0 + 1e-4)
    f = np.float32(y)
    if float(f) < y:
        f = np.nextafter(f, np.float32(np.inf))
    return f


def capsule_centred(cap_p0, cap_p1):
    """(centre, half-axis, |h|^2, 1 / |h|^2) of every capsule as the kernels use them: double arithmetic on the fp32 end points,
    each rounded to fp32 once -- c = 0.5 (p0 + p1), h = 0.5 (p1 - p0), a = (h0 h0 + h1 h1) + h2 h2 over the ROUNDED h, 1 / a (0 when a < 2^-100).
    The same four lines are in csrc/cppflow_hip.hip (capsule_centred) and oracle/lmik_oracle.c (orc_robot_create)."""
    p0 = np.asarray(cap_p0, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    p1 = np.asarray(cap_p1, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    c = (0.5 * (p0 + p1)).astype(np.float32)
    h = (0.5 * (p1 - p0)).astype(np.float32)
    hd = h.astype(np.float64)
    a = (hd[:, 0] * hd[:, 0] + hd[:, 1] * hd[:, 1]) + hd[:, 2] * hd[:, 2]
    ia = np.where(a >= 2.0**-100, 1.0 / np.maximum(a, 2.0**-100), 0.0)  # a zero-length capsule is a sphere: its parameter stays 0
    return c, h, a.astype(np.float32), ia.astype(np.float32), np.sqrt(a)


def emit_robot(name: str, ch: CanonicalChain) -> str:
    d, L, P = ch.ndof, ch.n_capsules, ch.n_pairs
    cname = "".join(p.capitalize() for p in name.split("_"))
    s = [f"struct {cname} {{"]
    s.append(f'    static constexpr const char* name = "{name}";')
    s.append(f"    static constexpr int D = {d}, L = {L}, P = {P};")
    s.append(f"    static constexpr uint32_t pris_mask = {sum(1 << j for j in range(d) if ch.jtype[j] == 1)}u;")
    s.append(f"    static constexpr float F[{d}][12] = {{")
    for j in range(d):
        s.append("        " + _arr(ch.F[j]) + ",")
    s.append("    };")
    s.append(f"    static constexpr float Fee[12] = {_arr(ch.F_ee)};")
    s.append(f"    static constexpr float lo[{d}] = {_arr(ch.lo)};")
    s.append(f"    static constexpr float hi[{d}] = {_arr(ch.hi)};")
    Lm, Pm = max(L, 1), max(P, 1)
    s.append(f"    static constexpr int cap_link[{Lm}] = {{" + ", ".join(str(int(v)) for v in (ch.cap_link if L else [0])) + "};")
    s.append(f"    static constexpr float cap_p0[{Lm}][3] = {{" + ", ".join(_arr(v) for v in (ch.cap_p0 if L else [[0, 0, 0]])) + "};")
    s.append(f"    static constexpr float cap_p1[{Lm}][3] = {{" + ", ".join(_arr(v) for v in (ch.cap_p1 if L else [[0, 0, 0]])) + "};")
    s.append(f"    static constexpr float cap_r[{Lm}] = {_arr(ch.cap_r if L else [0])};")
    cc, hh, aa, ia, half = capsule_centred(ch.cap_p0, ch.cap_p1) if L else (np.zeros((1, 3)), np.zeros((1, 3)), [0], [0], [])
    s.append(f"    static constexpr float cap_c[{Lm}][3] = {{" + ", ".join(_arr(v) for v in cc) + "};  // centre 0.5 (p0 + p1)")
    s.append(f"    static constexpr float cap_h[{Lm}][3] = {{" + ", ".join(_arr(v) for v in hh) + "};  // half-axis 0.5 (p1 - p0)")
    s.append(f"    static constexpr float cap_a[{Lm}] = {_arr(aa)};  // |h|^2")
    s.append(f"    static constexpr float cap_ia[{Lm}] = {_arr(ia)};  // 1 / |h|^2")
    r32 = ch.cap_r.astype(np.float32)
    pair_thr = [sqrt_threshold(np.float32(r32[a] + r32[b])) for a, b in ch.pairs] if P else [0]
    cap_thr = [sqrt_threshold(r) for r in r32] if L else [0]
    s.append(f"    static constexpr float pair_thr[{Pm}] = {_arr(pair_thr)};  // sqrt thresholds of r_a + r_b")
    s.append(f"    static constexpr float cap_thr[{Lm}] = {_arr(cap_thr)};  // sqrt thresholds of r")
    pair_cull = [cull_threshold(half[a] + half[b] + float(r32[a]) + float(r32[b])) for a, b in ch.pairs] if P else [0]
    cap_cull = [cull_threshold(half[c] + float(r32[c])) for c in range(L)] if L else [0]
    s.append(f"    static constexpr float pair_cull[{Pm}] = {_arr(pair_cull)};  // broad phase, pairs")
    s.append(f"    static constexpr float cap_cull[{Lm}] = {_arr(cap_cull)};  // broad phase, capsule vs cuboid")
    s.append(f"    static constexpr int pair_a[{Pm}] = {{" + ", ".join(str(int(v)) for v in (ch.pairs[:, 0] if P else [0])) + "};")
    s.append(f"    static constexpr int pair_b[{Pm}] = {{" + ", ".join(str(int(v)) for v in (ch.pairs[:, 1] if P else [0])) + "};")
    s.append("};")
    return "\n".join(s)


def generate() -> str:
    parts = [
        "// robots_gen.h -- GENERATED by cppflow_amd/gen_robots.py from cppflow_amd/robot_zoo.py; do not edit.",
        "// Canonical chains (cppflow_amd/robot_model.py) of the shipped robots as compile-time tables.",
        "#pragma once",
        "#include <stdint.h>",
        "",
        "namespace cppf {",
        "namespace gen {",
        "",
    ]
    names = []
    for name in ORDER:
        ch = canonicalize(ROBOT_SPECS[name]())
        parts.append(emit_robot(name, ch))
        parts.append("")
        names.append("".join(p.capitalize() for p in name.split("_")))
    parts.append(f"constexpr int kNumStaticRobots = {len(names)};")
    parts.append("}  // namespace gen")
    parts.append("}  // namespace cppf")
    parts.append("")
    parts.append("// X-macro over the generated robots: CPPF_FOR_EACH_STATIC_ROBOT(M) expands M(index, Type)")
    parts.append("#define CPPF_FOR_EACH_STATIC_ROBOT(M) \\")
    parts.append(" \\\n".join(f"    M({i}, ::cppf::gen::{n})" for i, n in enumerate(names)))
    parts.append("")
    return "\n".join(parts)


def write(force: bool = False) -> bool:
    text = generate()
    if not force and os.path.exists(OUT) and open(OUT).read() == text:
        return False
    with open(OUT, "w") as f:
        f.write(text)
    return True


if __name__ == "__main__":
    print(OUT, "written" if write() else "up to date")
