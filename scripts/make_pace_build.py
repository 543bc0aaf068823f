#!/usr/bin/env python3
"""Diagnostic build for the FAIR-SHARE PACING experiment (scripts/pace_probe.py): a patched COPY of csrc/kernels_fused.h +
csrc/fused_static.hip in which every wavefront of the fused kernel compares its own progress with a clock-derived schedule before
each LM iteration, compiled (that unit only, with its own flags, ~1 minute) and linked with the in-tree object of the rest of the
library into build_var/lib_pace.so.  Never shipped, never loaded unless CPPFLOW_HIP_LIB points at it.

Why: an isolated launch's four wavefronts per SIMD finish in AGE order (oldest-first issue arbitration): the
oldest runs at nearly the lone-wavefront rate, the youngest does half its work alone at one instruction per 5.5 cycles.  If every
wavefront kept to the fair-share schedule (iteration k not before  start + k * pace), the four would finish together and the SIMD
would stay full to the end.  The environment variable CPPF_PACE, read by the patched launcher at every launch, holds the pace in
10 ns ticks of s_memrealtime per iteration: > 0 = a wavefront ahead of the schedule SLEEPS (s_sleep) until it is due; < 0 = it only
lowers its priority (s_setprio 0 when ahead, 3 when behind); 0 / unset = off.  The kernel also stamps start / loop end / end of every
wavefront into the n_iters output like scripts/make_stamp_build.py does (scripts/wave_timeline.py reads them)."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cppflow_amd import build  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build_var", "lib_pace.so")
build.build()  # the in-tree objects must be current
tree = tempfile.mkdtemp(prefix="cppf_pace_")
os.makedirs(os.path.join(tree, "cppflow_amd"))
shutil.copytree(build.CSRC, os.path.join(tree, "cppflow_amd", "csrc"), ignore=shutil.ignore_patterns("*.so", "obj"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tree, "include"))
csrc = os.path.join(tree, "cppflow_amd", "csrc")


def patch(name, pairs):
    p = os.path.join(csrc, name)
    s = open(p).read()
    for old, new in pairs:
        assert s.count(old) == 1, (name, old[:80])
        s = s.replace(old, new)
    open(p, "w").write(s)


patch("kernels_fused.h", [
    ("// ---- one row of the fused launch, in three pieces", """// progress table of the lag-ranked priority mode: one 64-byte line per physical SIMD, one word per wavefront slot = launch id << 8 | k
__device__ unsigned g_fair[8192 * 16];

__device__ __forceinline__ void set_prio(int p) {
    switch (p) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// mode (prm.n): 0 off; 1 = lag-ranked priority (the wavefront publishes its iteration count, reads its SIMD's line and takes priority
// 3 - number of wavefronts of the same launch on the SIMD that are BEHIND it); otherwise the clock schedule of prm.tol_rot2
__device__ __forceinline__ void pace_wait(unsigned long long t0, int k, float pace, int mode, unsigned id, unsigned& pending) {
    if (mode >= 1 && mode <= 3) {  // 1: synchronous read of the line; 2: the same traffic, priority untouched; 3: the line as read one check ago
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
        const unsigned simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u, my = hw & 15u;
        unsigned* const line = g_fair + (size_t)(((((xcc * 8u + se) * 2u + sh) * 16u + cu) * 4u + simd) * 16u);
        const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (lane == my) __hip_atomic_store(line + my, (id << 8) | (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned e;
        if (mode == 3) {
            e = pending;
            pending = __hip_atomic_load(line + (lane & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            e = __hip_atomic_load(line + (lane & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool behind = lane < 16u && lane != my && (e >> 8) == id && (int)(e & 0xffu) < k;
        const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(behind));
        if (mode != 2) set_prio(3 - (cnt > 3 ? 3 : cnt));
        return;
    }
    if (mode >= 5 && mode <= 7) {  // static bands (no clock, no constant): band = quarter of the grid = age rank on the SIMD
        const int band = (int)((blockIdx.x * 4u) / gridDim.x);
        if (mode == 5) set_prio(band < 2 ? 3 : (((k + band) & 1) ? 2 : 1));          // the two youngest bands take turns behind the two oldest
        else if (mode == 6) set_prio(((k + band) & 1) ? 2 : 1);                      // neighbours take turns, all four bands
        else set_prio(band < 2 ? 3 : (band == 2 ? 1 : 2));                           // the youngest band ahead of the third, statically
        return;
    }
    if (pace == 0.f) return;  // (scalar)
    const unsigned long long due = t0 + (unsigned long long)((float)k * fabsf(pace));
    if (pace > 0.f) {
        while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(4);  // 4 x 64 clocks ~ 0.1 us
    } else if (mode == 4) {  // four levels: behind by more than half an iteration 3, behind 2, ahead 1, ahead by more than half an iteration 0
        const long long lag = (long long)(__builtin_amdgcn_s_memrealtime() - due), half = (long long)(0.5f * fabsf(pace));
        set_prio(lag > half ? 3 : lag > 0 ? 2 : lag > -half ? 1 : 0);
    } else {
        if (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(3);
    }
}

// ---- one row of the fused launch, in three pieces"""),
    ("    const bool active = row < (size_t)itp->n;\n    float q[D];",
     "    const bool active = row < (size_t)itp->n;\n    const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();\n    float q[D];"),
    ("    RowSummary rs;\n    if (active) {\n        float Rt[9], tt[3];",
     "    RowSummary rs;\n    unsigned long long stamp_mid = 0;\n    unsigned fair_pending = 0;\n    if (active) {\n        float Rt[9], tt[3];"),
    ("            for (; it < prm.n_steps - 1; ++it) (void)lm_row_iterate<RB, true>(rb, prm, out, row, false, Rt, tt, gate_lds, q);",
     "            for (; it < prm.n_steps - 1; ++it) {\n                pace_wait(stamp0, it, prm.tol_rot2, prm.n, (unsigned)prm.W, fair_pending);\n"
     "                (void)lm_row_iterate<RB, true>(rb, prm, out, row, false, Rt, tt, gate_lds, q);\n            }"),
    ("        for (; it < prm.n_steps; ++it) {\n            const bool conv = lm_row_iterate<RB, false>",
     "        for (; it < prm.n_steps; ++it) {\n            if (!(prm.tol_pos2 > 0.f)) pace_wait(stamp0, it, prm.tol_rot2, prm.n, (unsigned)prm.W, fair_pending);\n            const bool conv = lm_row_iterate<RB, false>"),
    ("        if (out.n_iters) out.n_iters[row_b] = iters;",
     "        stamp_mid = __builtin_amdgcn_s_memrealtime();\n        (void)iters;\n        if (!(prm.tol_pos2 > 0.f)) pace_wait(stamp0, prm.n_steps, prm.tol_rot2, prm.n, (unsigned)prm.W, fair_pending);"),
    ("""            block_seed_summary<RB>(rb, itp->W, (size_t)(blk * (unsigned)kBlock + (unsigned)tid_c), active, q, rs, out.seed_summary);
        }
    }
}""", """            block_seed_summary<RB>(rb, itp->W, (size_t)(blk * (unsigned)kBlock + (unsigned)tid_c), active, q, rs, out.seed_summary);
        }
    }
    if (out.n_iters && active) {  // lane 0: start | end << 16 (10 ns ticks mod 65536), lane 1: start | loop end << 16, lanes 2, 3: HW_ID, XCC_ID
        const unsigned long long stamp1 = __builtin_amdgcn_s_memrealtime();
        int tid_d = threadIdx.x;
        asm volatile("" : "+v"(tid_d));
        const size_t row_d = (size_t)(blk * (unsigned)kBlock + (unsigned)tid_d);
        const unsigned long long e = ((tid_d & 63) == 1) ? stamp_mid : stamp1;
        int word = (int)((stamp0 & 0xFFFFull) | ((e & 0xFFFFull) << 16));
        if ((tid_d & 63) == 2) word = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        if ((tid_d & 63) == 3) word = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);
        out.n_iters[row_d] = word;
    }
}"""),
])
if os.environ.get("CPPF_PACE_HEAVY"):  # a wavefront that runs an exact distance test in the finish stage raises its priority (stragglers of the launch)
    patch("kernels_collision.h", [
        ("            const float d2 = seg_seg_dist2(wc[a], wh[a], wc[b], wh[b], T::cap_a[a], T::cap_ia[a], T::cap_a[b], T::cap_ia[b]);",
         "            if constexpr (!WANT_MIN) __builtin_amdgcn_s_setprio(3);\n            const float d2 = seg_seg_dist2(wc[a], wh[a], wc[b], wh[b], T::cap_a[a], T::cap_ia[a], T::cap_a[b], T::cap_ia[b]);"),
        ("                near = cuboids_in_reach(co, wc[c], T::cap_cull[c]);\n                if (near == 0u) continue;",
         "                near = cuboids_in_reach(co, wc[c], T::cap_cull[c]);\n                if (near == 0u) continue;\n                __builtin_amdgcn_s_setprio(3);"),
    ])
patch("fused_static.hip", [
    ("template <class Type>\nvoid launch_one(int coll, unsigned grid, size_t lds, hipStream_t st, const FusedArgs& args) {\n    using RB = StaRobot<Type>;",
     "template <class Type>\nvoid launch_one(int coll, unsigned grid, size_t lds, hipStream_t st, const FusedArgs& args_in) {\n    using RB = StaRobot<Type>;\n"
     "    FusedArgs args = args_in;\n    const char* pace = getenv(\"CPPF_PACE\");\n    if (!(args.prm.tol_pos2 > 0.f)) args.prm.tol_rot2 = pace ? (float)atof(pace) : 0.f;\n"
     "    static unsigned launch_id = 0;\n    const char* fair = getenv(\"CPPF_FAIR\");\n    args.prm.n = fair ? atoi(fair) : 0;\n    args.prm.W = (int)(++launch_id & 0xFFFFFFu);"),
    ('#include "kernels_chain.h"', '#include <cstdlib>\n#include "kernels_chain.h"'),
])
src = "fused_static.hip"
objdir = os.path.join(ROOT, "build_var", "obj_variant")
os.makedirs(objdir, exist_ok=True)
obj = os.path.join(objdir, os.path.basename(out_path) + ".fused_static.o")
cmd = ([os.environ.get("HIPCC", "hipcc")] + build.HIPCC_FLAGS + build.EXTRA_FLAGS.get(src, []) + sys.argv[2:] +
       [f'-DCPPF_BUILD_ID="{build.source_hash()}"', "-c", "-o", obj, os.path.join(csrc, src)])
print(" ".join(cmd))
subprocess.run(cmd, check=True, cwd=csrc)
others = [os.path.join(build.CSRC, "obj", u.replace(".hip", ".o")) for u in build.SOURCES if u != src]
subprocess.run([os.environ.get("HIPCC", "hipcc")] + build.link_flags() + ["-o", out_path] + others + [obj], check=True)
if not os.environ.get("CPPF_KEEP_TREE"):
    shutil.rmtree(tree)
else:
    print("kept", tree)
print(out_path)
