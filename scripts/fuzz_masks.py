"""Developer fuzz of the collision stage: obstacle sets from none to the maximum (CPPF_MAX_OBSTACLES), cuboids that are thin plates /
rods / points, touch each other or contain the robot's base, random robots with spheres among their capsules -- masks, cost and signed
minimum distances of the standalone launch and of the fused launch against the canonical-fp32 oracle, bit for bit.  Prints every
disagreement; exits 1 if any."""
import sys, os
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers as H  # noqa: E402
from cppflow_amd.robot_model import CapsuleSpec, canonicalize  # noqa: E402
from cppflow_amd.robots import Robot, get_robot  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

DEV = "cuda:0"
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)
MAX_OBS = 8  # CPPF_MAX_OBSTACLES


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)


bad = checked = 0
rng = np.random.RandomState(7)


def obstacle_sets():
    yield "none", []
    yield "one", [H.cuboid_obstacle(0.3, 0.2, 0.5, 0.2, 0.2, 0.2)]
    yield "max", [H.cuboid_obstacle(*rng.uniform(-0.8, 0.8, 3), *rng.uniform(0.05, 0.3, 3)) for _ in range(MAX_OBS)]
    yield "plate rod point", [H.cuboid_obstacle(0.3, 0.0, 0.6, 0.5, 0.5, 0.0), H.cuboid_obstacle(-0.3, 0.2, 0.5, 0.0, 0.0, 0.8),
                              H.cuboid_obstacle(0.2, -0.3, 0.4, 0.0, 0.0, 0.0)]
    yield "around the base", [H.cuboid_obstacle(0.0, 0.0, 0.0, 0.4, 0.4, 0.4)]
    yield "far", [H.cuboid_obstacle(50.0, 50.0, 50.0, 1.0, 1.0, 1.0)]


def robots():
    for name in ("panda", "fetch", "chain12"):
        yield name, get_robot(name), H.oracle32(name), H.chain(name)
    for ndof, seed in ((6, 3), (7, 4), (9, 5)):
        spec = H.random_chain_spec(ndof, seed)
        # two of its capsules become spheres
        for i in (0, len(spec.capsules) - 1):
            c = spec.capsules[i]
            spec.capsules[i] = CapsuleSpec(c.link, c.p0, c.p0, c.radius)
        ch = canonicalize(spec)
        for specialize in (False, True):
            yield f"random{ndof}{'_rtc' if specialize else ''}", Robot(spec, specialize=specialize), Oracle(ch, f32=True), ch


for rname, rb, o32, ch in robots():
    d = rb.ndof
    rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
    jl_lo, jl_hi = rb.padded_joint_limits()
    for oname, obs in obstacle_sets():
        rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
        lo, hi = H.box_corners([c for c, _ in obs], [T for _, T in obs]) if obs else (None, None)
        for (S, W) in ((1, 1), (3, 64), (2, 257), (5, 100)):
            q = H.f32(rng.uniform(ch.lo, ch.hi, size=(S * W, d)))
            want = o32.masks(q, lo, hi, jl_lo, jl_hi)
            got = rb.collision_masks(dev(q).reshape(S, W, d), want_min_dists=True)
            torch.cuda.synchronize()
            checked += 1
            for k in ("self_mask", "env_mask", "jlim_mask"):
                if not np.array_equal(got[k].cpu().numpy().reshape(-1).astype(np.uint8), want[k]):
                    bad += 1
                    print("DIFF", rname, oname, S, W, k, int((got[k].cpu().numpy().reshape(-1) != want[k]).sum()), "rows")
            for k in ("ext_cost", "min_self", "min_env"):
                g = got[k].cpu().numpy().reshape(-1).astype(np.float64)
                if not np.array_equal(g, want[k]):
                    bad += 1
                    print("DIFF", rname, oname, S, W, k, np.nanmax(np.abs(g - want[k])))
            # the fused launch's collision stage at its own x (K = 1, so x stays near q)
            target = dev(H.oracle64(rname).fk(q[:W]) if rname in ("panda", "fetch", "chain12") else Oracle(ch, f32=False).fk(q[:W]))
            r = rb.lm_pose_steps(dev(q), target, n_steps=1, want_errors=True, want_collisions=True, want_min_dists=True, **LM)
            x1 = r["x"].cpu().numpy().astype(np.float64)
            w1 = o32.masks(x1, lo, hi, jl_lo, jl_hi)
            for k in ("self_mask", "env_mask", "jlim_mask"):
                if not np.array_equal(r[k].cpu().numpy().astype(np.uint8), w1[k]):
                    bad += 1
                    print("DIFF fused", rname, oname, S, W, k)
            for k in ("ext_cost", "min_self", "min_env"):
                if not np.array_equal(r[k].cpu().numpy().astype(np.float64), w1[k]):
                    bad += 1
                    print("DIFF fused", rname, oname, S, W, k)
    rb.set_obstacles([], [])
print("comparisons:", checked, " disagreements:", bad)
sys.exit(1 if bad else 0)
