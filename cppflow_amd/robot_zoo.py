"""The robots the reference's problems use (`cppflow/problems/*.yaml`: "panda", "fetch", "fetch_arm") plus the
synthetic 12-DoF chain of BASELINE.json config 5.

THESE ARE THE BUILD'S OWN MODEL DEFINITIONS.  The reference gets its robots from the un-vendored `jrl` package
(`cppflow/data_type_utils.py:197`); no URDF, capsule table or ignored-pair list exists in the reference tree
(SURVEY.md Appendix A).  Kinematic constants below are the public franka_description / fetch_description URDF values
as listed in SURVEY.md Appendix A; what the reference tree itself pins is honoured and tested:

  * Fetch joint limits                       - `tests/search_test.py:35-42`
  * Fetch joint 0 prismatic (+z), 1-7 revolute - `tests/optimization_utils_test.py:69-94, 377-402`
  * Panda 7 revolute joints, `panda_link0` -> `panda_hand` - `tests/optimization_utils_test.py:98-107`,
    `cppflow/ros2/ros2_publisher.py:60-61`
  * `torso_lift_link` at q = 0 is unrotated w.r.t. the world - `cppflow/data_type_utils.py:65-73`

Collision capsules ("minimum bounding capsules for each joint", `cppflow/optimization_utils.py:644-647`) and the
checked-pair lists are authored here: one capsule per link along the segment to the next joint; pairs whose links are
fewer than `min_link_gap` moving links apart, and pairs that overlap in the all-zeros configuration, are not checked.

Fetch: what the reference's problem set says about the capsules (round 4; scripts/problem_plausibility.py,
tests/golden/problem_plausibility.json, tests/test_problem_plausibility.py).  No number in the reference pins a capsule, but its 13
Fetch / FetchArm problems -- copied from TORM with their obstacles -- are problems it SOLVES, so a model under which they have no
collision-free IK solution is wrong.  Round 3's Fetch flagged 42-50 % of all configurations as self-colliding and every solution of
the two `circle` problems as environment-colliding.  Three causes, each a modelling artefact rather than geometry:
  * capsules of links TWO joints apart (torso / upper arm, upper arm / forearm, forearm / wrist) end within a few millimetres of each
    other at the zero pose by construction -- each is a bounding volume around a joint housing, radius ~ half the length of the link
    between them -- so any bend of the joint between them overlaps their end caps (28 % of random configurations were flagged by
    these three pairs alone).  On the robot the joint limits ARE where those shells meet.  `min_link_gap = 3` for the Fetch arm
    (as for the synthetic chain): a pair is checked from three joints apart on.
  * the base was a horizontal capsule of radius 0.28 m: its top reached z = 0.47 m, above the real base (0.36 m), and touched the
    lower bar of the `circle` problems' window frame (z = 0.40 ... 0.45 m at x >= 0.25 m) in EVERY configuration -- the base does
    not move.  It is now a vertical capsule (a disc-like body): radius 0.30 m about the base's axis, 0.376 m high at x = 0.25 m.
  * the torso column (radius 0.14 m about x = -0.02 m of torso_lift_link) stood 7 cm IN FRONT of the shoulder's mounting face: the
    upper arm swung to +-90 deg lies at x = 0.12 m in that frame.  Now radius 0.12 m about x = -0.075 m: its front face at 0.045 m.
With these, every waypoint of the 13 problems has a collision-free IK solution (>= 99.7 %), 0-8 % of the IK solutions found are
flagged self-colliding, and 11 % (Fetch) / 19 % (FetchArm) of uniformly random configurations are (arm folded into the body).
"""

from math import pi
from typing import Dict

from cppflow_amd.robot_model import CapsuleSpec, JointSpec, RobotSpec

HALF_PI = pi / 2


def panda_spec() -> RobotSpec:
    joints = [
        JointSpec("panda_joint1", "panda_link1", (0, 0, 0.333), (0, 0, 0), (0, 0, 1), "revolute", (-2.8973, 2.8973)),
        JointSpec("panda_joint2", "panda_link2", (0, 0, 0), (-HALF_PI, 0, 0), (0, 0, 1), "revolute", (-1.7628, 1.7628)),
        JointSpec(
            "panda_joint3", "panda_link3", (0, -0.316, 0), (HALF_PI, 0, 0), (0, 0, 1), "revolute", (-2.8973, 2.8973)
        ),
        JointSpec(
            "panda_joint4", "panda_link4", (0.0825, 0, 0), (HALF_PI, 0, 0), (0, 0, 1), "revolute", (-3.0718, -0.0698)
        ),
        JointSpec(
            "panda_joint5", "panda_link5", (-0.0825, 0.384, 0), (-HALF_PI, 0, 0), (0, 0, 1), "revolute", (-2.8973, 2.8973)
        ),
        JointSpec("panda_joint6", "panda_link6", (0, 0, 0), (HALF_PI, 0, 0), (0, 0, 1), "revolute", (-0.0175, 3.7525)),
        JointSpec(
            "panda_joint7", "panda_link7", (0.088, 0, 0), (HALF_PI, 0, 0), (0, 0, 1), "revolute", (-2.8973, 2.8973)
        ),
        JointSpec("panda_joint8", "panda_link8", (0, 0, 0.107), jtype="fixed"),
        JointSpec("panda_hand_joint", "panda_hand", (0, 0, 0), (0, 0, -pi / 4), jtype="fixed"),
    ]
    capsules = [
        CapsuleSpec("panda_link0", (-0.06, 0, 0.06), (0.0, 0, 0.06), 0.09),
        CapsuleSpec("panda_link1", (0, 0, -0.19), (0, 0, -0.03), 0.065),
        CapsuleSpec("panda_link2", (0, 0, 0), (0, -0.13, 0), 0.065),
        CapsuleSpec("panda_link3", (0, 0, -0.17), (0.0825, 0, 0), 0.06),
        CapsuleSpec("panda_link4", (0, 0, 0), (-0.0825, 0.11, 0), 0.06),
        CapsuleSpec("panda_link5", (0, 0, -0.26), (0, 0.04, -0.02), 0.055),
        CapsuleSpec("panda_link6", (0, 0, 0), (0.088, 0, 0), 0.05),
        CapsuleSpec("panda_link7", (0, 0, 0), (0, 0, 0.09), 0.045),
        CapsuleSpec("panda_hand", (0, -0.085, 0.04), (0, 0.085, 0.04), 0.04),
    ]
    # link5's forearm capsule overlaps the wrist capsules at q = 0 (and in 23 % of random configurations): not checked
    return RobotSpec("panda", "Panda", "panda_link0", joints, capsules, min_link_gap=2, ignored_pairs=[(5, 7), (5, 8)])


def _fetch_joints(torso_fixed: bool):
    torso = JointSpec(
        "torso_lift_joint",
        "torso_lift_link",
        (-0.086875, 0, 0.37743),
        (0, 0, 0),
        (0, 0, 1),
        "fixed" if torso_fixed else "prismatic",
        (0.0, 0.38615),
    )
    return [
        torso,
        JointSpec(
            "shoulder_pan_joint", "shoulder_pan_link", (0.119525, 0, 0.34858), (0, 0, 0), (0, 0, 1), "revolute", (-1.6056, 1.6056)
        ),
        JointSpec(
            "shoulder_lift_joint", "shoulder_lift_link", (0.117, 0, 0.06), (0, 0, 0), (0, 1, 0), "revolute", (-1.221, 1.518)
        ),
        JointSpec("upperarm_roll_joint", "upperarm_roll_link", (0.219, 0, 0), (0, 0, 0), (1, 0, 0), "revolute", (-pi, pi)),
        JointSpec("elbow_flex_joint", "elbow_flex_link", (0.133, 0, 0), (0, 0, 0), (0, 1, 0), "revolute", (-2.251, 2.251)),
        JointSpec("forearm_roll_joint", "forearm_roll_link", (0.197, 0, 0), (0, 0, 0), (1, 0, 0), "revolute", (-pi, pi)),
        JointSpec("wrist_flex_joint", "wrist_flex_link", (0.1245, 0, 0), (0, 0, 0), (0, 1, 0), "revolute", (-2.16, 2.16)),
        JointSpec("wrist_roll_joint", "wrist_roll_link", (0.1385, 0, 0), (0, 0, 0), (1, 0, 0), "revolute", (-pi, pi)),
        JointSpec("gripper_axis", "gripper_link", (0.16645, 0, 0), jtype="fixed"),
    ]


_FETCH_CAPSULES = [
    # mobile base and the torso column (see "Fetch: what the reference's problem set says about the capsules" above)
    CapsuleSpec("base_link", (0, 0, 0.15), (0, 0, 0.21), 0.30),
    CapsuleSpec("torso_lift_link", (-0.075, 0, 0.05), (-0.075, 0, 0.55), 0.12),
    CapsuleSpec("shoulder_pan_link", (0, 0, 0), (0.117, 0, 0.06), 0.07),
    CapsuleSpec("shoulder_lift_link", (0, 0, 0), (0.219, 0, 0), 0.065),
    CapsuleSpec("upperarm_roll_link", (0, 0, 0), (0.133, 0, 0), 0.06),
    CapsuleSpec("elbow_flex_link", (0, 0, 0), (0.197, 0, 0), 0.06),
    CapsuleSpec("forearm_roll_link", (0, 0, 0), (0.1245, 0, 0), 0.055),
    CapsuleSpec("wrist_flex_link", (0, 0, 0), (0.1385, 0, 0), 0.055),
    CapsuleSpec("wrist_roll_link", (0, 0, 0), (0.11, 0, 0), 0.05),
    CapsuleSpec("gripper_link", (-0.03, -0.06, 0), (-0.03, 0.06, 0), 0.04),
]


def fetch_spec() -> RobotSpec:
    return RobotSpec("fetch", "Fetch", "base_link", _fetch_joints(False), list(_FETCH_CAPSULES), min_link_gap=3)


def fetch_arm_spec() -> RobotSpec:
    return RobotSpec("fetch_arm", "Fetch - Arm (no lift joint)", "base_link", _fetch_joints(True), list(_FETCH_CAPSULES), min_link_gap=3)


def chain12_spec() -> RobotSpec:
    """Synthetic 12-DoF chain of BASELINE.json config 5: axes alternate z / y, 0.15 m links along z, limits +-2.8."""
    joints, capsules = [], [CapsuleSpec("link_base", (0, 0, 0), (0, 0, 0.10), 0.05)]
    for i in range(12):
        axis = (0, 0, 1) if i % 2 == 0 else (0, 1, 0)
        joints.append(JointSpec(f"joint_{i}", f"link_{i}", (0, 0, 0.15), (0, 0, 0), axis, "revolute", (-2.8, 2.8)))
        capsules.append(CapsuleSpec(f"link_{i}", (0, 0, 0.02), (0, 0, 0.13), 0.035))
    joints.append(JointSpec("tool_joint", "tool", (0, 0, 0.15), jtype="fixed"))
    return RobotSpec("chain12", "Synthetic 12-DoF chain", "link_base", joints, capsules, min_link_gap=3)


ROBOT_SPECS: Dict[str, callable] = {
    "panda": panda_spec,
    "fetch": fetch_spec,
    "fetch_arm": fetch_arm_spec,
    "chain12": chain12_spec,
}
