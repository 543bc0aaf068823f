"""The C-ABI library loads without a GPU and exports every symbol include/cppflow_hip.h declares (no compute calls)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cppflow_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cppf_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from cppflow_amd import _hip, build

    build.build()
    return _hip.lib()


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("cppf_robot_create", "cppf_lm_pose_steps", "cppf_collision_masks", "cppf_forward_kinematics",
                 "cppf_jacobian", "cppf_pose_errors", "cppf_clamp_to_joint_limits", "cppf_seed_validity"):  # fmt: skip
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    from cppflow_amd import _hip

    names = declared_functions()
    assert set(names) == set(_hip.SIGNATURES), set(names) ^ set(_hip.SIGNATURES)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.cppf_abi_version() == 2


def test_struct_layouts_match_header_constants():
    from cppflow_amd import _hip
    from cppflow_amd.robot_model import MAX_CAPSULES, MAX_DOF, MAX_OBSTACLES, MAX_PAIRS

    text = open(HEADER).read()
    for name, val in (("CPPF_MAX_DOF", MAX_DOF), ("CPPF_MAX_CAPSULES", MAX_CAPSULES), ("CPPF_MAX_PAIRS", MAX_PAIRS),
                      ("CPPF_MAX_OBSTACLES", MAX_OBSTACLES)):  # fmt: skip
        assert int(re.search(rf"#define {name} (\d+)", text).group(1)) == val
    # sizeof(cppf_robot_desc): 4 + 16*48 + 48 + 64 + 64 + 64 + 4 + 96 + 288 + 288 + 96 + 4 + 1024
    assert ctypes.sizeof(_hip.RobotDesc) == 4 + 768 + 48 + 64 + 64 + 64 + 4 + 96 + 288 + 288 + 96 + 4 + 1024
    assert ctypes.sizeof(_hip.LmParams) == 20
    assert ctypes.sizeof(_hip.LmOutputs) == 12 * ctypes.sizeof(ctypes.c_void_p)


def test_invalid_descriptions_are_rejected_without_a_gpu(lib):
    """Argument validation happens before any HIP call, so it is testable here."""
    from cppflow_amd import _hip
    from cppflow_amd.robot_model import canonicalize
    from cppflow_amd.robot_zoo import ROBOT_SPECS

    desc = _hip.chain_to_desc(canonicalize(ROBOT_SPECS["panda"]()))
    out = ctypes.c_void_p()
    desc.ndof = 0
    assert lib.cppf_robot_create(ctypes.byref(desc), 0, ctypes.byref(out)) == _hip.CPPF_ERR_INVALID
    assert b"ndof" in lib.cppf_last_error()
    desc.ndof = 7
    desc.cap_p1[3][0], desc.cap_p1[3][1], desc.cap_p1[3][2] = desc.cap_p0[3][0], desc.cap_p0[3][1], desc.cap_p0[3][2]
    assert lib.cppf_robot_create(ctypes.byref(desc), 0, ctypes.byref(out)) == _hip.CPPF_ERR_INVALID
    assert b"degenerate" in lib.cppf_last_error()
    with pytest.raises(AssertionError):
        _hip.check(_hip.CPPF_ERR_INVALID)


def test_generated_robot_tables_are_current():
    """csrc/robots_gen.h is generated from robot_zoo.py; a stale header would silently run the generic kernels."""
    from cppflow_amd import gen_robots

    assert open(gen_robots.OUT).read() == gen_robots.generate()
