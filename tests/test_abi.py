"""The C-ABI library loads without a GPU and exports every symbol include/cppflow_hip.h and include/cppflow_hip_debug.h declare
(no compute calls)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cppflow_hip.h")
DEBUG_HEADER = os.path.join(ROOT, "include", "cppflow_hip_debug.h")


def build_c_client(out_dir: str) -> str:
    """gcc -std=c99 of tests/c_client/abi_client.c against include/cppflow_hip.h, the in-tree library and the HIP runtime's C
    API (no hipcc, no C++): proof that the boundary is a C ABI.  Returns the executable's path."""
    import subprocess

    from cppflow_amd import _hip, build

    build.build()
    exe = os.path.join(out_dir, "abi_client")
    libdir = os.path.dirname(_hip.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror=implicit-function-declaration", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_client", "abi_client.c"),
           "-o", exe, "-L" + libdir, "-lcppflow_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir,
           "-Wl,-rpath,/opt/rocm/lib"]  # fmt: skip
    run = subprocess.run(cmd, capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    return exe


def declared_functions(headers=(HEADER, DEBUG_HEADER)):
    names = set()
    for h in headers:
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(cppf_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


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
    assert lib.cppf_abi_version() == 6


def test_debug_header_holds_every_hook_and_nothing_of_the_boundary():
    """include/cppflow_hip_debug.h: every cppf_debug_* symbol lives there and only there (the public header declares none), its
    tuning keys are the ones the Python binding uses, and there is no process-wide setter left (every hook that changes dispatch
    takes the robot handle)."""
    from cppflow_amd import _hip

    public, debug = declared_functions((HEADER,)), declared_functions((DEBUG_HEADER,))
    assert not [n for n in public if n.startswith("cppf_debug_")]
    assert debug and all(n.startswith("cppf_debug_") for n in debug), debug
    text = open(DEBUG_HEADER).read()
    keys = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"#define CPPF_TUNE_([A-Z_0-9]+) (\d+)\n", text)}
    count = keys.pop("count")
    assert keys == _hip.TUNE_KEYS and count == len(keys)
    assert re.search(r"int cppf_debug_set\(cppf_robot\* robot, int key, int value\);", text)
    assert "process-wide" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    # without a device a handle cannot be created, but the argument checks run first
    lib_ = _hip.lib()
    assert lib_.cppf_debug_set(None, 0, 1) == _hip.CPPF_ERR_INVALID
    assert b"NULL" in lib_.cppf_last_error()


def test_struct_layouts_match_header_constants():
    from cppflow_amd import _hip
    from cppflow_amd.robot_model import MAX_CAPSULES, MAX_DOF, MAX_OBSTACLES, MAX_PAIRS

    text = open(HEADER).read()
    for name, val in (("CPPF_MAX_DOF", MAX_DOF), ("CPPF_MAX_CAPSULES", MAX_CAPSULES), ("CPPF_MAX_PAIRS", MAX_PAIRS),
                      ("CPPF_MAX_OBSTACLES", MAX_OBSTACLES)):  # fmt: skip
        assert int(re.search(rf"#define {name} (\d+)", text).group(1)) == val
    # sizeof(cppf_robot_desc): 4 + 12*48 + 48 + 48 + 48 + 48 + 4 + 96 + 288 + 288 + 96 + 4 + 1024
    assert ctypes.sizeof(_hip.RobotDesc) == 4 + 576 + 48 + 48 + 48 + 48 + 4 + 96 + 288 + 288 + 96 + 4 + 1024
    assert ctypes.sizeof(_hip.LmParams) == 40
    assert ctypes.sizeof(_hip.LmOutputs) == 13 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_hip.Constraints) == 24


def test_batch_item_layout_and_fused_argument_offset(lib):
    """cppf_lm_batch_item as ctypes sees it equals the C struct (two pointers, S, W, the 13 output pointers), and the fused kernel's
    view of its own kernel-argument segment is right: the device code reads the problem of a plain launch at offsetof(FusedArgs,
    single) from the segment's base (csrc/kernels_fused.h: fused_item) -- the code object's metadata must place the kernel's fourth
    argument exactly there, for a shipped table (the second translation unit) and a generic instantiation (the first)."""
    import os
    import subprocess
    import sys

    from cppflow_amd import _hip, build

    assert ctypes.sizeof(_hip.LmBatchItem) == 8 + 8 + 4 + 4 + 13 * 8
    text = open(HEADER).read()
    assert int(re.search(r"#define CPPF_MAX_BATCH (\d+)", text).group(1)) == _hip.MAX_BATCH
    want = lib.cppf_debug_fused_single_offset()
    assert want > 0 and want % 8 == 0
    sys.path.insert(0, os.path.join(os.path.dirname(build.CSRC), "..", "scripts"))
    import kernel_resources as kr

    co = os.path.join(os.environ.get("TMPDIR", "/tmp"), "cppf_abi_test.co")
    notes = ""
    for path in kr.extract_code_objects(build.OUT, co):
        notes += subprocess.run([f"{kr.LLVM}/llvm-readelf", "--notes", path], capture_output=True, text=True, check=True).stdout
    seen = 0
    for blk in notes.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        if "lm_fused_kernel" not in name:
            continue
        # the explicit arguments come first, in order: ChainK, CollK, LmK, BatchItemK single, table
        offs = [int(m.group(1)) for m in re.finditer(r"- \.offset:\s+(\d+)\n\s+\.size:\s+\d+\n\s+\.value_kind:\s+by_value", blk)]
        sizes = [int(m.group(1)) for m in re.finditer(r"- \.offset:\s+\d+\n\s+\.size:\s+(\d+)\n\s+\.value_kind:\s+by_value", blk)]
        assert len(offs) >= 4 and offs[3] == want and sizes[3] == 128, (name, offs[:5], sizes[:5], want)
        seen += 1
    assert seen >= 12 * 3, seen  # every shipped table x COLL and every generic ndof x COLL


def test_no_test_kernel_ships_in_the_product_library(lib):
    """a test-only kernel (the exhaustive reciprocal sweep) lived in libcppflow_hip.so.  Test kernels are a translation
    unit of their own (tests/native/, built by build_test_kernels into its own shared object); the product library exports no
    cppf_test_* / *_sweep symbol and holds no such kernel."""
    import os
    import subprocess
    import sys

    from cppflow_amd import build

    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(build.CSRC), "..", "scripts", "kernel_resources.py")],
                         capture_output=True, text=True, check=True).stdout
    assert "lm_fused_kernel" in out and "sweep" not in out and "test" not in out.lower().replace("latest", ""), [ln for ln in out.splitlines() if "sweep" in ln or "test" in ln.lower()]
    syms = subprocess.run(["nm", "-D", "--defined-only", build.OUT], capture_output=True, text=True, check=True).stdout
    assert "cppf_lm_batch_launch" in syms and "rcp_sweep" not in syms and "cppf_test_" not in syms
    assert os.path.exists(build.TEST_SRC)
    build.build_test_kernels()
    tk = subprocess.run(["nm", "-D", "--defined-only", build.TEST_OUT], capture_output=True, text=True, check=True).stdout
    assert "cppf_test_rcp_sweep" in tk


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
    # a capsule shorter than a micrometre but not of length 0 is a mistake (p0 == p1 exactly is a sphere and accepted)
    desc.cap_p1[3][0], desc.cap_p1[3][1], desc.cap_p1[3][2] = desc.cap_p0[3][0] + 3e-7, desc.cap_p0[3][1], desc.cap_p0[3][2]
    assert lib.cppf_robot_create(ctypes.byref(desc), 0, ctypes.byref(out)) == _hip.CPPF_ERR_INVALID
    assert b"degenerate" in lib.cppf_last_error()
    with pytest.raises(AssertionError):
        _hip.check(_hip.CPPF_ERR_INVALID)


def test_generated_robot_tables_are_current():
    """csrc/robots_gen.h is generated from robot_zoo.py; a stale header would silently run the generic kernels."""
    from cppflow_amd import gen_robots

    assert open(gen_robots.OUT).read() == gen_robots.generate()


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """include/cppflow_hip.h parses as C99 and as C++17 on its own, and the C99 client of tests/c_client/ compiles and
    links against the in-tree library (it is run on the GPU by tests/test_gpu_c_client.py)."""
    import subprocess

    for lang, std, header in (("c", "-std=c99", HEADER), ("c++", "-std=c++17", HEADER), ("c", "-std=c99", DEBUG_HEADER),
                              ("c++", "-std=c++17", DEBUG_HEADER)):
        run = subprocess.run(["gcc", "-x", lang, std, "-fsyntax-only", "-Wall", "-Wextra", "-Werror", header],
                             capture_output=True, text=True)  # fmt: skip
        assert run.returncode == 0, run.stderr
    exe = build_c_client(str(tmp_path))
    assert os.access(exe, os.X_OK)
    nm = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    assert "cppf_lm_pose_steps" in nm and "cppf_robot_create" in nm


def test_graft_entry_build_runs():
    """__graft_entry__.build(): compiles (or finds up to date) the HIP library and the oracle, checks the ABI version."""
    import importlib

    entry = importlib.import_module("__graft_entry__")
    entry.build()


def test_run_time_specialisation_compiles_without_a_gpu(lib, tmp_path):
    """cppf_robot_specialize's generate -> hipRTC -> cache stages need no device (cppf_debug_rtc_compile): a random 7-DoF chain
    compiles for gfx950, the cache entry appears, and a second call finds it (no recompilation)."""
    import time

    from cppflow_amd import _hip
    from cppflow_amd.robot_model import canonicalize
    from tests import helpers as H

    desc = _hip.chain_to_desc(canonicalize(H.random_chain_spec(7, seed=21)))
    t0 = time.perf_counter()
    rc = lib.cppf_debug_rtc_compile(ctypes.byref(desc), str(tmp_path).encode())
    t_compile = time.perf_counter() - t0
    msg = lib.cppf_last_error().decode()
    assert rc == _hip.CPPF_ERR_HIP and "device" in msg and "compilation failed" not in msg, msg[:2000]
    files = list(tmp_path.glob("robot_*.cppfrtc"))
    assert len(files) == 1 and files[0].stat().st_size > 50_000
    head = files[0].read_bytes()[:4096].split(b"\n")
    # magic, the checksum of the code object (16 hex digits), then the lowered kernel names
    assert head[0] == b"CPPFRTC2" and len(head[1]) == 16 and int(head[1], 16) >= 0
    assert b"lm_fused_kernel" in head[2] and b"Custom" in head[2]
    t0 = time.perf_counter()
    lib.cppf_debug_rtc_compile(ctypes.byref(desc), str(tmp_path).encode())
    assert time.perf_counter() - t0 < 0.5 * t_compile + 0.05  # served from the cache
    # a damaged entry (one byte of the code object flipped) fails its checksum: it is compiled again and rewritten, never loaded
    good = files[0].read_bytes()
    bad = bytearray(good)
    bad[-1000] ^= 0x55
    files[0].write_bytes(bytes(bad))
    t0 = time.perf_counter()
    lib.cppf_debug_rtc_compile(ctypes.byref(desc), str(tmp_path).encode())
    assert time.perf_counter() - t0 > 0.5 * t_compile  # (not served from the cache)
    assert files[0].read_bytes() != bytes(bad)  # rewritten (hipRTC's output is not byte-reproducible, so not compared with `good`)
    t0 = time.perf_counter()
    lib.cppf_debug_rtc_compile(ctypes.byref(desc), str(tmp_path).encode())
    assert time.perf_counter() - t0 < 0.5 * t_compile + 0.05  # and valid again
    # a different robot gets a different entry
    desc2 = _hip.chain_to_desc(canonicalize(H.random_chain_spec(7, seed=22)))
    lib.cppf_debug_rtc_compile(ctypes.byref(desc2), str(tmp_path).encode())
    assert len(list(tmp_path.glob("robot_*.cppfrtc"))) == 2


def test_library_carries_the_hash_of_the_sources_it_was_built_from(lib):
    """Build provenance: the binary is git-ignored and travels to the GPU box as a built artefact; its build id is the
    sha256 of the sources next to it, read both through the ABI and out of the file, and `_hip.lib()` refuses a mismatch."""
    from cppflow_amd import build

    assert lib.cppf_build_id().decode() == build.source_hash() == build.built_id()
    assert not build.needs_build()


def test_build_id_covers_every_translation_unit_and_its_flags(monkeypatch):
    """The library is linked from two translation units (csrc/fused_static.hip holds the headline kernel under other scheduler
    flags): the build id must change when either source, a shared header or the flags of ONE unit change -- otherwise records keyed
    by the build id (profiles/r3_issue.json, bench.py's roofline) could outlive the code they were taken from."""
    import os

    from cppflow_amd import build

    assert set(build.SOURCES) == {"cppflow_hip.hip", "fused_static.hip"}
    for src in build.SOURCES:
        assert os.path.exists(os.path.join(build.CSRC, src))
    assert "-amdgpu-sched-strategy=max-ilp" in build.EXTRA_FLAGS["fused_static.hip"]
    base = build.source_hash()
    monkeypatch.setitem(build.EXTRA_FLAGS, "fused_static.hip", build.EXTRA_FLAGS["fused_static.hip"] + ["-O1"])
    assert build.source_hash() != base
    monkeypatch.undo()
    assert build.source_hash() == base
    # the code object of EACH unit is found by the resource tool (the headline kernel lives in the second one)
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(build.CSRC), "..", "scripts", "kernel_resources.py"), "",
                          "lm_fused_kernel<cppf::StaRobot<cppf::gen::Panda>, 1>"], capture_output=True, text=True, check=True).stdout  # fmt: skip
    line = [ln for ln in out.splitlines() if "lm_fused_kernel" in ln]
    assert len(line) == 1, out
    vgpr, scratch = int(line[0].split()[1]), int(line[0].split()[7])
    assert vgpr <= 128 and scratch == 0, line[0]  # four wavefronts per SIMD, nothing spilled


def test_no_kernel_of_the_library_touches_scratch(lib):
    """the coupled step's parallel-in-time elimination at 8 joints (300 - 372 B per lane), the large-launch block kernel
    of FetchArm and the generic 7- / 8-joint chains, and the four-lanes-per-row kernel at 11 / 12 joints spilled registers to scratch
    memory.  Round 4: the elimination streams its coupling blocks instead of holding them, the other two are built for the
    occupancy their registers allow.  Every kernel of the shipped code objects must report private_segment_fixed_size 0."""
    import os
    import subprocess
    import sys

    from cppflow_amd import build

    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(build.CSRC), "..", "scripts", "kernel_resources.py")],
                         capture_output=True, text=True, check=True).stdout
    rows = [ln.split() for ln in out.splitlines() if ln.startswith("vgpr")]
    assert len(rows) > 300, len(rows)  # every instantiation of both translation units
    spilled = [" ".join(r[:10] + r[10:14]) for r in rows if int(r[7]) != 0]
    assert not spilled, spilled
