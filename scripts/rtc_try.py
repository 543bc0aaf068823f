#!/usr/bin/env python3
"""Compile a program dumped by CPPF_RTC_DUMP with hipRTC against the ON-DISK device headers (developer tool: iterating on the
headers without rebuilding the library, whose copy of them is embedded at build time).

    CPPF_RTC_DUMP=/tmp/prog.hip python -c "..."   # any call of cppf_robot_specialize / cppf_debug_rtc_compile
    python scripts/rtc_try.py /tmp/prog.hip
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cppflow_amd.build import CSRC, EMBEDDED, HIPCC_FLAGS  # noqa: E402

rtc = ctypes.CDLL("libhiprtc.so")
src = open(sys.argv[1]).read().encode()
names = EMBEDDED + ["../../include/cppflow_hip.h"]
paths = [os.path.join(CSRC, n) for n in EMBEDDED] + [os.path.join(ROOT, "include", "cppflow_hip.h")]
hdr_src = (ctypes.c_char_p * len(names))(*[open(p).read().encode() for p in paths])
hdr_names = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
prog = ctypes.c_void_p()
assert rtc.hiprtcCreateProgram(ctypes.byref(prog), src, b"cppf_custom_robot.hip", len(names), hdr_src, hdr_names) == 0
exprs = [b"cppf_rtc::lm_fused_kernel<cppf::StaRobot<cppf::gen::Custom>, 1, false>", b"cppf_rtc::collision_kernel<cppf::StaRobot<cppf::gen::Custom>, false>",
         b"cppf_rtc::lm_quad_kernel<cppf::StaRobot<cppf::gen::Custom>, 1, false>"]
for e in exprs:
    rtc.hiprtcAddNameExpression(prog, e)
opts = [o.encode() for o in HIPCC_FLAGS if o not in ("-fPIC", "-shared")]
t0 = time.time()
rc = rtc.hiprtcCompileProgram(prog, len(opts), (ctypes.c_char_p * len(opts))(*opts))
n = ctypes.c_size_t()
rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
log = ctypes.create_string_buffer(n.value + 1)
rtc.hiprtcGetProgramLog(prog, log)
print("rc", rc, f"{time.time() - t0:.1f}s")
print(log.value.decode()[:6000])
if rc == 0:
    for e in exprs:
        low = ctypes.c_char_p()
        rtc.hiprtcGetLoweredName(prog, e, ctypes.byref(low))
        print(e.decode(), "->", low.value.decode()[:100])
    rtc.hiprtcGetCodeSize(prog, ctypes.byref(n))
    print("code object bytes", n.value)
