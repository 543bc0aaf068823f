// rtc_specialize.h -- compile-time tables for ANY robot description: the heavy kernels compiled for one robot at run time.
// Host side only; part of the translation unit cppflow_hip.hip (included after struct cppf_robot); gfx950 only.
//
// The shipped robots run instantiations over generated `static constexpr` tables (robots_gen.h): chain constants are literals, the
// 0 / +-1 entries of the fixed transforms fold away, capsule end points stay in VGPRs.  A description that matches no table used to
// fall back to the generic kernels (constants from the kernel-argument segment, capsules in LDS), 2x slower on the fused launch.
// cppf_robot_specialize closes that gap: it writes the same table for the handle's description, hands it to hipRTC together with
// the device headers (embedded in the library as strings, embedded_src.inc) and loads the resulting code object as a module; the
// dispatch then launches the module's functions through hipModuleLaunchKernel.  The code object is cached on disk, keyed by a
// hash of the generated source, the embedded headers and the compile options.
//
// hipRTC is loaded with dlopen (like RCCL): a process that never specialises needs no libhiprtc.
#pragma once

#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <cstdlib>
#include <fstream>
#include <sstream>

#include "embedded_src.inc"

namespace {

enum RtcKernel { RTC_FUSED0 = 0, RTC_FUSED1, RTC_FUSED2, RTC_COLL_MASK, RTC_COLL_MIN, RTC_QUAD0, RTC_QUAD1, RTC_COUNT };

const char* const kRtcNameExpr[RTC_COUNT] = {
    "cppf_rtc::lm_fused_kernel<cppf::StaRobot<cppf::gen::Custom>, 0>",
    "cppf_rtc::lm_fused_kernel<cppf::StaRobot<cppf::gen::Custom>, 1>",
    "cppf_rtc::lm_fused_kernel<cppf::StaRobot<cppf::gen::Custom>, 2>",
    "cppf_rtc::collision_kernel<cppf::StaRobot<cppf::gen::Custom>, false>",
    "cppf_rtc::collision_kernel<cppf::StaRobot<cppf::gen::Custom>, true>",
    "cppf_rtc::lm_quad_kernel<cppf::StaRobot<cppf::gen::Custom>, 0, false>",
    "cppf_rtc::lm_quad_kernel<cppf::StaRobot<cppf::gen::Custom>, 1, false>",
};

// (the machine scheduler of fused_static.hip: the fused kernel gains 2.4 %, the quad kernels 1.5 %, the collision kernel loses 1 %)
const char* const kRtcOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                   "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-Wno-comment",
                                   "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-enable-post-misched=0"};
constexpr int kRtcOptionCount = (int)(sizeof(kRtcOptions) / sizeof(kRtcOptions[0]));

struct RtcModule {
    hipModule_t module = nullptr;
    hipFunction_t fn[RTC_COUNT] = {};
};

// ---- hipRTC through dlopen ------------------------------------------------------------------------------------------------------
typedef struct _hiprtcProgram* RtcProgram;
struct RtcApi {
    void* handle = nullptr;
    int (*CreateProgram)(RtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*AddNameExpression)(RtcProgram, const char*) = nullptr;
    int (*CompileProgram)(RtcProgram, int, const char* const*) = nullptr;
    int (*GetProgramLogSize)(RtcProgram, size_t*) = nullptr;
    int (*GetProgramLog)(RtcProgram, char*) = nullptr;
    int (*GetCodeSize)(RtcProgram, size_t*) = nullptr;
    int (*GetCode)(RtcProgram, char*) = nullptr;
    int (*GetLoweredName)(RtcProgram, const char*, const char**) = nullptr;
    int (*DestroyProgram)(RtcProgram*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*Version)(int*, int*) = nullptr;  // optional (hiprtcVersion): part of the cache key
};
RtcApi g_rtc;

int load_hiprtc() {
    if (g_rtc.handle) return CPPF_OK;
    const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(CPPF_ERR_UNSUPPORTED, std::string("cppflow_hip: cannot load hipRTC (libhiprtc.so): ") + dlerror());
    RtcApi api;
    api.handle = h;
#define CPPF_SYM(field, name)                                                                          \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name));                                 \
    if (!api.field) return fail(CPPF_ERR_UNSUPPORTED, std::string("cppflow_hip: hipRTC lacks ") + name)
    CPPF_SYM(CreateProgram, "hiprtcCreateProgram");
    CPPF_SYM(AddNameExpression, "hiprtcAddNameExpression");
    CPPF_SYM(CompileProgram, "hiprtcCompileProgram");
    CPPF_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize");
    CPPF_SYM(GetProgramLog, "hiprtcGetProgramLog");
    CPPF_SYM(GetCodeSize, "hiprtcGetCodeSize");
    CPPF_SYM(GetCode, "hiprtcGetCode");
    CPPF_SYM(GetLoweredName, "hiprtcGetLoweredName");
    CPPF_SYM(DestroyProgram, "hiprtcDestroyProgram");
    CPPF_SYM(GetErrorString, "hiprtcGetErrorString");
#undef CPPF_SYM
    api.Version = reinterpret_cast<decltype(api.Version)>(dlsym(h, "hiprtcVersion"));
    g_rtc = api;
    return CPPF_OK;
}

// "major.minor" of the hipRTC in this process ("?" if it cannot say): a code object compiled by another compiler is not reused
std::string rtc_version_string() {
    int major = 0, minor = 0;
    if (load_hiprtc() != CPPF_OK || !g_rtc.Version || g_rtc.Version(&major, &minor) != 0) return "?";
    return std::to_string(major) + "." + std::to_string(minor);
}

// ---- the table, in the format of robots_gen.h (cppflow_amd/gen_robots.py:emit_robot) ------------------------------------------
std::string rtc_float(float v) {
    if (v != v) return "__builtin_nanf(\"\")";  // (%a would print "nanf" / "inff", which is not a literal)
    if (v == INFINITY) return "__builtin_huge_valf()";
    if (v == -INFINITY) return "(-__builtin_huge_valf())";
    char buf[64];
    std::snprintf(buf, sizeof buf, "%af", (double)v);  // C99 hex float: exact
    return buf;
}

template <class It>
std::string rtc_array(It begin, It end) {
    std::string s = "{";
    for (It it = begin; it != end; ++it) {
        if (it != begin) s += ", ";
        s += rtc_float(*it);
    }
    return s + "}";
}

std::string rtc_table_source(const cppf_robot& rb) {
    const cppf_robot_desc& d = rb.desc;
    const CollK& co = rb.coll;
    const int D = d.ndof, L = d.n_capsules, P = d.n_pairs, Lm = L > 0 ? L : 1, Pm = P > 0 ? P : 1;
    std::ostringstream o;
    o << "namespace cppf { namespace gen {\nstruct Custom {\n";
    o << "    static constexpr const char* name = \"custom\";\n";
    o << "    static constexpr int D = " << D << ", L = " << L << ", P = " << P << ";\n";
    o << "    static constexpr uint32_t pris_mask = " << rb.chain.pris_mask << "u;\n";
    o << "    static constexpr float F[" << D << "][12] = {\n";
    for (int j = 0; j < D; ++j) o << "        " << rtc_array(d.F[j], d.F[j] + 12) << ",\n";
    o << "    };\n";
    o << "    static constexpr float Fee[12] = " << rtc_array(d.F_ee, d.F_ee + 12) << ";\n";
    o << "    static constexpr float lo[" << D << "] = " << rtc_array(d.lo, d.lo + D) << ";\n";
    o << "    static constexpr float hi[" << D << "] = " << rtc_array(d.hi, d.hi + D) << ";\n";
    o << "    static constexpr int cap_link[" << Lm << "] = {";
    for (int c = 0; c < Lm; ++c) o << (c ? ", " : "") << (L ? d.cap_link[c] : 0);
    o << "};\n";
    auto vec3s = [&](const char* nm, const float (*p)[3]) {
        o << "    static constexpr float " << nm << "[" << Lm << "][3] = {";
        for (int c = 0; c < Lm; ++c) {
            const float z[3] = {0.f, 0.f, 0.f};
            const float* v = L ? p[c] : z;
            o << (c ? ", " : "") << rtc_array(v, v + 3);
        }
        o << "};\n";
    };
    vec3s("cap_p0", d.cap_p0);
    vec3s("cap_p1", d.cap_p1);
    vec3s("cap_c", co.cap_c);
    vec3s("cap_h", co.cap_h);
    std::vector<float> cap_r(Lm, 0.f), cap_thr(Lm, 0.f), cap_cull(Lm, 0.f), pair_thr(Pm, 0.f), pair_cull(Pm, 0.f), cap_a(Lm, 0.f), cap_ia(Lm, 0.f);
    for (int c = 0; c < L; ++c)
        cap_r[c] = d.cap_r[c], cap_thr[c] = co.cap_thr[c], cap_cull[c] = co.cap_cull[c], cap_a[c] = co.cap_a[c], cap_ia[c] = co.cap_ia[c];
    for (int p = 0; p < P; ++p) pair_thr[p] = co.pair_thr[p], pair_cull[p] = co.pair_cull[p];
    o << "    static constexpr float cap_a[" << Lm << "] = " << rtc_array(cap_a.begin(), cap_a.end()) << ";\n";
    o << "    static constexpr float cap_ia[" << Lm << "] = " << rtc_array(cap_ia.begin(), cap_ia.end()) << ";\n";
    o << "    static constexpr float cap_r[" << Lm << "] = " << rtc_array(cap_r.begin(), cap_r.end()) << ";\n";
    o << "    static constexpr float pair_thr[" << Pm << "] = " << rtc_array(pair_thr.begin(), pair_thr.end()) << ";\n";
    o << "    static constexpr float cap_thr[" << Lm << "] = " << rtc_array(cap_thr.begin(), cap_thr.end()) << ";\n";
    o << "    static constexpr float pair_cull[" << Pm << "] = " << rtc_array(pair_cull.begin(), pair_cull.end()) << ";\n";
    o << "    static constexpr float cap_cull[" << Lm << "] = " << rtc_array(cap_cull.begin(), cap_cull.end()) << ";\n";
    for (int side = 0; side < 2; ++side) {
        o << "    static constexpr int pair_" << (side ? "b" : "a") << "[" << Pm << "] = {";
        for (int p = 0; p < Pm; ++p) o << (p ? ", " : "") << (P ? d.pairs[p][side] : 0);
        o << "};\n";
    }
    o << "};\n} }\n";
    return o.str();
}

std::string rtc_program_source(const cppf_robot& rb) {
    std::string s;
    s += "// generated by cppf_robot_specialize (csrc/rtc_specialize.h)\n";
    s += "#include \"lmik_device.h\"\n";
    s += rtc_table_source(rb);
    s += "using namespace cppf;\n";
    s += "namespace cppf_rtc {\n";
    s += "constexpr int kBlock = " + std::to_string(kBlock) + ";\n";
    s += "#define CPPF_WAVES_LM " + std::to_string(CPPF_WAVES_LM) + "\n";
    s += "#define CPPF_WAVES_COLL " + std::to_string(CPPF_WAVES_COLL) + "\n";
    s += "#include \"kernels_chain.h\"\n#include \"kernels_collision.h\"\n#include \"kernels_fused.h\"\n#include \"kernels_quad.h\"\n";
    s += "}  // namespace cppf_rtc\n";
    return s;
}

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

std::string rtc_cache_dir(const char* cache_dir) {
    if (cache_dir && *cache_dir) return cache_dir;
    if (const char* e = std::getenv("CPPF_CACHE_DIR"))
        if (*e) return e;
    if (const char* home = std::getenv("HOME"))
        if (*home) return std::string(home) + "/.cache/cppflow_amd";
    return "/tmp/cppflow_amd-" + std::to_string((long)getuid());
}

// Parents with the usual 0755, the cache directory itself private to the user (0700): what is read back from it is loaded
// into the GPU as code.
void mkdir_p(const std::string& path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty()) (void)mkdir(cur.c_str(), i == path.size() ? 0700 : 0755);
        }
        if (i < path.size()) cur += path[i];
    }
}

// Is `dir` ours alone?  (owned by this user, a directory, not writable by group or others.)  A cache in a directory another local
// user could have created -- the /tmp fall-back when neither $CPPF_CACHE_DIR nor $HOME is set -- is neither read nor written.
bool rtc_cache_dir_trusted(const std::string& dir) {
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

// cache file: "CPPFRTC2\n" + fnv1a-64 of the code object (hex) + "\n" + RTC_COUNT lowered names (one per line) + code object bytes
bool rtc_cache_read(const std::string& dir, const std::string& file, std::vector<std::string>& names, std::string& code) {
    if (!rtc_cache_dir_trusted(dir)) return false;
    std::ifstream f(file, std::ios::binary);
    if (!f) return false;
    std::string magic, sum;
    if (!std::getline(f, magic) || magic != "CPPFRTC2" || !std::getline(f, sum)) return false;
    names.clear();
    for (int i = 0; i < RTC_COUNT; ++i) {
        std::string n;
        if (!std::getline(f, n) || n.empty()) return false;
        names.push_back(n);
    }
    std::ostringstream rest;
    rest << f.rdbuf();
    code = rest.str();
    char want[32];
    std::snprintf(want, sizeof want, "%016llx", (unsigned long long)fnv1a(code));
    return code.size() > 64 && sum == want;  // a truncated or altered file is recompiled, not loaded
}

void rtc_cache_write(const std::string& dir, const std::string& file, const std::vector<std::string>& names, const std::string& code) {
    mkdir_p(dir);
    if (!rtc_cache_dir_trusted(dir)) return;  // an unusable cache only costs the next process a compile
    const std::string tmp = file + ".tmp." + std::to_string((long)getpid());
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) return;
        char sum[32];
        std::snprintf(sum, sizeof sum, "%016llx", (unsigned long long)fnv1a(code));
        f << "CPPFRTC2\n" << sum << "\n";
        for (const std::string& n : names) f << n << "\n";
        f.write(code.data(), (std::streamsize)code.size());
        if (!f) return;
    }
    (void)std::rename(tmp.c_str(), file.c_str());
}

int rtc_compile(const std::string& source, bool with_quad, std::vector<std::string>& names, std::string& code) {
    if (int rc = load_hiprtc()) return rc;
    RtcProgram prog = nullptr;
    int r = g_rtc.CreateProgram(&prog, source.c_str(), "cppf_custom_robot.hip", kEmbeddedCount, kEmbeddedSources, kEmbeddedNames);
    if (r != 0) return fail(CPPF_ERR_HIP, std::string("cppflow_hip: hiprtcCreateProgram: ") + g_rtc.GetErrorString(r));
    auto wanted = [&](int i) { return with_quad || (i != RTC_QUAD0 && i != RTC_QUAD1); };  // the quad shape needs ndof >= 6
    for (int i = 0; i < RTC_COUNT; ++i)
        if (wanted(i)) (void)g_rtc.AddNameExpression(prog, kRtcNameExpr[i]);
    r = g_rtc.CompileProgram(prog, kRtcOptionCount, kRtcOptions);
    if (r != 0) {
        size_t n = 0;
        std::string log;
        if (g_rtc.GetProgramLogSize(prog, &n) == 0 && n > 1) {
            log.resize(n);
            (void)g_rtc.GetProgramLog(prog, &log[0]);
        }
        (void)g_rtc.DestroyProgram(&prog);
        return fail(CPPF_ERR_HIP, std::string("cppflow_hip: hipRTC compilation failed: ") + g_rtc.GetErrorString(r) + "\n" + log.substr(0, 4000));
    }
    names.clear();
    for (int i = 0; i < RTC_COUNT; ++i) {
        const char* low = nullptr;
        if (!wanted(i)) {
            names.push_back("-");
            continue;
        }
        r = g_rtc.GetLoweredName(prog, kRtcNameExpr[i], &low);
        if (r != 0 || !low) {
            (void)g_rtc.DestroyProgram(&prog);
            return fail(CPPF_ERR_HIP, std::string("cppflow_hip: no lowered name for ") + kRtcNameExpr[i]);
        }
        names.push_back(low);
    }
    size_t n = 0;
    r = g_rtc.GetCodeSize(prog, &n);
    if (r == 0) {
        code.resize(n);
        r = g_rtc.GetCode(prog, &code[0]);
    }
    (void)g_rtc.DestroyProgram(&prog);
    if (r != 0) return fail(CPPF_ERR_HIP, std::string("cppflow_hip: hiprtcGetCode: ") + g_rtc.GetErrorString(r));
    return CPPF_OK;
}

}  // namespace
