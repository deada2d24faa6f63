// slp_check.hip -- standalone reproducer for the SLP-vectorizer miscompile found through the run-time compiled kernels
// (profiles/r02_rtc_miscompile_bisect.txt).
//
// hiprtc (ROCm 7.2) compiles sampler_kernel<GLMCMC, theta_dim 3, y_dim 2, N, one lane per chain, VAR_GENERIC> -- the library's
// own headers, read from this repository at run time -- around a simulator that reads the normals eps[0] and eps[2], the
// cosine halves of BOTH Box-Muller pairs of one Philox block (the configuration of tests/test_rtc.py::
// test_hip_self_check_refuses_a_miscompiled_kernel).  With the SLP vectorizer on (the compiler's default) the
// simulator is handed noise that is no Philox output of the chain for N >= 12; with -fno-slp-vectorize the same source gives
// the CPU checker's bits (tests/test_rtc.py).  This program compiles the same source both ways, runs both kernels on the same
// inputs and counts the chains whose histories differ.  It depends on nothing but hiprtc and the five headers.
// (Compiled OFFLINE by hipcc with the same options the two builds agree -- an earlier form of this file did that: the fault
// needs hiprtc's own device headers in the translation unit.)
//
//   cd tools/ubench && hipcc --offload-arch=gfx950 -O2 -std=c++17 slp_check.hip -o slp_check -lhiprtc && ./slp_check
//   -> "N = 12: 171 of 512 chains differ ..."   (0 everywhere once the toolchain is fixed)
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../gl-abc-mcmc_amd/csrc/glabc_pack.h"       // host side: StepArgs<D, YD> and its packer (pulls in glabc_device.h)

constexpr int D = 3, YD = 2;
using Args = glabc::StepArgs<D, YD>;

static std::string slurp(const std::string& path)
{
    std::ifstream f(path);
    if (!f) {
        std::fprintf(stderr, "cannot read %s (run from tools/ubench)\n", path.c_str());
        std::exit(2);
    }
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// the simulator of tests/test_rtc.py (NONLINEAR): theta[3], eps[4] -> y[2]; y[0] reads eps[0] and eps[1], y[1] reads eps[2] and eps[3]
static const char* SIMULATOR =
    "GLABC_SIMULATOR void glabc_user_simulate(const float* theta, const float* eps, float* y)\n"
    "{\n"
    "    const float a = glabc_expf(-0.5f * fabsf(theta[0]));\n"
    "    const float r = sqrtf(theta[1] * theta[1] + 0.25f);\n"
    "    float sn, cs;\n"
    "    glabc_sincos2pi(0.125f, &sn, &cs);\n"
    "    y[0] = fmaf(a, r, 0.1f * eps[0]) + cs * (0.05f * eps[1]);\n"
    "    y[1] = glabc_logf(1.0f + theta[2] * theta[2]) * glabc_expf(0.1f * eps[2]) + 0.02f * eps[3];\n"
    "}\n";

struct Built {
    hipModule_t module;
    hipFunction_t fn;
};

static Built build(int n_batch, bool slp)
{
    // the translation unit glabc_rtc.hip (glabc_rtc_compile) hands to hiprtc, text for text
    const std::string glabc_h = slurp("../../include/glabc.h"), numerics_h = slurp("../../include/glabc_numerics.h"),
                      device_h = slurp("../../gl-abc-mcmc_amd/csrc/glabc_device.h"),
                      sampler_h = slurp("../../gl-abc-mcmc_amd/csrc/glabc_sampler.h"),
                      kernel_h = slurp("../../gl-abc-mcmc_amd/csrc/glabc_rtc_kernel.h");
    char defs[600], name[200];
    std::snprintf(defs, sizeof defs,
                  "#define GLABC_RTC_L 1\n"
                  "#define GLABC_RTC_ALGO 0\n#define GLABC_RTC_D %d\n#define GLABC_RTC_YD %d\n#define GLABC_RTC_N %d\n"
                  "#define GLABC_USER_SIM 1\n#define GLABC_USER_NOISE_DIM 4\n#define GLABC_THETA_DIM %d\n#define GLABC_Y_DIM %d\n"
                  "#define GLABC_NOISE_DIM 4\n#define GLABC_SIMULATOR static __device__ __forceinline__\n",
                  D, YD, n_batch, D, YD);
    std::string src =
        "typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
        "typedef int int32_t; typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t;\n"
        "typedef unsigned long uintptr_t;\n";
    src += defs;
    src += "#include \"glabc_numerics.h\"\n#line 1 \"user_simulator\"\n";
    src += SIMULATOR;
    src += "\n#line 1 \"glabc_rtc_main\"\n#include \"glabc_rtc_kernel.h\"\n";
    std::snprintf(name, sizeof name, "glabc::sampler_kernel<0, %d, %d, %d, 1, glabc::VAR_GENERIC, 0>", D, YD, n_batch);
    const char* headers[] = {glabc_h.c_str(), numerics_h.c_str(), glabc_h.c_str(), numerics_h.c_str(), device_h.c_str(), sampler_h.c_str(),
                             kernel_h.c_str()};
    const char* names[] = {"glabc.h", "glabc_numerics.h", "../../include/glabc.h", "../../include/glabc_numerics.h", "glabc_device.h",
                           "glabc_sampler.h", "glabc_rtc_kernel.h"};
    if (const char* dump = std::getenv("SLP_CHECK_DUMP")) {
        if (FILE* f = std::fopen(dump, "w")) {
            std::fputs(src.c_str(), f);
            std::fclose(f);
        }
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "glabc_rtc.hip", 7, headers, names) != HIPRTC_SUCCESS) std::exit(3);
    hiprtcAddNameExpression(prog, name);
    const char* opts[8] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"};
    int n_opts = 5;
    if (slp) opts[n_opts++] = "-fslp-vectorize";             // the later option wins: the vectorizer back on
    if (hiprtcCompileProgram(prog, n_opts, opts) != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        hiprtcGetProgramLog(prog, &log[0]);
        std::fprintf(stderr, "hiprtc: %s\n", log.c_str());
        std::exit(4);
    }
    const char* lowered = nullptr;
    if (hiprtcGetLoweredName(prog, name, &lowered) != HIPRTC_SUCCESS) std::exit(5);
    size_t size = 0;
    hiprtcGetCodeSize(prog, &size);
    std::vector<char> code(size);
    hiprtcGetCode(prog, code.data());
    Built b;
    if (hipModuleLoadData(&b.module, code.data()) != hipSuccess) std::exit(6);
    if (hipModuleGetFunction(&b.fn, b.module, lowered) != hipSuccess) std::exit(7);
    hiprtcDestroyProgram(&prog);
    return b;
}

static void fill_gauss(glabc_dist* g, int dim, float loc, float scale)
{
    std::memset(g, 0, sizeof *g);
    g->kind = GLABC_DIST_DIAG_GAUSS;
    g->dim = dim;
    for (int j = 0; j < dim; ++j) {
        g->p0[j] = loc;
        g->p1[j] = logf(scale);
        g->p2[j] = scale;
    }
    g->c0 = (float)(-0.5 * dim * 1.8378770664093453);
}

int main()
{
    const int n = 512, T = 6;
    glabc_model m;
    std::memset(&m, 0, sizeof m);
    m.sim_kind = GLABC_SIM_USER;
    m.theta_dim = D;
    m.y_dim = YD;
    // the configuration of CompiledModel.self_check in tests/test_rtc.py::test_hip_self_check_refuses_a_miscompiled_kernel:
    // prior N((0, 0.5, 0), diag(1.5, 1, 2)^2), importance proposal = the prior, local increments 0.3 x the prior's scales
    const float ploc[3] = {0.0f, 0.5f, 0.0f}, pscale[3] = {1.5f, 1.0f, 2.0f};
    fill_gauss(&m.prior, D, 0.0f, 1.0f);
    fill_gauss(&m.noise, 4, 0.0f, 1.0f);
    glabc_dist local, global;
    fill_gauss(&local, D, 0.0f, 1.0f);
    for (int j = 0; j < D; ++j) {
        m.prior.p0[j] = ploc[j];
        m.prior.p1[j] = logf(pscale[j]);
        m.prior.p2[j] = expf(logf(pscale[j]));
        local.p1[j] = logf(0.3f * pscale[j]);
        local.p2[j] = expf(logf(0.3f * pscale[j]));
    }
    global = m.prior;
    m.y_obs[0] = 0.9f;
    m.y_obs[1] = 0.6f;
    m.kern_log_scale = logf(0.15f);
    m.kern_scale = expf(logf(0.15f));
    m.kern_c0 = (float)(-0.5 * 1.8378770664093453);
    std::vector<float> th0(D * n), y0(YD * n);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; };
    for (auto& v : th0) v = 2.0f * rnd();
    for (auto& v : y0) v = 0.5f + rnd();
    int total_bad = 0;
    for (int N : {8, 11, 12, 13, 16}) {        // the run-time compiled kernels went wrong from N = 12 on
        float* hist[2];
        for (int b = 0; b < 2; ++b) {
            const Built k = build(N, b == 0);
            float *theta, *y, *log_w;
            uint32_t* flags;
            (void)hipMalloc(&theta, sizeof(float) * D * n);
            (void)hipMalloc(&y, sizeof(float) * YD * n);
            (void)hipMalloc(&log_w, sizeof(float) * n);
            (void)hipMalloc(&flags, sizeof(uint32_t) * n);
            (void)hipMalloc(&hist[b], sizeof(float) * T * D * n);
            (void)hipMemcpy(theta, th0.data(), sizeof(float) * D * n, hipMemcpyHostToDevice);
            (void)hipMemcpy(y, y0.data(), sizeof(float) * YD * n, hipMemcpyHostToDevice);
            (void)hipMemset(log_w, 0, sizeof(float) * n);
            std::vector<uint32_t> fl(n, GLABC_FLAG_LOCAL);
            (void)hipMemcpy(flags, fl.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice);
            glabc_chains c;
            std::memset(&c, 0, sizeof c);
            c.n_chains = n;
            c.stride = n;
            c.theta = theta;
            c.y = y;
            c.log_w = log_w;
            c.flags = flags;
            glabc_run r;
            std::memset(&r, 0, sizeof r);
            r.seed = 20240229;
            r.step0 = 1;
            r.n_steps = T;
            r.global_frequency = 0.5f;                 // as the self-check: iSIR and random-walk moves
            r.batch_size = N;
            r.history = hist[b];
            r.hist_stride = n;
            Args a = glabc::pack_args_rinv<D, YD>(&m, &local, &global, &c, &r, 0.0f);
            void* params[] = {&a};
            if (hipModuleLaunchKernel(k.fn, (n + 63) / 64, 1, 1, 64, 1, 1, 0, nullptr, params, nullptr) != hipSuccess ||
                hipDeviceSynchronize() != hipSuccess) {
                std::printf("launch failed\n");
                return 2;
            }
        }
        std::vector<float> h0(T * D * n), h1(T * D * n);
        (void)hipMemcpy(h0.data(), hist[0], sizeof(float) * h0.size(), hipMemcpyDeviceToHost);
        (void)hipMemcpy(h1.data(), hist[1], sizeof(float) * h1.size(), hipMemcpyDeviceToHost);
        int bad = 0, moved = 0;
        for (int ch = 0; ch < n; ++ch) {
            bool diff = false, mv = false;
            for (int t = 0; t < T; ++t)
                for (int j = 0; j < D; ++j) {
                    const size_t idx = ((size_t)t * D + j) * n + ch;
                    diff = diff || std::memcmp(&h0[idx], &h1[idx], 4) != 0;
                    mv = mv || h1[idx] != th0[(size_t)j * n + ch];
                }
            bad += diff;
            moved += mv;
        }
        std::printf("N = %2d: %d of %d chains differ between the SLP and the -fno-slp-vectorize build (%d chains moved)\n", N, bad, n, moved);
        total_bad += bad;
    }
    std::printf(total_bad ? "MISCOMPILE REPRODUCED\n" : "no difference\n");
    return 0;
}
