// Microbenchmark: wave-instructions per second per SIMD for the VALU ops the sampler leans on, measured with HIP
// events (wall clock) at 1, 2, 4, 8 waves per SIMD and 8 independent chains per wave -> the VALU issue roofline.
// Build: hipcc --offload-arch=gfx950 -O2 valu_peak.hip -o valu_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int OP>
__global__ void __launch_bounds__(64) k(float* out, int iters, float seed)
{
    float f[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { f[i] = seed + threadIdx.x + i; u[i] = (unsigned)(threadIdx.x * 977 + i); }
    float f8 = f[0] + 8, f9 = f[1] + 9, f10 = f[2] + 10, f11 = f[3] + 11, f12 = f[4] + 12, f13 = f[5] + 13, f14 = f[6] + 14, f15 = f[7] + 15;
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                               "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(1.0000001f));)
        } else if constexpr (OP == 1) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a = {f[0], f[1]}, b = {f[2], f[3]}, c = {f[4], f[5]}, d = {f[6], f[7]}, e = {f8, f9}, g = {f10, f11}, h = {f12, f13}, j = {f14, f15};
            f2 m = {1.0000001f, 0.9999999f};
            for (int r = 0; r < 16; ++r)
                asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                             "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(g), "+v"(h), "+v"(j) : "v"(m));
            f[0] = a.x + b.x + c.x + d.x + e.x + g.x + h.x + j.x + a.y + b.y + c.y + d.y + e.y + g.y + h.y + j.y;
        } else if constexpr (OP == 2) {
            REP16(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                               "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                               : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(0x9e3779b9u));)
        } else if constexpr (OP == 3) {
            REP16(asm volatile("v_mad_u64_u32 v[40:41], vcc, %0, %8, 0\n v_mad_u64_u32 v[42:43], vcc, %1, %8, 0\n v_mad_u64_u32 v[44:45], vcc, %2, %8, 0\n"
                               "v_mad_u64_u32 v[46:47], vcc, %3, %8, 0\n v_mad_u64_u32 v[48:49], vcc, %4, %8, 0\n v_mad_u64_u32 v[50:51], vcc, %5, %8, 0\n"
                               "v_mad_u64_u32 v[52:53], vcc, %6, %8, 0\n v_mad_u64_u32 v[54:55], vcc, %7, %8, 0"
                               : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "s"(0xD2511F53u)
                               : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "vcc");)
        } else if constexpr (OP == 4) {
            REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                               "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                               : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(0x9e3779b9u));)
        } else if constexpr (OP == 5) {
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(1.0000001f));)
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)u[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, float* out)
{
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int waves = 1024 * wps;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL((k<OP>), dim3(waves), dim3(64), 0, 0, out, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k<OP>), dim3(waves), dim3(64), 0, 0, out, iters, 1.0f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double insts = (double)iters * 128.0 * waves;           // wave-instructions
        const double per_simd = insts / (ms * 1e-3) / 1024.0;
        printf("%-14s waves/SIMD %d  %.3f ms  %.3g wave-inst/s/SIMD  = %.2f cycles/inst at 2.4 GHz\n", name, wps, ms, per_simd, 2.4e9 / per_simd);
    }
}

int main()
{
    float* out;
    hipMalloc(&out, 1024 * 8 * 64 * sizeof(float));
    run<0>("v_fma_f32", out);
    run<1>("v_pk_fma_f32", out);
    run<5>("v_mul_f32", out);
    run<2>("v_xor_b32", out);
    run<4>("v_add_u32", out);
    run<3>("v_mad_u64_u32", out);
    return 0;
}
