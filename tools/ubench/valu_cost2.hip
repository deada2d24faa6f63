// Second microbenchmark: select / compare / divide helpers (what does a lane select cost on gfx950?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int OP>
__global__ void k(unsigned long long* out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u, a3 = a0 * 3u, m = (threadIdx.x & 1) ? 0xffffffffu : 0u;
    float f0 = (float)(a0 & 1023) + 1.5f, f1 = f0 + 1.0f, f2 = f0 + 2.0f, f3 = f0 + 3.0f;
    double d0 = f0, d1 = f1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // each REP16 body = 4 independent instructions -> 64 instructions per iteration
        if constexpr (OP == 0) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));) }
        if constexpr (OP == 1) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "s10", "s11");) }
        if constexpr (OP == 2) { REP16(asm volatile("v_bfi_b32 %0, %5, %0, %4\n v_bfi_b32 %1, %5, %1, %4\n v_bfi_b32 %2, %5, %2, %4\n v_bfi_b32 %3, %5, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed), "v"(m));) }
        if constexpr (OP == 3) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4" : : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f1) : "vcc");) }
        if constexpr (OP == 4) { REP16(asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %4\n v_cmp_lt_f32_e64 s[12:13], %1, %4\n v_cmp_lt_f32_e64 s[14:15], %2, %4\n v_cmp_lt_f32_e64 s[16:17], %3, %4" : : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f1) : "s10","s11","s12","s13","s14","s15","s16","s17");) }
        if constexpr (OP == 5) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n s_nop 1\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n s_nop 1\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f1) : "vcc");) }   // 4 VALU + 2 nop
        if constexpr (OP == 6) { REP16(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f1));) }
        if constexpr (OP == 7) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0000001f));) }
        if constexpr (OP == 8) { REP16(asm volatile("v_mad_u64_u32 v[20:21], vcc, %0, %4, 0\n v_mad_u64_u32 v[22:23], vcc, %1, %4, 0\n v_mad_u64_u32 v[24:25], vcc, %2, %4, 0\n v_mad_u64_u32 v[26:27], vcc, %3, %4, 0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(0xD2511F53u) : "v20","v21","v22","v23","v24","v25","v26","v27","vcc");) }
        if constexpr (OP == 9) { REP16(asm volatile("v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_scale_f32 %1, vcc, %1, %4, %1\n v_div_scale_f32 %2, vcc, %2, %4, %2\n v_div_scale_f32 %3, vcc, %3, %4, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.5f) : "vcc");) }
        if constexpr (OP == 10) { REP16(asm volatile("v_div_fmas_f32 %0, %0, %4, %4\n v_div_fmas_f32 %1, %1, %4, %4\n v_div_fmas_f32 %2, %2, %4, %4\n v_div_fmas_f32 %3, %3, %4, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0000001f) : "vcc");) }
        if constexpr (OP == 11) { REP16(asm volatile("v_div_fixup_f32 %0, %0, %4, %4\n v_div_fixup_f32 %1, %1, %4, %4\n v_div_fixup_f32 %2, %2, %4, %4\n v_div_fixup_f32 %3, %3, %4, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.5f));) }
        if constexpr (OP == 12) { REP16(asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %6\n v_cvt_f32_u32 %3, %7" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if constexpr (OP == 13) { REP16(asm volatile("v_cvt_f64_f32 %0, %2\n v_cvt_f64_f32 %1, %3\n v_cvt_f64_f32 %0, %3\n v_cvt_f64_f32 %1, %2" : "=v"(d0), "=v"(d1) : "v"(f0), "v"(f1));) }
        if constexpr (OP == 14) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0\n v_cmp_lt_f64 vcc, %0, %1\n v_cmp_lt_f64 vcc, %1, %0" : : "v"(d0), "v"(d1) : "vcc");) }
        if constexpr (OP == 15) { REP16(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");) }
        if constexpr (OP == 16) { REP16(asm volatile("s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1");) }
        if constexpr (OP == 17) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));) }   // 1 cndmask among 3 xor
        if constexpr (OP == 18) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if constexpr (OP == 19) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned sink = a0 ^ a1 ^ a2 ^ a3 ^ __float_as_uint(f0 + f1 + f2 + f3) ^ (unsigned)(long long)(d0 + d1);
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (sink == 0x12345678u) out[blockIdx.x * 2 + 1] = sink;
}

template <int OP>
void run(const char* name, unsigned long long* d_out, double per_iter = 64.0)
{
    const int iters = 200;
    printf("%-34s:", name);
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 1024 * wps;
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(64), 0, 0, d_out, iters, 12345u);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 2);
        (void)hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost);
        double sum = 0;
        for (int b = 0; b < blocks; ++b) sum += (double)h[2 * b];
        double cyc = sum / blocks / (per_iter * iters);
        printf("  w%d %6.2f (%5.2f/SIMD)", wps, cyc, cyc / wps);
    }
    printf("\n");
}

int main()
{
    unsigned long long* d_out;
    (void)hipMalloc(&d_out, sizeof(unsigned long long) * 2 * 1024 * 8);
    printf("cycles per wave-instruction, 4 independent streams; (x/SIMD) = cycles of SIMD time per instruction\n");
    run<0>("v_cndmask_b32 vcc", d_out);
    run<1>("v_cndmask_b32_e64 sgpr-pair", d_out);
    run<17>("1 cndmask + 3 xor (per inst)", d_out);
    run<2>("v_bfi_b32", d_out);
    run<18>("v_and_b32", d_out);
    run<3>("v_cmp_lt_f32 -> vcc", d_out);
    run<4>("v_cmp_lt_f32_e64 -> sgpr pairs", d_out);
    run<5>("cmp,nop1,cndmask x2 (per VALU of 4)", d_out);
    run<19>("v_add_co_u32 (vcc out)", d_out);
    run<6>("v_max_f32", d_out);
    run<7>("v_mul_f32", d_out);
    run<8>("v_mad_u64_u32 alone", d_out);
    run<9>("v_div_scale_f32", d_out);
    run<10>("v_div_fmas_f32", d_out);
    run<11>("v_div_fixup_f32", d_out);
    run<12>("v_cvt_f32_u32", d_out);
    run<13>("v_cvt_f64_f32", d_out);
    run<14>("v_cmp_lt_f64", d_out);
    run<15>("s_nop 0", d_out);
    run<16>("s_nop 1", d_out);
    return 0;
}
