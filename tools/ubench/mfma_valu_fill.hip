// How many vector instructions does one v_mfma_f32_32x32x2_f32 (64 cycles of a SIMD's matrix pipe) hide?
// A wavefront alternates G MFMAs (two accumulators in turn: no dependency stall) with G x F independent v_fma_f32 (+ optionally G
// ds_read_b32 with a dependent add each), W wavefronts per SIMD run the same stream.  Reported: shader cycles per MFMA per SIMD (s_memtime), i.e. 64 = the
// matrix pipe never waits.  The NF coupling kernel (csrc/glabc_nf.hip) carries ~3.2 vector + 1 LDS instruction per MFMA.
//   hipcc --offload-arch=gfx950 -O2 mfma_valu_fill.hip -o mfma_valu_fill && ./mfma_valu_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F, bool LDS, int G = 1>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = 0.5f;
    float f[F > 0 ? F : 1];
    for (int i = 0; i < (F > 0 ? F : 1); ++i) f[i] = (float)i;
    const float* p = lds + (threadIdx.x & 63);
    float l = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u0 = 0; u0 < 8; u0 += G) {                            // G MFMAs back to back, then their G x F vector instructions
#pragma unroll
            for (int u = u0; u < u0 + G; ++u) {
                if (u & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int u = u0; u < u0 + G; ++u) {
#pragma unroll
                for (int i = 0; i < F; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(a), "v"(b));
                if (LDS) l += p[((it * 8 + u) & 31) * 64];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = l;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int i = 0; i < F; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int F, bool LDS, int G = 1>
void run(float* d, unsigned long long* c)
{
    const int iters = 2000;
    for (int w : {1, 2, 3}) {
        const int blocks = 256 * w;                           // 256-thread workgroups (one wavefront per SIMD each), w per CU
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k<F, LDS, G>), dim3(blocks), dim3(256), 0, 0, d, c, 50);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<F, LDS, G>), dim3(blocks), dim3(256), 0, 0, d, c, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[3072]; (void)hipMemcpy(h, c, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < blocks * 4; ++i) mean += (double)h[i]; mean /= blocks * 4;
        const double n_mfma_simd = (double)iters * 8 * w;     // MFMAs a SIMD's pipe executed
        printf("F=%2d lds=%d G=%d waves/SIMD=%d : %6.1f cycles per MFMA per SIMD (wave lifetime / MFMAs of the SIMD), %6.2f ns, %.1f TFLOP/s\n", F, (int)LDS, G, w,
               mean / n_mfma_simd, ms * 1e6 / n_mfma_simd, (double)blocks * 4 * iters * 8 * 4096.0 / ms / 1e9);
    }
}

int main()
{
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 4 * 256 * 768); (void)hipMalloc(&c, 8 * 3072);
    run<0, false>(d, c); run<1, false>(d, c); run<2, false>(d, c); run<3, false>(d, c); run<4, false>(d, c); run<6, false>(d, c);
    run<8, false>(d, c); run<3, false, 2>(d, c); run<3, false, 4>(d, c); run<3, false, 8>(d, c); run<4, false, 8>(d, c);
    run<3, true>(d, c); run<3, true, 8>(d, c);
    return 0;
}
