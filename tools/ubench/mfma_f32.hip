// Calibration: sustained v_mfma_f32_32x32x2_f32 rate, NACC independent accumulators per wave, W waves per SIMD,
// optionally with one ds_read_b32 (A operand) + 2 VALU per MFMA as in the NF coupling kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDSOP>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float b = (float)threadIdx.x * 1e-3f;
    const float* p = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            float av = LDSOP ? p[((it * NACC + a) & 127) * 128] : b + (float)a;
            float bv = LDSOP ? __builtin_fmaxf(__builtin_fmaf(av, b, 0.5f), 0.0f) : b;
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDSOP>
void run(const char* name, float* d)
{
    const int iters = 4000;
    for (int wgs_per_cu : {1, 2}) {
        int blocks = 256 * wgs_per_cu;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k<NACC, LDSOP>), dim3(blocks), dim3(256), 0, 0, d, 100);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, LDSOP>), dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * NACC * 4096.0;
        printf("%-28s NACC=%d waves/SIMD=%d : %.1f TFLOP/s (%.1f%% of 157.3)\n", name, NACC, wgs_per_cu, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    }
}

int main()
{
    float* d; (void)hipMalloc(&d, 4 * 256 * 512);
    run<1, false>("registers", d); run<2, false>("registers", d); run<4, false>("registers", d);
    run<4, true>("lds A + fma/max B", d);
    return 0;
}
