// Microbenchmark: does a wave64 whose upper 32 lanes are masked off (exec = 0x00000000FFFFFFFF) issue VALU ops faster?
// If the SIMD skipped the empty half, 65 536 chains could run as 2048 half-filled waves (two per SIMD) at no extra
// instruction cost.  Build: hipcc --offload-arch=gfx950 -O2 half_exec.hip -o half_exec ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int ACTIVE>
__global__ void __launch_bounds__(64) k(float* out, int iters, float seed)
{
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = seed + threadIdx.x + i;
    if ((int)threadIdx.x < ACTIVE) {
        for (int it = 0; it < iters; ++it) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                               "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(1.0000001f));)
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ACTIVE>
void run(float* out)
{
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int waves = 1024 * wps;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(k<ACTIVE>, dim3(waves), dim3(64), 0, 0, out, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k<ACTIVE>, dim3(waves), dim3(64), 0, 0, out, iters, 1.0f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double insts = (double)waves * iters * 128;
        printf("active lanes %2d  waves/SIMD %d  %.3f ms  %.3e wave-insts/s  (%.2f cycles per inst per SIMD at 2.4 GHz)\n", ACTIVE, wps, ms,
               insts / (ms * 1e-3), 2.4e9 / (insts / (ms * 1e-3) / 1024));
    }
}

int main()
{
    float* out;
    hipMalloc(&out, 1024 * 8 * 64 * sizeof(float));
    run<64>(out);
    run<32>(out);
    run<16>(out);
    return 0;
}
