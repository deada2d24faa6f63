// Where do the waves of a launch land?  Each wave records XCC / SE / CU / SIMD ids; the host prints the histogram of
// waves per SIMD for a few (grid, block) shapes.  Build: hipcc --offload-arch=gfx950 -O2 placement.hip -o placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256) k(unsigned* out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float f = threadIdx.x;
    for (int i = 0; i < spin; ++i) f = f * 1.0000001f + 0.5f;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    if ((threadIdx.x & 63) == 0) { out[2 * wave] = hw; out[2 * wave + 1] = xcc & 0xf; }
    if (f == 12345.678f) out[0] = 0;
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 2 * 8192 * sizeof(unsigned));
    const int shapes[][2] = {{1024, 64}, {2048, 64}, {512, 256}, {1024, 128}, {4096, 64}, {1024, 256}, {256, 512}};
    for (auto& sh : shapes) {
        const int waves = sh[0] * sh[1] / 64;
        hipLaunchKernelGGL(k, dim3(sh[0]), dim3(sh[1]), 0, 0, d, 200000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(2 * waves);
        hipMemcpy(h.data(), d, 2 * waves * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_simd, per_cu;
        for (int w = 0; w < waves; ++w) {
            const unsigned hw = h[2 * w], xcc = h[2 * w + 1];
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh_id = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned cu_key = (xcc << 12) | (se << 8) | (sh_id << 4) | cu;
            per_cu[cu_key]++;
            per_simd[(cu_key << 2) | simd]++;
        }
        std::map<int, int> hist_simd, hist_cu;
        for (auto& kv : per_simd) hist_simd[kv.second]++;
        for (auto& kv : per_cu) hist_cu[kv.second]++;
        printf("grid %5d x %3d: %zu CUs, %zu SIMDs used | waves/CU:", sh[0], sh[1], per_cu.size(), per_simd.size());
        for (auto& kv : hist_cu) printf(" %dx%d", kv.second, kv.first);
        printf(" | waves/SIMD:");
        for (auto& kv : hist_simd) printf(" %dx%d", kv.second, kv.first);
        printf("\n");
    }
    return 0;
}
