// device check: glabc_fxsplit vs glabc_fxsum over random terms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../include/glabc_numerics.h"
__global__ void k(const int64_t* q, int n, uint64_t* out)
{
    glabc_fxsum a = {0, 0, 0};
    glabc_fxsplit b = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 64) { glabc_fx_add(&a, q[i]); glabc_fxs_add(&b, q[i]); }
    glabc_fxsum c = glabc_fxs_finish(&b);
    uint64_t* o = out + 6 * threadIdx.x;
    if (threadIdx.x == 0) { out[64*6+0] = b.a; out[64*6+1] = (uint64_t)b.b; out[64*6+2] = b.c; out[64*6+3]=(uint64_t)b.s1; }
    o[0] = (uint64_t)a.s1; o[1] = a.s2_lo; o[2] = a.s2_hi; o[3] = (uint64_t)c.s1; o[4] = c.s2_lo; o[5] = c.s2_hi;
}
int main()
{
    const int n = 64 * 500;
    int64_t* h = new int64_t[n];
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int64_t)(s >> 20) - ((int64_t)1 << 43); if (i % 7 == 0) h[i] = -h[i] / 3; }
    int64_t* d; uint64_t* o;
    hipMalloc(&d, n * 8); hipMalloc(&o, (64 * 6 + 4) * 8);
    hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, o);
    uint64_t r[64 * 6 + 4];
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) for (int j = 0; j < 3; ++j) if (r[6 * t + j] != r[6 * t + 3 + j]) { if (bad < 5) printf("lane %d word %d: %llx vs %llx\n", t, j, (unsigned long long)r[6*t+j], (unsigned long long)r[6*t+3+j]); ++bad; }
    { glabc_fxsum a = {0,0,0}; glabc_fxsplit b = {0,0,0,0}; for (int i = 0; i < n; i += 64) { glabc_fx_add(&a, h[i]); glabc_fxs_add(&b, h[i]); }
      glabc_fxsum c = glabc_fxs_finish(&b);
      printf("host lane0: fxsum %llx %llx %llx | split-finish %llx %llx %llx\n", (unsigned long long)a.s1, (unsigned long long)a.s2_lo, (unsigned long long)a.s2_hi, (unsigned long long)c.s1, (unsigned long long)c.s2_lo, (unsigned long long)c.s2_hi);
      printf("host a b c: %llx %llx %llx | dev a b c: %llx %llx %llx\n", (unsigned long long)b.a, (unsigned long long)b.b, (unsigned long long)b.c, (unsigned long long)r[384], (unsigned long long)r[385], (unsigned long long)r[386]);
      printf("dev lane0: fxsum %llx %llx %llx | split-finish %llx %llx %llx\n", (unsigned long long)r[0], (unsigned long long)r[1], (unsigned long long)r[2], (unsigned long long)r[3], (unsigned long long)r[4], (unsigned long long)r[5]); }
    printf("mismatches: %d\n", bad);
    return 0;
}
