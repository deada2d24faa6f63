// Microbenchmark: cycles per wave-instruction of the VALU ops the sampler leans on, for a
// dependent chain and for 4 independent chains, at 1 / 2 / 4 / 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 valu_cost.hip -o valu_cost ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int OP, int ILP>
__global__ void k(unsigned long long* out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u, a3 = a0 * 3u;
    float f0 = (float)(a0 & 1023) + 1.5f, f1 = f0 + 1.0f, f2 = f0 + 2.0f, f3 = f0 + 3.0f;
    double d0 = f0, d1 = f1, d2 = f2, d3 = f3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) {   // v_mad_u64_u32 (hi used)
            if constexpr (ILP == 1) { REP64(asm volatile("v_mad_u64_u32 v[20:21], vcc, %0, %1, 0\n v_mov_b32 %0, v21" : "+v"(a0) : "s"(0xD2511F53u) : "v20", "v21", "vcc");) }
            else { REP16(asm volatile("v_mad_u64_u32 v[20:21], vcc, %0, %4, 0\n v_mad_u64_u32 v[22:23], vcc, %1, %4, 0\n v_mad_u64_u32 v[24:25], vcc, %2, %4, 0\n v_mad_u64_u32 v[26:27], vcc, %3, %4, 0\n v_mov_b32 %0, v21\n v_mov_b32 %1, v23\n v_mov_b32 %2, v25\n v_mov_b32 %3, v27" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(0xD2511F53u) : "v20","v21","v22","v23","v24","v25","v26","v27","vcc");) }
        } else if constexpr (OP == 1) {   // v_mul_hi_u32
            if constexpr (ILP == 1) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a0) : "s"(0xD2511F53u));) }
            else { REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(0xD2511F53u));) }
        } else if constexpr (OP == 2) {   // v_mul_lo_u32
            if constexpr (ILP == 1) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "s"(0xD2511F53u));) }
            else { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(0xD2511F53u));) }
        } else if constexpr (OP == 3) {   // v_xor_b32
            if constexpr (ILP == 1) { REP64(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(a1));) }
            else { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));) }
        } else if constexpr (OP == 4) {   // v_fma_f32
            if constexpr (ILP == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(f1));) }
            else { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0000001f));) }
        } else if constexpr (OP == 5) {   // v_sqrt_f32
            if constexpr (ILP == 1) { REP64(asm volatile("v_sqrt_f32 %0, %0\n s_nop 0" : "+v"(f0));) }
            else { REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n s_nop 0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        } else if constexpr (OP == 6) {   // v_rcp_f32
            if constexpr (ILP == 1) { REP64(asm volatile("v_rcp_f32 %0, %0\n s_nop 0" : "+v"(f0));) }
            else { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n s_nop 0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
        } else if constexpr (OP == 7) {   // v_fma_f64
            if constexpr (ILP == 1) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(d1));) }
            else { REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0000001));) }
        } else if constexpr (OP == 8) {   // v_add_f64
            if constexpr (ILP == 1) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(d1));) }
            else { REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0000001));) }
        } else if constexpr (OP == 9) {   // ds_bpermute_b32 (dependent: wait each)
            if constexpr (ILP == 1) { REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(a1 & 252u));) }
            else { REP16(asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"((threadIdx.x ^ 1u) * 4u));) }
        } else if constexpr (OP == 10) {  // v_cndmask
            if constexpr (ILP == 1) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(a1) : );) }
            else { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));) }
        } else if constexpr (OP == 11) {  // v_pk_fma_f32
            if constexpr (ILP == 1) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d0) : "v"(d1));) }
            else { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d3));) }
        } else if constexpr (OP == 12) {  // v_mov_dpp quad_perm
            if constexpr (ILP == 1) { REP64(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0));) }
            else { REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned sink = a0 ^ a1 ^ a2 ^ a3 ^ __float_as_uint(f0 + f1 + f2 + f3) ^ (unsigned)(long long)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    if (sink == 0x12345678u) out[blockIdx.x * 2 + 1] = sink;
}

template <int OP, int ILP>
void run(const char* name, unsigned long long* d_out)
{
    const int iters = 200;
    const double n_inst = 64.0 * iters;
    printf("%-18s ilp=%d :", name, ILP);
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 1024 * wps;          // 64-thread blocks: wps waves per SIMD on 256 CUs x 4 SIMDs
        hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(64), 0, 0, d_out, iters, 12345u);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * blocks * 2, hipMemcpyDeviceToHost);
        double sum = 0;
        for (int b = 0; b < blocks; ++b) sum += (double)h[2 * b];
        double cyc = sum / blocks / n_inst;                 // s_memtime ticks (100 MHz?) vs shader cycles: report raw
        printf("  w%d: %7.2f (x%d waves = %6.2f/SIMD-inst)", wps, cyc, wps, cyc / wps);
    }
    printf("\n");
}

int main()
{
    unsigned long long* d_out;
    hipMalloc(&d_out, sizeof(unsigned long long) * 2 * 1024 * 8);
    printf("cycles (s_memtime ticks) per wave-instruction; 'per SIMD-inst' = ticks / waves-per-SIMD\n");
#define RUN(op, name) run<op, 1>(name, d_out); run<op, 4>(name, d_out);
    RUN(4, "v_fma_f32") RUN(3, "v_xor_b32") RUN(10, "v_cndmask_b32") RUN(0, "v_mad_u64_u32+mov") RUN(1, "v_mul_hi_u32") RUN(2, "v_mul_lo_u32")
    RUN(5, "v_sqrt_f32") RUN(6, "v_rcp_f32") RUN(7, "v_fma_f64") RUN(8, "v_add_f64") RUN(11, "v_pk_fma_f32") RUN(9, "ds_bpermute_b32") RUN(12, "v_mov_dpp")
    return 0;
}
