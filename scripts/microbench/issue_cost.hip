// Microbenchmark (diagnostic, not product): issue cost / dependent latency of the instruction forms the LDL' kernels are made of,
// one wave alone on its SIMD.  hipcc --offload-arch=gfx950 -O3 issue_cost.hip -o issue_cost && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__device__ __forceinline__ long long now() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__global__ void k(long long *out, double *sink)
{
    __shared__ double L[256];
    const int lane = threadIdx.x;
    double a0 = lane * 0.5, a1 = 1.0 + lane, a2 = 2.0, a3 = 3.0, a4 = 4, a5 = 5, a6 = 6, a7 = 7, m = 1.0000001, s = lane * 0.25 + 1.0;
    L[lane] = a0; L[lane + 64] = a1;
    __syncthreads();
    long long t[20];
    int ti = 0;
    // 0: empty
    t[ti++] = now(); t[ti++] = now();
    // 1: 64 independent v_fma_f64 (8 accumulators)
    t[ti] = now();
    asm volatile(REP8("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\tv_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(m));
    t[++ti] = now(); ti++;
    // 2: 64 dependent v_fma_f64 (one accumulator)
    t[ti] = now();
    asm volatile(REP64("v_fma_f64 %0, %1, %2, %0\n\t") : "+v"(a0) : "v"(s), "v"(m));
    t[++ti] = now(); ti++;
    // 3: 64 independent v_fmac_f64_dpp row_newbcast
    t[ti] = now();
    asm volatile("s_nop 1\n\t" REP8("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(m));
    t[++ti] = now(); ti++;
    // 4: 64 x (2 v_readlane_b32 + v_fma_f64 with SGPR operand), independent accumulators
    t[ti] = now();
    asm volatile(REP8("v_readlane_b32 s40, %8, 3\n\tv_readlane_b32 s41, %9, 3\n\tv_fma_f64 %0, s[40:41], %10, %0\n\t"
                      "v_readlane_b32 s42, %8, 4\n\tv_readlane_b32 s43, %9, 4\n\tv_fma_f64 %1, s[42:43], %10, %1\n\t"
                      "v_readlane_b32 s44, %8, 5\n\tv_readlane_b32 s45, %9, 5\n\tv_fma_f64 %2, s[44:45], %10, %2\n\t"
                      "v_readlane_b32 s46, %8, 6\n\tv_readlane_b32 s47, %9, 6\n\tv_fma_f64 %3, s[46:47], %10, %3\n\t"
                      "v_readlane_b32 s40, %8, 7\n\tv_readlane_b32 s41, %9, 7\n\tv_fma_f64 %4, s[40:41], %10, %4\n\t"
                      "v_readlane_b32 s42, %8, 8\n\tv_readlane_b32 s43, %9, 8\n\tv_fma_f64 %5, s[42:43], %10, %5\n\t"
                      "v_readlane_b32 s44, %8, 9\n\tv_readlane_b32 s45, %9, 9\n\tv_fma_f64 %6, s[44:45], %10, %6\n\t"
                      "v_readlane_b32 s46, %8, 10\n\tv_readlane_b32 s47, %9, 10\n\tv_fma_f64 %7, s[46:47], %10, %7\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(__double2loint(s)), "v"(__double2hiint(s)), "v"(m) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    t[++ti] = now(); ti++;
    // 5: 64 v_mov_b64_dpp row_newbcast (independent destinations)
    t[ti] = now();
    asm volatile("s_nop 1\n\t" REP8("v_mov_b64_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %4, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %6, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
    t[++ti] = now(); ti++;
    // 6: 16 dependent v_rcp_f64
    t[ti] = now();
    asm volatile(REP8("v_rcp_f64 %0, %0\n\tv_rcp_f64 %0, %0\n\t") : "+v"(s));
    t[++ti] = now(); ti++;
    // 7: 16 dependent ds_read_b64 (address from the loaded value's low bits)
    int addr = (lane & 63) * 8;
    t[ti] = now();
    asm volatile(REP8("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0x1f8, %1\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %1, 0x1f8, %1\n\t") : "+v"(a7), "+v"(addr));
    t[++ti] = now(); ti++;
    // 8: 16 dependent v_mfma_f64_16x16x4_f64 (one accumulator)
    typedef double v4d __attribute__((ext_vector_type(4)));
    v4d acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    t[ti] = now();
#pragma unroll
    for (int i = 0; i < 16; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, s, acc, 0, 0, 0);
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
    t[++ti] = now(); ti++;
    // 9: 16 v_mfma_f64_16x16x4_f64 on two alternating accumulators
    t[ti] = now();
#pragma unroll
    for (int i = 0; i < 8; i++) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, s, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, s, acc2, 0, 0, 0); }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc), "+v"(acc2));
    t[++ti] = now(); ti++;
    a0 += acc[0] + acc[1] + acc2[2] + acc2[3];
    if (lane == 0) for (int i = 0; i < 20; i++) out[i] = t[i];
    sink[lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s + addr;
}

int main()
{
    long long *d; double *sink;
    hipMalloc(&d, 20 * 8); hipMalloc(&sink, 64 * 8);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sink);
    std::vector<long long> h(20);
    hipMemcpy(h.data(), d, 20 * 8, hipMemcpyDeviceToHost);
    const char *nm[] = {"empty", "64 indep v_fma_f64", "64 dep v_fma_f64", "64 indep v_fmac_f64_dpp row_newbcast", "64 x (2 readlane + fma sgpr)", "64 v_mov_b64_dpp", "16 dep v_rcp_f64", "16 dep ds_read_b64", "16 dep v_mfma_f64_16x16x4 (+32 nop)", "16 v_mfma_f64_16x16x4, 2 accumulators (+32 nop)"};
    const int cnt[] = {1, 64, 64, 64, 64, 64, 16, 16, 16, 16};
    const long long base = h[1] - h[0];
    for (int i = 0; i < 10; i++) printf("%-40s total %6lld  -> %.1f cycles each\n", nm[i], h[2 * i + 1] - h[2 * i], (double)(h[2 * i + 1] - h[2 * i] - base) / cnt[i]);
    return 0;
}
