// Microbenchmark (diagnostic): issue cost and dependent latency of v_mfma_f64_4x4x4_4b_f64, one wave alone on its SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ long long now() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__global__ void k(long long *out, double *sink)
{
    const int lane = threadIdx.x;
    double a = lane * 0.01 + 1.0, b = 1.0 - lane * 0.001;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    long long t[16]; int ti = 0;
    t[ti++] = now(); t[ti++] = now();
    // 1: 32 independent (8 accumulators x 4)
    t[ti] = now();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    }
    asm volatile("s_nop 15" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
    t[++ti] = now(); ti++;
    // 2: 16 dependent through C
    t[ti] = now();
#pragma unroll
    for (int i = 0; i < 16; i++) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    asm volatile("s_nop 15" : "+v"(c0));
    t[++ti] = now(); ti++;
    // 3: 16 dependent through B (result feeds the next B operand: the chained products of the tree recursions)
    t[ti] = now();
#pragma unroll
    for (int i = 0; i < 16; i++) c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c1, 0.0, 0, 0, 0);
    asm volatile("s_nop 15" : "+v"(c1));
    t[++ti] = now(); ti++;
    // 4: 16 dependent through A
    t[ti] = now();
#pragma unroll
    for (int i = 0; i < 16; i++) c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(c2, b, 0.0, 0, 0, 0);
    asm volatile("s_nop 15" : "+v"(c2));
    t[++ti] = now(); ti++;
    // 5: 2 x 2 x 2 tile product (8 mfma: 4 output tiles, 2 k-steps each), then a second one that consumes its result as B: one tree level
    double y00 = 0, y01 = 0, y10 = 0, y11 = 0, z00 = 0, z01 = 0, z10 = 0, z11 = 0;
    t[ti] = now();
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
        y00 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0); y01 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c3, 0.0, 0, 0, 0);
        y10 = __builtin_amdgcn_mfma_f64_4x4x4f64(c4, b, 0.0, 0, 0, 0); y11 = __builtin_amdgcn_mfma_f64_4x4x4f64(c4, c3, 0.0, 0, 0, 0);
        y00 = __builtin_amdgcn_mfma_f64_4x4x4f64(c5, c6, y00, 0, 0, 0); y01 = __builtin_amdgcn_mfma_f64_4x4x4f64(c5, c7, y01, 0, 0, 0);
        y10 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c6, y10, 0, 0, 0); y11 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c7, y11, 0, 0, 0);
        z00 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, y00, 0.0, 0, 0, 0); z01 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, y01, 0.0, 0, 0, 0);
        z10 = __builtin_amdgcn_mfma_f64_4x4x4f64(c3, y00, 0.0, 0, 0, 0); z11 = __builtin_amdgcn_mfma_f64_4x4x4f64(c3, y01, 0.0, 0, 0, 0);
        z00 = __builtin_amdgcn_mfma_f64_4x4x4f64(c6, y10, z00, 0, 0, 0); z01 = __builtin_amdgcn_mfma_f64_4x4x4f64(c6, y11, z01, 0, 0, 0);
        z10 = __builtin_amdgcn_mfma_f64_4x4x4f64(c7, y10, z10, 0, 0, 0); z11 = __builtin_amdgcn_mfma_f64_4x4x4f64(c7, y11, z11, 0, 0, 0);
        a = z00 + 1.0; c4 = z10; c5 = z01; c6 = z11;              // next level depends on this one
    }
    asm volatile("s_nop 15" : "+v"(a), "+v"(c4), "+v"(c5), "+v"(c6));
    t[++ti] = now(); ti++;
    if (lane == 0) for (int i = 0; i < 16; i++) out[i] = t[i];
    sink[lane] = a + b + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
int main()
{
    long long *d; double *sink;
    (void)hipMalloc(&d, 16 * 8); (void)hipMalloc(&sink, 64 * 8);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sink);
    std::vector<long long> h(16);
    (void)hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
    const char *nm[] = {"empty", "32 independent mfma_f64_4x4x4", "16 dependent through C", "16 dependent through B", "16 dependent through A", "4 chained levels of 16 mfma"};
    const int cnt[] = {1, 32, 16, 16, 16, 64};
    const long long base = h[1] - h[0];
    for (int i = 0; i < 6; i++) printf("%-36s total %6lld  -> %.1f cycles each\n", nm[i], h[2 * i + 1] - h[2 * i], (double)(h[2 * i + 1] - h[2 * i] - base) / cnt[i]);
    return 0;
}
