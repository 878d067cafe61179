// Microbenchmark (diagnostic, not product): cost of the instruction patterns of the row-per-lane tree phases, one wave alone on its SIMD.
// hipcc --offload-arch=gfx950 -O3 tree_cost.hip -o tree_cost && ./tree_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x
__device__ __forceinline__ long long now() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define BLK(acc, src) "s_nop 1\n\t" \
    "v_fmac_f64_dpp " acc ", " src ", %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " acc ", " src ", %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " acc ", " src ", %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " acc ", " src ", %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " acc ", " src ", %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f64_dpp " acc ", " src ", %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t"
__global__ void k(long long *out, double *sink)
{
    __shared__ double L[4096];
    const int lane = threadIdx.x;
    double a0 = lane * 0.5, a1 = 1.0 + lane, a2 = 2.0, a3 = 3.0, a4 = 4, a5 = 5, a6 = 6, a7 = 7, m = 1.0000001, s = lane * 0.25 + 1.0;
    for (int i = lane; i < 4096; i += 64) L[i] = i * 0.001;
    __syncthreads();
    long long t[24];
    int ti = 0;
    t[ti++] = now(); t[ti++] = now();
    // 1: 8 blocks of 6 DEPENDENT dpp fmacs (one accumulator per block, sources written before): the CRBA pattern
    t[ti] = now();
    asm volatile(BLK("%0", "%1") BLK("%2", "%3") BLK("%4", "%5") BLK("%6", "%7") BLK("%0", "%1") BLK("%2", "%3") BLK("%4", "%5") BLK("%6", "%7")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(m));
    t[++ti] = now(); ti++;
    // 2: chained blocks: each block's source is the previous block's accumulator (the sweep pattern: v -> ag)
    t[ti] = now();
    asm volatile(BLK("%0", "%1") BLK("%2", "%0") BLK("%4", "%2") BLK("%6", "%4") BLK("%1", "%6") BLK("%3", "%1") BLK("%5", "%3") BLK("%7", "%5")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(m));
    t[++ti] = now(); ti++;
    // 3: the 18 LDS reads of one CRBA iteration: 6 b64 at per-lane addresses, 6 b128 + 6 read2_b64 at row-uniform addresses, then wait
    int rowa = (lane >> 4) * 288 * 8, lanea = rowa + (lane & 7) * 8;
    double r0, r1, r2, r3, r4, r5;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 q0, q1, q2, q3, q4, q5, p0, p1, p2, p3, p4, p5;
    t[ti] = now();
    asm volatile("ds_read_b64 %0, %18\n\tds_read_b64 %1, %18 offset:48\n\tds_read_b64 %2, %18 offset:96\n\tds_read_b64 %3, %18 offset:144\n\tds_read_b64 %4, %18 offset:192\n\tds_read_b64 %5, %18 offset:240\n\t"
                 "ds_read_b128 %6, %19 offset:0\n\tds_read_b128 %7, %19 offset:48\n\tds_read_b128 %8, %19 offset:96\n\tds_read_b128 %9, %19 offset:144\n\tds_read_b128 %10, %19 offset:192\n\tds_read_b128 %11, %19 offset:240\n\t"
                 "ds_read2_b64 %12, %19 offset0:2 offset1:8\n\tds_read2_b64 %13, %19 offset0:14 offset1:20\n\tds_read2_b64 %14, %19 offset0:26 offset1:32\n\t"
                 "ds_read2_b64 %15, %19 offset0:3 offset1:9\n\tds_read2_b64 %16, %19 offset0:15 offset1:21\n\tds_read2_b64 %17, %19 offset0:27 offset1:33\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5),
                   "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3), "=v"(p4), "=v"(p5) : "v"(lanea), "v"(rowa));
    t[++ti] = now(); ti++;
    // 4: 8 exec-masked LDS stores (s_and_saveexec / ds_write / s_or)
    t[ti] = now();
    asm volatile(REP8("s_and_saveexec_b64 s[40:41], %1\n\tds_write_b64 %2, %0\n\ts_or_b64 exec, exec, s[40:41]\n\t") "s_waitcnt lgkmcnt(0)" :: "v"(a0), "s"((long long)0x3f3f3f3f3f3f3f3full), "v"(lanea) : "s40", "s41", "memory");
    t[++ti] = now(); ti++;
    // 5: 27 plain FMAs with 6 accumulators (the Y product)
    t[ti] = now();
    asm volatile(REP4("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\tv_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\t") "v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(m));
    t[++ti] = now(); ti++;
    // 6: 16 x v_mov_b32_dpp row_shl:1 (the crm(v) S exchange)
    int i0 = lane, i1 = lane + 1;
    t[ti] = now();
    asm volatile(REP8("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") : "+v"(i0), "+v"(i1));
    t[++ti] = now(); ti++;
    if (lane == 0) for (int i = 0; i < 24; i++) out[i] = t[i];
    sink[lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s + r0 + r1 + r2 + r3 + r4 + r5 + q0.x + q1.y + q2.x + q3.y + q4.x + q5.y + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + i0 + i1;
}
int main()
{
    long long *d; double *sink;
    (void)hipMalloc(&d, 24 * 8); (void)hipMalloc(&sink, 64 * 8);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sink);
    std::vector<long long> h(24);
    (void)hipMemcpy(h.data(), d, 24 * 8, hipMemcpyDeviceToHost);
    const char *nm[] = {"empty", "8 blocks x 6 dependent v_fmac_f64_dpp", "8 chained blocks (src = previous acc)", "18 LDS reads of a CRBA iteration + wait", "8 exec-masked ds_write_b64", "27 v_fma_f64, 6 accumulators", "16 v_mov_b32_dpp row_shl/shr (dependent)"};
    const int cnt[] = {1, 48, 48, 18, 8, 27, 16};
    const long long base = h[1] - h[0];
    for (int i = 0; i < 7; i++) printf("%-44s total %6lld  -> %.1f cycles each\n", nm[i], h[2 * i + 1] - h[2 * i], (double)(h[2 * i + 1] - h[2 * i] - base) / cnt[i]);
    return 0;
}
