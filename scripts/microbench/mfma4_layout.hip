// Probe (diagnostic): operand / result lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950.
// For every B lane lb: b = delta(lane == lb), a[l] = l + 1  ->  D[lane] = (1 + A-lane paired with B-lane lb for output lane), 0 if lb does not feed it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(double *out)
{
    const int lane = threadIdx.x;
    for (int lb = 0; lb < 64; lb++) {
        const double a = lane + 1.0, b = (lane == lb) ? 1.0 : 0.0;
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[64 * lb + lane] = d;
    }
}
int main()
{
    double *d; (void)hipMalloc(&d, 64 * 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64);
    (void)hipMemcpy(h.data(), d, 64 * 64 * 8, hipMemcpyDeviceToHost);
    // for each output lane: list of (A lane, B lane) pairs
    for (int lane = 0; lane < 64; lane++) {
        printf("D lane %2d <-", lane);
        for (int lb = 0; lb < 64; lb++) if (h[64 * lb + lane] != 0.0) printf(" (a%d,b%d)", (int)h[64 * lb + lane] - 1, lb);
        printf("\n");
    }
    return 0;
}
