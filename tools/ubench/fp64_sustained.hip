// Diagnostic microbenchmark (not shipped): the FP64 vector rate the chip SUSTAINS.  W waves on every SIMD run eight
// independent v_fma_f64 chains on non-trivial data for tens of milliseconds, launched back to back for two seconds;
// reported: TFLOP/s from HIP events and the clock the chip held (s_memtime over s_memrealtime, 100 MHz reference).
// The nominal peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s) assumes 2.4 GHz under load.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void fma_loop(double* out, const double* in, int iters, unsigned long long* clk) {
    const int gid = blockIdx.x * 64 + threadIdx.x;
    double a0 = in[gid], a1 = a0 + 0.125, a2 = a0 + 0.25, a3 = a0 + 0.375, a4 = a0 + 0.5, a5 = a0 + 0.625, a6 = a0 + 0.75, a7 = a0 + 0.875;
    const double m = 0.99999 + 1e-9 * (gid & 1023), c = 1e-5 * in[gid ^ 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#define F(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[gid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int max_blocks = 1024 * 8;
    double *d_in, *d_out;
    unsigned long long* d_clk;
    hipMalloc(&d_in, max_blocks * 64 * sizeof(double));
    hipMalloc(&d_out, max_blocks * 64 * sizeof(double));
    hipMalloc(&d_clk, max_blocks * 2 * sizeof(unsigned long long));
    std::vector<double> h(max_blocks * 64);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = 1.0 + (s >> 8) * (1.0 / 16777216.0); }
    hipMemcpy(d_in, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int W : {1, 2, 4, 8}) {
        const int blocks = 1024 * W;
        const int iters = 200000 / W;  // ~25.6 M fma per lane-chain set / W: tens of ms per launch
        float ms = 0;
        double total_ms = 0;
        int launches = 0;
        while (total_ms < 2000.0) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fma_loop, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters, d_clk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            total_ms += ms; ++launches;
        }
        std::vector<unsigned long long> clk(2 * blocks);
        hipMemcpy(clk.data(), d_clk, clk.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> ghz(blocks);
        for (int b = 0; b < blocks; ++b) ghz[b] = (double)clk[2 * b] / (double)clk[2 * b + 1] * 0.1;
        std::sort(ghz.begin(), ghz.end());
        const double flops = (double)blocks * 64 * (double)iters * 16 * 8 * 2;
        printf("{\"waves_per_simd\": %d, \"ms\": %.3f, \"launches\": %d, \"tflops\": %.2f, \"clock_ghz_median\": %.3f, \"cycles_per_fma_per_simd\": %.3f}\n", W, ms,
               launches, flops / (ms * 1e-3) / 1e12, ghz[blocks / 2], ghz[blocks / 2] * 1e9 * (ms * 1e-3) / ((double)iters * 16 * 8 * W));
    }
    return 0;
}
