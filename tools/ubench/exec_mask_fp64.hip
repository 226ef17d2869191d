// Diagnostic microbenchmark (not shipped): does a wave64 FP64 VALU instruction get cheaper
// when only one 16-lane quarter of EXEC is live?  Decides whether a "few chains per wave"
// layout could lower the per-attempt latency at small batches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int DEP>
__global__ void k(double* out, long long* cyc, unsigned long long mask, int iters) {
    int lane = threadIdx.x;
    double a0 = 1.0 + lane, a1 = 2.0 + lane, a2 = 3.0 + lane, a3 = 4.0 + lane;
    double a4 = 5.0 + lane, a5 = 6.0 + lane, a6 = 7.0 + lane, a7 = 8.0 + lane;
    double m = 0.999999, c = 1e-9;
    long long t0 = 0, t1 = 0;
    if ((mask >> lane) & 1ull) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            if (DEP) {
                a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
                a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
                a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
                a0 = __builtin_fma(a0, m, c); a0 = __builtin_fma(a0, m, c);
            } else {
                a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c);
                a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c);
                a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((mask >> lane) & 1ull) cyc[blockIdx.x * 64 + lane] = t1 - t0;
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8 * 4096); hipMalloc(&cyc, 64 * 8 * 4096);
    struct { const char* name; unsigned long long mask; } cases[] = {
        {"all64", ~0ull}, {"lanes0-15", 0xFFFFull}, {"lanes16-31", 0xFFFF0000ull},
        {"lanes0-31", 0xFFFFFFFFull}, {"every4th", 0x1111111111111111ull}, {"lane0", 1ull}};
    const int iters = 4096;
    for (int dep = 0; dep < 2; ++dep)
        for (auto& cs : cases) {
            hipMemset(cyc, 0, 64 * 8);
            for (int rep = 0; rep < 2; ++rep) {
                if (dep) k<1><<<1, 64>>>(out, cyc, cs.mask, iters);
                else k<0><<<1, 64>>>(out, cyc, cs.mask, iters);
                hipDeviceSynchronize();
            }
            long long h[64]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
            long long v = 0; for (int i = 0; i < 64; ++i) if (h[i] > v) v = h[i];
            printf("%s %-11s  %.2f clk/fma (counter units)\n", dep ? "dependent  " : "independent", cs.name,
                   (double)v / (iters * 8.0));
        }
    // wall-clock version: many waves, one per SIMD (1024 blocks), compare total time
    for (auto& cs : cases) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<0><<<1024, 64>>>(out, cyc, cs.mask, iters); hipDeviceSynchronize();
        hipEventRecord(e0); k<0><<<1024, 64>>>(out, cyc, cs.mask, iters * 8); hipEventRecord(e1);
        hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("wall 1024 waves %-11s %.3f ms  -> %.2f ns/fma-instr\n", cs.name, ms, ms * 1e6 / (iters * 64.0));
    }
    return 0;
}
