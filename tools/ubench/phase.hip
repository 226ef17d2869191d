// Diagnostic microbenchmark (not shipped): does the 4-byte PHASE of 64-bit instruction encodings matter to a lone wave?
// One wave per SIMD; a loop of 256 v_fma_f64 (8-byte VOP3 encodings, eight independent accumulators), shifted by PAD 4-byte
// s_nops in front of the loop label; and the same with every fourth instruction a 4-byte v_fmac_f64_e32 (the phase then
// alternates along the stream, as in compiler output).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define F8 "v_fma_f64 %[a0], %[a0], %[m], %[c]\n v_fma_f64 %[a1], %[a1], %[m], %[c]\n v_fma_f64 %[a2], %[a2], %[m], %[c]\n v_fma_f64 %[a3], %[a3], %[m], %[c]\n" \
           "v_fma_f64 %[a4], %[a4], %[m], %[c]\n v_fma_f64 %[a5], %[a5], %[m], %[c]\n v_fma_f64 %[a6], %[a6], %[m], %[c]\n v_fma_f64 %[a7], %[a7], %[m], %[c]\n"
#define M8 "v_fma_f64 %[a0], %[a0], %[m], %[c]\n v_fma_f64 %[a1], %[a1], %[m], %[c]\n v_fma_f64 %[a2], %[a2], %[m], %[c]\n v_fmac_f64_e32 %[a3], %[m], %[c]\n" \
           "v_fma_f64 %[a4], %[a4], %[m], %[c]\n v_fma_f64 %[a5], %[a5], %[m], %[c]\n v_fma_f64 %[a6], %[a6], %[m], %[c]\n v_fmac_f64_e32 %[a7], %[m], %[c]\n"

template <int MIXED, int PAD>
__global__ __launch_bounds__(1024) void k(double* out, int iters, long long* ticks) {
    const int lane = threadIdx.x;
    double a0 = 1.0 + lane, a1 = 2.0 + lane, a2 = 3.0 + lane, a3 = 4.0 + lane, a4 = 5.0 + lane, a5 = 6.0 + lane, a6 = 7.0 + lane, a7 = 8.0 + lane;
    double m = 0.999999 + 1e-12 * lane, c = 1e-9 + 1e-15 * lane;
    int n = iters;
    long long t0 = __builtin_readcyclecounter();
#define LOOP(B) asm volatile(".rept %c[pad]\n s_nop 0\n .endr\n 1:\n .rept 32\n" B ".endr\n s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n" \
                             : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7), [n] "+s"(n) \
                             : [m] "v"(m), [c] "v"(c), [pad] "n"(PAD) : "scc")
    if (MIXED) LOOP(M8); else LOOP(F8);
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static int g_waves = 1;  // waves per SIMD
template <int MIXED, int PAD>
double run(double* out, long long* ticks, int iters) {
    k<MIXED, PAD><<<256, 256 * g_waves>>>(out, iters, ticks);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<MIXED, PAD><<<256, 256 * g_waves>>>(out, iters, ticks);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    (void)hipMemcpy(h.data(), ticks, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (long long v : h) s += (double)v;
    printf("  waves/SIMD %d mixed %d pad %d: %.3f counter ticks per instruction of a wave, %.3f ns per instruction (wall)\n", g_waves, MIXED, PAD, s / 256.0 / iters / 256.0, ms * 1e6 / iters / 256.0);
    return s;
}

int main() {
    double* out; long long* ticks;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(double));
    (void)hipMalloc(&ticks, 256 * sizeof(long long));
    const int iters = 4000;
    run<0, 0>(out, ticks, iters); run<0, 1>(out, ticks, iters); run<0, 2>(out, ticks, iters); run<0, 3>(out, ticks, iters);
    run<0, 4>(out, ticks, iters); run<0, 5>(out, ticks, iters); run<0, 8>(out, ticks, iters); run<0, 9>(out, ticks, iters);
    run<1, 0>(out, ticks, iters); run<1, 1>(out, ticks, iters); run<1, 2>(out, ticks, iters); run<1, 3>(out, ticks, iters);
    for (g_waves = 2; g_waves <= 4; g_waves += 2) {  // with company on the SIMD: per-wave ticks double, the phase effect?
        run<0, 0>(out, ticks, iters); run<0, 1>(out, ticks, iters); run<0, 2>(out, ticks, iters); run<0, 3>(out, ticks, iters);
    }
    return 0;
}
