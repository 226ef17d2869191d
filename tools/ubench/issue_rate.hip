// Diagnostic microbenchmark (not shipped): per-wave issue interval of FP64 / FP32 VALU
// instructions on gfx950 as a function of waves per SIMD.  256 blocks (one per CU) of
// 256*W threads = W waves on every SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
enum { OP_FMA64, OP_MUL64, OP_ADD64, OP_FMA64_SGPR, OP_FMA32, OP_MOV32, OP_MIX, OP_MAX64, OP_DPP_MOV, OP_FMA64_DEP1, OP_FMA64_DEP2, OP_FMA64_DEP3, OP_FMAC_DPP64, OP_PK_FMA32, OP_FMA32_DEP1, OP_MIX32, OP_FMA64_REGS };

template <int OP>
__global__ void k(double* out, int iters, long long* ticks) {
    long long t0 = __builtin_readcyclecounter(); long long w0 = wall_clock64();
    int lane = threadIdx.x;
    double a0 = 1.0 + lane, a1 = 2.0 + lane, a2 = 3.0 + lane, a3 = 4.0 + lane;
    double a4 = 5.0 + lane, a5 = 6.0 + lane, a6 = 7.0 + lane, a7 = 8.0 + lane;
    double m = 0.999999 + 1e-12 * lane, c = 1e-9 + 1e-15 * lane;
    double b0 = m, b1 = m + 1e-13, b2 = m + 2e-13, b3 = m + 3e-13, b4 = c, b5 = c * 1.5, b6 = c * 2.5, b7 = c * 3.5;
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3, f4 = lane + 4, f5 = lane + 5, f6 = lane + 6, f7 = lane + 7;
    float fm = 0.9999f + 1e-6f * lane, fc = 1e-5f * lane;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        if (OP == OP_FMA64) {
#define F(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else if (OP == OP_MUL64) {
#define F(a) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(m));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else if (OP == OP_ADD64) {
#define F(a) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else if (OP == OP_MAX64) {
#define F(a) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else if (OP == OP_FMA64_SGPR) {
            double sm = 0.999999;
#define F(a) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(a) : "s"(sm), "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else if (OP == OP_FMA32) {
#define F(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(fm), "v"(fc));
            F(f0) F(f1) F(f2) F(f3) F(f4) F(f5) F(f6) F(f7)
#undef F
        } else if (OP == OP_MOV32) {
#define F(a, b) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
            F(f0, f1) F(f1, f2) F(f2, f3) F(f3, f4) F(f4, f5) F(f5, f6) F(f6, f7) F(f7, fm)
#undef F
        } else if (OP == OP_DPP_MOV) {
#define F(a, b) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
            F(f0, f1) F(f1, f2) F(f2, f3) F(f3, f4) F(f4, f5) F(f5, f6) F(f6, f7) F(f7, fm)
#undef F
        } else if (OP == OP_FMA64_DEP1) {  // one dependent chain: the result latency, not the issue interval
#define F(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
            F(a0) F(a0) F(a0) F(a0) F(a0) F(a0) F(a0) F(a0)
#undef F
        } else if (OP == OP_FMA64_DEP2) {
#define F(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
            F(a0) F(a1) F(a0) F(a1) F(a0) F(a1) F(a0) F(a1)
#undef F
        } else if (OP == OP_FMA64_DEP3) {
#define F(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
            F(a0) F(a1) F(a2) F(a0) F(a1) F(a2) F(a0) F(a1)
#undef F
        } else if (OP == OP_FMAC_DPP64) {  // v_fmac_f64 with a row_newbcast source: same rate as a plain fmac?
#define F(a) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(m), "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else if (OP == OP_PK_FMA32) {  // two FP32 FMAs per lane and instruction: twice the flops per issue slot?
#define F(a) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else if (OP == OP_FMA32_DEP1) {
#define F(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(fm), "v"(fc));
            F(f0) F(f0) F(f0) F(f0) F(f0) F(f0) F(f0) F(f0)
#undef F
        } else if (OP == OP_MIX32) {  // FP32 fma interleaved with FP64 fma (the fp32 arm's inline likelihood)
#define F(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(fm), "v"(fc));
#define G(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
            F(f0) F(f1) F(f2) G(a0) F(f3) F(f4) F(f5) G(a1)
#undef F
#undef G
        } else if (OP == OP_FMA64_REGS) {  // three DIFFERENT register pairs per fma, sixteen source pairs in rotation
#define F(a, b, cc) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(cc));
            F(a0, b0, b5) F(a1, b1, b6) F(a2, b2, b7) F(a3, b3, b4) F(a4, b4, b1) F(a5, b5, b2) F(a6, b6, b3) F(a7, b7, b0)
#undef F
        } else if (OP == OP_MIX) {
            // alternate FP64 fma and 32-bit moves: does a 32-bit op hide behind an FP64 op?
#define F(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
#define G(a, b) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
            F(a0) G(f0, f1) F(a1) G(f1, f2) F(a2) G(f2, f3) F(a3) G(f3, f4)
#undef F
#undef G
        }
      }
    }
    long long t1 = __builtin_readcyclecounter(); long long w1 = wall_clock64();
    if (blockIdx.x == 0 && lane == 0) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
    out[blockIdx.x * blockDim.x + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int OP>
void run(const char* name, double* out) {
    const int iters = 1 << 15;
    static long long* ticks = nullptr; if (!ticks) (void)hipMalloc(&ticks, 16);
    for (int w = 1; w <= 4; w *= 2) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<OP><<<256, 256 * w>>>(out, iters, ticks); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k<OP><<<256, 256 * w>>>(out, iters, ticks); (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize(); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double per_wave = ms * 1e6 / (iters * 256.0);          // ns per instruction as seen by one wave
        double per_simd = per_wave / w;                       // ns per instruction per SIMD
        long long h[2]; (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
        printf("%-12s waves/SIMD=%d  %.1f ms  per-wave %.2f ns/instr  per-SIMD %.2f ns/instr | memtime %.2f ticks/instr (tick rate %.0f MHz)\n",
               name, w, ms, per_wave, per_simd, (double)h[0] / (iters * 256.0), (double)h[0] / ((double)h[1] / 100.0));
    }
}

int main() {
    double* out; (void)hipMalloc(&out, 8 * 256 * 1024);
    run<OP_FMA64>("fma64", out); run<OP_MUL64>("mul64", out); run<OP_ADD64>("add64", out); run<OP_MAX64>("max64", out);
    run<OP_FMA64_SGPR>("fma64_sgpr", out); run<OP_FMA32>("fma32", out); run<OP_MOV32>("mov32", out);
    run<OP_DPP_MOV>("mov32_dpp", out); run<OP_MIX>("fma64+mov32", out);
    run<OP_FMA64_DEP1>("fma64_dep1", out); run<OP_FMA64_DEP2>("fma64_dep2", out); run<OP_FMA64_DEP3>("fma64_dep3", out);
    run<OP_FMAC_DPP64>("fmac64_dpp", out);
    run<OP_PK_FMA32>("pk_fma32", out); run<OP_FMA32_DEP1>("fma32_dep1", out); run<OP_MIX32>("3fma32+fma64", out);
    run<OP_FMA64_REGS>("fma64_regs", out);
    return 0;
}
