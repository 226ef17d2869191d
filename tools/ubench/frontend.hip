// Diagnostic microbenchmark (not shipped): what the control code around the RK stages costs a LONE wave on gfx950.
// One wave per SIMD (256 blocks of 256 threads), a timed loop written as one asm block so that nothing is rescheduled:
// every iteration is 8 groups of [4 independent v_fma_f64 + ONE probe], and the report is cycles per group minus the
// plain group's, i.e. the cost of the probe in the shadow of four FP64 instructions -- the situation of the evaluation
// kernel's loop control.  PAD shifts the loop by 4-byte s_nops: placement sensitivity of each probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMA4 "v_fma_f64 %[a0], %[a0], %[m], %[c]\n v_fma_f64 %[a1], %[a1], %[m], %[c]\n v_fma_f64 %[a2], %[a2], %[m], %[c]\n v_fma_f64 %[a3], %[a3], %[m], %[c]\n"
#define G8(P) FMA4 P FMA4 P FMA4 P FMA4 P FMA4 P FMA4 P FMA4 P FMA4 P

enum { PLAIN, BR_TAKEN, BR_NOT_TAKEN, VCMP_BRANCH, LDS_WRITE_WAIT, LDS_WRITE, LDS_READ_WAIT, SAVEEXEC, READLANE, WAITCNT_NOP, CNDMASK2, SALU2, VCMP_SAND,
       LDS_WRITE_FLAG_WAIT, BR_TAKEN_FAR, N_PROBES };
static const char* NAMES[] = {"plain (4 fma)", "s_branch taken (over 1 nop)", "s_cmp + s_cbranch not taken", "v_cmp->s_and->s_cmp->s_cbranch (not taken)",
                              "ds_write_b64 + s_waitcnt", "ds_write_b64 (no wait)", "ds_read_b64 + s_waitcnt", "s_and_saveexec + s_or exec", "v_readlane_b32",
                              "s_waitcnt (nothing pending)", "2 x v_cndmask_b32", "2 x s_and_b64", "v_cmp + s_and (no branch)",
                              "ds_write_b64, waitcnt, ds_write_b32 flag, waitcnt", "s_branch taken over 40 instructions"};

template <int PROBE, int PAD>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* ticks) {
    __shared__ double lds[512];
    const int lane = threadIdx.x;
    double a0 = 1.0 + lane, a1 = 2.0 + lane, a2 = 3.0 + lane, a3 = 4.0 + lane;
    double m = 0.999999 + 1e-12 * lane, c = 1e-9 + 1e-15 * lane, big = 1e300;
    unsigned long long sm = 0, msk = 0xffffffff0000ffffull;
    int n = iters, sr = 0;
    unsigned addr = (unsigned)(lane * 8);
    double ld = 0.0;
    int v32a = lane, v32b = lane + 1;
    lds[lane] = lane;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
#define LOOP(P) asm volatile(".rept %c[pad]\n s_nop 0\n .endr\n 1:\n" G8(P) "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n" \
                             : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [n] "+s"(n), [sm] "+s"(sm), [sr] "+s"(sr), [ld] "+v"(ld), [va] "+v"(v32a), [vb] "+v"(v32b) \
                             : [m] "v"(m), [c] "v"(c), [big] "v"(big), [msk] "s"(msk), [addr] "v"(addr), [pad] "n"(PAD) : "vcc", "scc", "memory")
    if (PROBE == PLAIN) LOOP("");
    else if (PROBE == BR_TAKEN) LOOP("s_branch 2f\n s_nop 0\n 2:\n");
    else if (PROBE == BR_TAKEN_FAR) LOOP("s_branch 2f\n .rept 40\n s_nop 0\n .endr\n 2:\n");
    else if (PROBE == BR_NOT_TAKEN) LOOP("s_cmp_eq_u32 %[n], -1\n s_cbranch_scc1 3f\n");
    else if (PROBE == VCMP_BRANCH) LOOP("v_cmp_gt_f64 vcc, %[a0], %[big]\n s_and_b64 %[sm], vcc, exec\n s_cmp_lg_u64 %[sm], 0\n s_cbranch_scc1 3f\n");
    else if (PROBE == VCMP_SAND) LOOP("v_cmp_gt_f64 vcc, %[a0], %[big]\n s_and_b64 %[sm], vcc, exec\n");
    else if (PROBE == LDS_WRITE_WAIT) LOOP("ds_write_b64 %[addr], %[a0]\n s_waitcnt lgkmcnt(0)\n");
    else if (PROBE == LDS_WRITE) LOOP("ds_write_b64 %[addr], %[a0]\n");
    else if (PROBE == LDS_WRITE_FLAG_WAIT) LOOP("ds_write_b64 %[addr], %[a0]\n s_waitcnt lgkmcnt(0)\n ds_write_b32 %[addr], %[va] offset:2048\n s_waitcnt lgkmcnt(0)\n");
    else if (PROBE == LDS_READ_WAIT) LOOP("ds_read_b64 %[ld], %[addr]\n s_waitcnt lgkmcnt(0)\n");
    else if (PROBE == SAVEEXEC) LOOP("s_and_saveexec_b64 %[sm], %[msk]\n s_or_b64 exec, exec, %[sm]\n");
    else if (PROBE == READLANE) LOOP("v_readlane_b32 %[sr], %[va], 3\n");
    else if (PROBE == WAITCNT_NOP) LOOP("s_waitcnt lgkmcnt(0)\n");
    else if (PROBE == CNDMASK2) LOOP("v_cndmask_b32 %[va], %[va], %[vb], %[msk]\n v_cndmask_b32 %[vb], %[vb], %[va], %[msk]\n");
    else if (PROBE == SALU2) LOOP("s_and_b64 %[sm], %[sm], %[msk]\n s_and_b64 %[sm], %[sm], %[msk]\n");
    asm volatile("3:\n s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + ld + (double)sm + sr + v32a + v32b;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int PROBE, int PAD>
double run(double* out, long long* ticks, int iters) {
    k<PROBE, PAD><<<256, 256>>>(out, iters, ticks);
    (void)hipDeviceSynchronize();
    k<PROBE, PAD><<<256, 256>>>(out, iters, ticks);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(256);
    (void)hipMemcpy(h.data(), ticks, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (long long v : h) s += (double)v;
    return s / 256.0 / iters / 8.0;  // cycles per group
}

template <int PROBE>
void row(double* out, long long* ticks, int iters, double plain0) {
    const double r[8] = {run<PROBE, 0>(out, ticks, iters), run<PROBE, 1>(out, ticks, iters), run<PROBE, 2>(out, ticks, iters), run<PROBE, 3>(out, ticks, iters),
                         run<PROBE, 5>(out, ticks, iters), run<PROBE, 7>(out, ticks, iters), run<PROBE, 10>(out, ticks, iters), run<PROBE, 13>(out, ticks, iters)};
    printf("%-52s", NAMES[PROBE]);
    for (double v : r) printf(" %6.1f", v - plain0);
    printf("   (cycles per probe beyond the 4 fma = %.1f; pads 0 1 2 3 5 7 10 13)\n", plain0);
}

int main() {
    double* out; long long* ticks;
    (void)hipMalloc(&out, 256 * 256 * sizeof(double));
    (void)hipMalloc(&ticks, 256 * sizeof(long long));
    const int iters = 20000;
    const double plain0 = run<PLAIN, 0>(out, ticks, iters);
    row<PLAIN>(out, ticks, iters, plain0);
    row<BR_TAKEN>(out, ticks, iters, plain0);
    row<BR_TAKEN_FAR>(out, ticks, iters, plain0);
    row<BR_NOT_TAKEN>(out, ticks, iters, plain0);
    row<VCMP_BRANCH>(out, ticks, iters, plain0);
    row<VCMP_SAND>(out, ticks, iters, plain0);
    row<LDS_WRITE_WAIT>(out, ticks, iters, plain0);
    row<LDS_WRITE>(out, ticks, iters, plain0);
    row<LDS_WRITE_FLAG_WAIT>(out, ticks, iters, plain0);
    row<LDS_READ_WAIT>(out, ticks, iters, plain0);
    row<SAVEEXEC>(out, ticks, iters, plain0);
    row<READLANE>(out, ticks, iters, plain0);
    row<WAITCNT_NOP>(out, ticks, iters, plain0);
    row<CNDMASK2>(out, ticks, iters, plain0);
    row<SALU2>(out, ticks, iters, plain0);
    return 0;
}
