// Diagnostic: where do the 8 waves of a 512-thread workgroup land when the kernel's register allocation (256 VGPRs)
// lets a SIMD hold two waves?  Prints, for a few workgroups, the SIMD id (HW_REG_HW_ID bits 5:4) and CU id of each wave,
// and how many of all workgroups have exactly two waves on every SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/wave_placement.hip -o tools/ubench/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512, 2) void probe(int* out, int spin) {
    asm volatile("v_mov_b32 v255, 0" ::: "v255");  // 256-register allocation like the integrator
    const int wave = threadIdx.x / 64;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID, all 32 bits
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (int)hw;
}

int main() {
    const int blocks = 256;
    int* d;
    hipMalloc(&d, blocks * 8 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<int> h(blocks * 8);
    hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    int balanced = 0, rr = 0;
    for (int b = 0; b < blocks; ++b) {
        int cnt[4] = {0, 0, 0, 0};
        bool round_robin = true;
        for (int w = 0; w < 8; ++w) {
            const int simd = (h[b * 8 + w] >> 4) & 3;
            cnt[simd]++;
            if (simd != ((h[b * 8] >> 4) + w) % 4) round_robin = false;
        }
        if (cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2) ++balanced;
        if (round_robin) ++rr;
        if (b < 6) {
            printf("wg %d:", b);
            for (int w = 0; w < 8; ++w) printf(" w%d simd %d cu %d se %d |", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15, (h[b * 8 + w] >> 13) & 7);
            printf("\n");
        }
    }
    printf("workgroups with two waves on every SIMD: %d of %d; strictly round-robin from the first wave's SIMD: %d\n", balanced, blocks, rr);
    return 0;
}
