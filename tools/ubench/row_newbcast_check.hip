#include <hip/hip_runtime.h>
template <int J> __device__ double bc(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + J, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + J, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(const double* in, double* out) {
    double v = in[threadIdx.x];
    out[threadIdx.x] = bc<0>(v) + 2.0 * bc<5>(v) + 3.0 * bc<15>(v);
}
int main() {
    double h[64], *d_in, *d_out, r[64];
    for (int i = 0; i < 64; ++i) h[i] = i + 0.5;
    hipMalloc(&d_in, 512); hipMalloc(&d_out, 512);
    hipMemcpy(d_in, h, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_in, d_out);
    hipMemcpy(r, d_out, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { int b = (i / 16) * 16; double e = h[b] + 2.0 * h[b + 5] + 3.0 * h[b + 15]; if (r[i] != e) ++bad; }
    printf("row_newbcast mismatches: %d (lane 17 got %.1f)\n", bad, r[17]);
    return bad;
}
