// How accurate is v_rcp_f64 on gfx950 before any refinement?  (The error norm's quotient refines it with one Newton step;
// whether that step is needed for a VALUE that only scales the next trial step depends on this.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* in, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_rcp(in[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> h(n), r(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> U(-12.0, 8.0), M(1.0, 2.0);
    for (int i = 0; i < n; ++i) h[i] = (i & 1) ? std::pow(10.0, U(g)) : M(g);
    double *d_in, *d_out;
    hipMalloc(&d_in, n * 8); hipMalloc(&d_out, n * 8);
    hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(d_in, d_out, n);
    hipMemcpy(r.data(), d_out, n * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; ++i) {
        const long double exact = 1.0L / (long double)h[i];
        const double rel = (double)fabsl(((long double)r[i] - exact) / exact);
        if (rel > worst) worst = rel;
    }
    printf("v_rcp_f64: max relative error over %d arguments %.3e = 2^%.1f\n", n, worst, std::log2(worst));
    return 0;
}
