// vsin_accuracy.hip -- how accurate are the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) on gfx950, against fp64 libm?
// Decides whether k_center_sums<1> (Bai-Breen: 3 sin + 3 cos per atom, iterators.rs:1152-1191, auxiliary.rs:59-99) may use them.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/vsin_accuracy tools/microbench/vsin_accuracy.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *u, float *s, float *c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = __builtin_amdgcn_sinf(u[i]);
    c[i] = __builtin_amdgcn_cosf(u[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> u(n), s(n), c(n);
    for (int i = 0; i < n; ++i) u[i] = (float)((double)i / n);           // [0, 1) revolutions, every 2.4e-7
    float *du, *ds, *dc;
    hipMalloc(&du, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(du, u.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(du, ds, dc, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, ms = 0, mc = 0, bs = 0, bc = 0;
    for (int i = 0; i < n; ++i) {
        const double a = 6.283185307179586 * (double)u[i];
        const double e1 = (double)s[i] - sin(a), e2 = (double)c[i] - cos(a);
        es += e1 * e1; ec += e2 * e2; bs += e1; bc += e2;
        if (fabs(e1) > ms) ms = fabs(e1);
        if (fabs(e2) > mc) mc = fabs(e2);
    }
    printf("{\"n\": %d, \"sin_max_abs_err\": %.3e, \"cos_max_abs_err\": %.3e, \"sin_rms_err\": %.3e, \"cos_rms_err\": %.3e, \"sin_mean_err\": %.3e, \"cos_mean_err\": %.3e}\n",
           n, ms, mc, sqrt(es / n), sqrt(ec / n), bs / n, bc / n);
    return 0;
}
