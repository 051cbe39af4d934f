// mall_bench.hip -- does a second pass over X MB hit the 256 MiB Infinity Cache (MALL) on MI355X, and how fast?
// pass 1: read X MB (sum); pass 2: read the same X MB again; pass 3: read-modify-write the same X MB in place.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mall_bench tools/mall_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void rd(const float4 *p, size_t n4, float *out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += gridDim.x * 256ull) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void rmw(float4 *p, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += gridDim.x * 256ull) { float4 v = p[i]; v.x += 1.f; v.y *= 1.0001f; v.z -= 1.f; v.w += 2.f; p[i] = v; }
}
int main() {
    float *buf, *out, *trash;
    const size_t maxb = 1024ull << 20;
    CHECK(hipMalloc(&buf, maxb)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&trash, maxb));
    CHECK(hipMemset(buf, 0, maxb)); CHECK(hipMemset(trash, 0, maxb));
    hipEvent_t e0, e1, e2, e3; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2)); CHECK(hipEventCreate(&e3));
    for (size_t mb : { 24, 48, 96, 144, 192, 256, 384, 768 }) {
        const size_t n4 = (mb << 20) / 16;
        float b1 = 1e9f, b2 = 1e9f, b3 = 1e9f;
        for (int r = 0; r < 5; ++r) {
            rd<<<2048, 256>>>((const float4 *)trash, maxb / 16, out);   // evict
            CHECK(hipEventRecord(e0)); rd<<<2048, 256>>>((const float4 *)buf, n4, out);
            CHECK(hipEventRecord(e1)); rd<<<2048, 256>>>((const float4 *)buf, n4, out);
            CHECK(hipEventRecord(e2)); rmw<<<2048, 256>>>((float4 *)buf, n4);
            CHECK(hipEventRecord(e3)); CHECK(hipEventSynchronize(e3));
            float t1, t2, t3; CHECK(hipEventElapsedTime(&t1, e0, e1)); CHECK(hipEventElapsedTime(&t2, e1, e2)); CHECK(hipEventElapsedTime(&t3, e2, e3));
            if (t1 < b1) b1 = t1; if (t2 < b2) b2 = t2; if (t3 < b3) b3 = t3;
        }
        const double gb = (double)(mb << 20) / 1e9;
        printf("%4zu MB: cold read %7.1f GB/s | re-read %7.1f GB/s | rmw after read %7.1f GB/s (r+w bytes)\n", mb, gb / (b1 * 1e-3), gb / (b2 * 1e-3), 2 * gb / (b3 * 1e-3));
    }
    return 0;
}
