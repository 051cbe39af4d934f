// HBM streaming probes for MI355X: what do a read-only, a write-only and a read-modify-write (copy in place) kernel reach on
// buffers far larger than the caches?  Context for the roofline fractions in DESIGN.md (k_fit is a 12 B read + 12 B write per
// atom stream).   hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip && ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const v4f *in, size_t n4, float *sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    v4f acc = { 0, 0, 0, 0 };
    for (; i < n4; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(in + i);
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = 1.f;
}
template <int NT>
__global__ __launch_bounds__(256) void k_rmw(v4f *buf, size_t n4) {   // one 16-byte element per lane, in place
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    v4f v = NT ? __builtin_nontemporal_load(buf + i) : buf[i];
    v = v * 1.0001f + 0.5f;
    if (NT) __builtin_nontemporal_store(v, buf + i); else buf[i] = v;
}
template <int NT>
__global__ __launch_bounds__(256) void k_rmw_tile(v4f *buf, size_t n4) {   // k_fit's shape: a wave moves 3 KiB (3 x 16 B per lane)
    const size_t t = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 192 + (threadIdx.x & 63);
    if (t + 128 >= n4) return;
    v4f a = __builtin_nontemporal_load(buf + t), b = __builtin_nontemporal_load(buf + t + 64), c = __builtin_nontemporal_load(buf + t + 128);
    a = a * 1.0001f + 0.5f; b = b * 1.0001f + 0.5f; c = c * 1.0001f + 0.5f;
    __builtin_nontemporal_store(a, buf + t); __builtin_nontemporal_store(b, buf + t + 64); __builtin_nontemporal_store(c, buf + t + 128);
}
__global__ __launch_bounds__(256) void k_write(v4f *out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { v4f v = { (float)blockIdx.x, 1.f, 2.f, 3.f }; __builtin_nontemporal_store(v, out + i); }
}
template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t bytes = 3072ull << 20, n4 = bytes / 16;   // 3 GiB: 256 frames of 12 MB
    v4f *buf; float *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    auto rep = [&](const char *name, float ms, double traffic) { printf("%-46s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, traffic / (ms * 1e-3) / 1e9); };
    rep("read only (grid-stride, 16 B per lane)", timeit([&] { k_read<<<dim3(256 * 16), dim3(256)>>>(buf, n4, sink); }, 5), (double)bytes);
    rep("write only", timeit([&] { k_write<<<dim3((n4 + 255) / 256), dim3(256)>>>(buf, n4); }, 5), (double)bytes);
    rep("read-modify-write in place, plain", timeit([&] { k_rmw<0><<<dim3((n4 + 255) / 256), dim3(256)>>>(buf, n4); }, 5), 2.0 * bytes);
    rep("read-modify-write in place, non-temporal", timeit([&] { k_rmw<1><<<dim3((n4 + 255) / 256), dim3(256)>>>(buf, n4); }, 5), 2.0 * bytes);
    rep("read-modify-write, 3 KiB tile per wave (k_fit)", timeit([&] { k_rmw_tile<1><<<dim3((n4 / 192 + 3) / 4), dim3(256)>>>(buf, n4); }, 5), 2.0 * bytes);
    return 0;
}
