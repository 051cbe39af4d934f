// Pure-write bandwidth probe for MI355X: what can a store-only kernel reach?  (ceiling for k_pairdist, whose algorithmic
// traffic is the 4 B/pair matrix write).   hipcc --offload-arch=gfx950 -O3 -o write_bw write_bw.hip && ./write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT, int PER>
__global__ __launch_bounds__(256) void k_write(v4f *out, size_t n4) {
    // each workgroup writes PER consecutive 4 KiB rows (256 lanes x 16 B)
    size_t base = ((size_t)blockIdx.x * PER) * 256 + threadIdx.x;
    v4f v = { (float)blockIdx.x, 1.f, 2.f, 3.f };
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        size_t i = base + (size_t)r * 256;
        if (i < n4) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
    }
}
template <int NT, int PER>
__global__ __launch_bounds__(256) void k_write_strided(v4f *out, size_t n4, size_t row4) {
    // pairdist-like: workgroup (bx, by) writes PER rows of a row-major matrix, 4 KiB of each row
    v4f v = { (float)blockIdx.x, 1.f, 2.f, 3.f };
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        size_t col = (size_t)blockIdx.x * 256 + threadIdx.x;
        size_t i = ((size_t)blockIdx.y * PER + r) * row4 + col;
        if (col < row4 && i < n4) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t bytes = 400ull << 20, n4 = bytes / 16;
    v4f *out; CK(hipMalloc(&out, bytes));
    auto rep = [&](const char *name, float ms) { printf("%-40s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9); };
#define LIN(NT, PER) rep("linear NT=" #NT " PER=" #PER, timeit([&] { k_write<NT, PER><<<dim3((n4 + 256 * PER - 1) / (256 * PER)), dim3(256)>>>(out, n4); }, 20))
    LIN(0, 1); LIN(1, 1); LIN(0, 4); LIN(1, 4); LIN(0, 16); LIN(1, 16); LIN(1, 64);
    const size_t row4 = 10240 / 4;   // 10240 floats per row
    const size_t rows = n4 / row4;
#define STR(NT, PER) rep("matrix NT=" #NT " PER=" #PER, timeit([&] { k_write_strided<NT, PER><<<dim3((row4 + 255) / 256, rows / PER), dim3(256)>>>(out, n4, row4); }, 20))
    STR(0, 1); STR(1, 1); STR(0, 4); STR(1, 4); STR(0, 16); STR(1, 16); STR(1, 32);
    float ms = timeit([&] { (void)hipMemsetAsync(out, 0, bytes, 0); }, 10); rep("hipMemsetAsync", ms);
    // sustained: 16 different 400 MiB buffers in turn (nothing can linger in the 256 MB Infinity Cache)
    v4f *big; CK(hipMalloc(&big, 16 * bytes));
    ms = timeit([&] { for (int f = 0; f < 16; ++f) k_write_strided<1, 8><<<dim3((row4 + 255) / 256, rows / 8), dim3(256)>>>(big + (size_t)f * n4, n4, row4); }, 3) / 16;
    rep("matrix NT=1 PER=8, 16 buffers in turn", ms);
    ms = timeit([&] { for (int f = 0; f < 16; ++f) k_write<1, 1><<<dim3((n4 + 255) / 256), dim3(256)>>>(big + (size_t)f * n4, n4); }, 3) / 16;
    rep("linear NT=1 PER=1, 16 buffers in turn", ms);
    ms = timeit([&] { for (int f = 0; f < 16; ++f) k_write<0, 1><<<dim3((n4 + 255) / 256), dim3(256)>>>(big + (size_t)f * n4, n4); }, 3) / 16;
    rep("linear NT=0 PER=1, 16 buffers in turn", ms);
    ms = timeit([&] { (void)hipMemsetAsync(big, 0, 16 * bytes, 0); }, 3) / 16; rep("hipMemsetAsync 6.4 GiB (per 400 MiB)", ms);
    return 0;
}
