// layout_bench.hip -- which HBM layout lets a streaming read-modify-write of float3 positions run at
// memory speed on MI355X?  Same arithmetic (shift, wrap into a triclinic cell, rotate, translate) on
//   (1) packed xyz records, each lane loads/stores 3 float4 at a 48-byte stride      ("aos_strided")
//   (2) packed xyz records, lanes load/store contiguous float4 and transpose via LDS ("aos_lds")
//   (3) three planes x[], y[], z[], each lane loads/stores one float4 per plane      ("soa")
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/layout_bench tools/layout_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct P { float sx, sy, sz, ax, by, cz, bx, cx, cy, bcx, bcy, bcz, r[9], tx, ty, tz; };

__device__ __forceinline__ void tf(float &x, float &y, float &z, const P &p) {
    x += p.sx; y += p.sy; z += p.sz;
    float k = floorf(z / p.cz); x -= k * p.cx; y -= k * p.cy; z -= k * p.cz;
    k = floorf(y / p.by); x -= k * p.bx; y -= k * p.by;
    k = floorf(x / p.ax); x -= k * p.ax;
    x -= p.bcx; y -= p.bcy; z -= p.bcz;
    const float nx = p.r[0] * x + p.r[3] * y + p.r[6] * z, ny = p.r[1] * x + p.r[4] * y + p.r[7] * z, nz = p.r[2] * x + p.r[5] * y + p.r[8] * z;
    x = nx + p.tx; y = ny + p.ty; z = nz + p.tz;
}

__global__ __launch_bounds__(256) void aos_strided(float *base, size_t stride, uint32_t n, P p) {
    float4 *f4 = reinterpret_cast<float4 *>(base + blockIdx.y * stride);
    const uint32_t ng = n >> 2;
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        float4 a = f4[3 * (size_t)g], b = f4[3 * (size_t)g + 1], c = f4[3 * (size_t)g + 2];
        tf(a.x, a.y, a.z, p); tf(a.w, b.x, b.y, p); tf(b.z, b.w, c.x, p); tf(c.y, c.z, c.w, p);
        f4[3 * (size_t)g] = a; f4[3 * (size_t)g + 1] = b; f4[3 * (size_t)g + 2] = c;
    }
}

__global__ __launch_bounds__(256) void aos_lds(float *base, size_t stride, uint32_t n, P p) {
    __shared__ float4 tile[4][192];   // one 3 KiB tile per wave
    float4 *f4 = reinterpret_cast<float4 *>(base + blockIdx.y * stride);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ntiles = (n >> 2) / 64;   // tiles of 256 atoms = 192 float4 (n multiple of 256 here)
    for (uint32_t t = (blockIdx.x * 4 + wave); t < ntiles; t += gridDim.x * 4) {
        float4 *src = f4 + (size_t)t * 192;
        tile[wave][lane] = src[lane]; tile[wave][lane + 64] = src[lane + 64]; tile[wave][lane + 128] = src[lane + 128];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 a = tile[wave][3 * lane], b = tile[wave][3 * lane + 1], c = tile[wave][3 * lane + 2];
        tf(a.x, a.y, a.z, p); tf(a.w, b.x, b.y, p); tf(b.z, b.w, c.x, p); tf(c.y, c.z, c.w, p);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        tile[wave][3 * lane] = a; tile[wave][3 * lane + 1] = b; tile[wave][3 * lane + 2] = c;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        src[lane] = tile[wave][lane]; src[lane + 64] = tile[wave][lane + 64]; src[lane + 128] = tile[wave][lane + 128];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void soa(float *base, size_t stride, uint32_t n, P p) {
    float *fr = base + blockIdx.y * stride;
    float4 *X = reinterpret_cast<float4 *>(fr), *Y = reinterpret_cast<float4 *>(fr + n), *Z = reinterpret_cast<float4 *>(fr + 2 * (size_t)n);
    const uint32_t ng = n >> 2;
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        float4 x = X[g], y = Y[g], z = Z[g];
        tf(x.x, y.x, z.x, p); tf(x.y, y.y, z.y, p); tf(x.z, y.z, z.z, p); tf(x.w, y.w, z.w, p);
        X[g] = x; Y[g] = y; Z[g] = z;
    }
}

__global__ void fill(float *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (float)((i * 2654435761u) % 24000) * 1e-3f; }

int main() {
    const uint32_t n = 1000192 / 256 * 256;   // multiple of 256
    const int frames = 64, reps = 5;
    const size_t stride = (size_t)n * 3;
    float *buf;
    CHECK(hipMalloc(&buf, stride * frames * sizeof(float)));
    fill<<<(unsigned)((stride * frames + 255) / 256), 256>>>(buf, stride * frames);
    P p = { 1.f, 2.f, 3.f, 24.18f, 24.18f, 17.1f, -1e-6f, 12.09f, 12.09f, 18.f, 18.f, 8.5f, { .36f, .48f, -.8f, -.8f, .6f, 0.f, .48f, .64f, .6f }, 12.f, 12.f, 8.f };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[3] = { "aos_strided", "aos_lds", "soa" };
    for (int gx : { 256, 512, 1024 }) {
        for (int k = 0; k < 3; ++k) {
            float best = 1e30f;
            for (int r = 0; r < reps; ++r) {
                CHECK(hipEventRecord(e0));
                if (k == 0) aos_strided<<<dim3(gx, frames), 256>>>(buf, stride, n, p);
                if (k == 1) aos_lds<<<dim3(gx, frames), 256>>>(buf, stride, n, p);
                if (k == 2) soa<<<dim3(gx, frames), 256>>>(buf, stride, n, p);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("grid.x=%4d %-12s %8.3f ms  %7.2f us/frame  %7.1f GB/s (24 B/atom)\n", gx, names[k], best, 1e3 * best / frames, 24.0 * n * frames / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
