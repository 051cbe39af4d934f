// ceiling_bench.hip -- what the memory system gives the two passes of the RMSD fit at the bench's own launch shape, with
// the arithmetic taken out: 256 frames of 1e6 atoms (12 MB each, pair-tiled rows of float4) per launch, grid (x, frame).
//   read        every row of the frame loaded once (non-temporal), folded into one number per lane      12 MB / frame from HBM
//   read+pw     the same + the reference rows (12 MB) and the weights (4 MB) that every frame shares   + 16 MB / frame from L2 / MALL
//   copy        rows loaded, one fma, stored in place (non-temporal both ways)                           24 MB / frame
//   copy+pw     the same + reference rows and weights
// These are the floors of k_sums_pk (read+pw) and k_fit_pk<true> (copy+pw); DESIGN.md quotes the kernels against them.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ceiling_bench tools/ceiling_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float4 ldnt(const float4 *p) { float4 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w); return v; }
__device__ __forceinline__ void stnt(float4 *p, float4 v) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w); }
__device__ __forceinline__ float fold(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// tile t of 256 atoms = rows [3 t, 3 t + 3) of 64 float4; lane L of group g = 64 t + L reads row r at (g / 64) * 192 + r * 64 + L
// out-of-place copy: frame f of the first half of the buffer -> frame f of the second half (does an in-place read-modify-write
// stream cost anything against separate source and destination?)
__global__ __launch_bounds__(256) void k_oop(const float *src, float *dst, size_t stride, uint32_t ngroups) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src + blockIdx.y * stride);
    float4 *d4 = reinterpret_cast<float4 *>(dst + blockIdx.y * stride);
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += gridDim.x * 256) {
        const size_t b = (size_t)(g >> 6) * 192 + (g & 63);
        float4 r0 = ldnt(s4 + b), r1 = ldnt(s4 + b + 64), r2 = ldnt(s4 + b + 128);
        float4 *r[3] = { &r0, &r1, &r2 };
#pragma unroll
        for (int q = 0; q < 3; ++q) { r[q]->x = fmaf(r[q]->x, 1.0000001f, 1e-9f); r[q]->y = fmaf(r[q]->y, 1.0000001f, 1e-9f); r[q]->z = fmaf(r[q]->z, 1.0000001f, 1e-9f); r[q]->w = fmaf(r[q]->w, 1.0000001f, 1e-9f); }
        stnt(d4 + b, r0); stnt(d4 + b + 64, r1); stnt(d4 + b + 128, r2);
    }
}

template <bool PW, bool COPY>
__global__ __launch_bounds__(256) void k(float *frames, size_t stride, const float4 *__restrict__ p4, const float4 *__restrict__ w4, uint32_t ngroups, float *out) {
    float4 *f4 = reinterpret_cast<float4 *>(frames + blockIdx.y * stride);
    float acc = 0.f;
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += gridDim.x * 256) {
        const size_t b = (size_t)(g >> 6) * 192 + (g & 63);
        float4 r0 = ldnt(f4 + b), r1 = ldnt(f4 + b + 64), r2 = ldnt(f4 + b + 128);
        if (PW) {
            const float4 a0 = p4[b], a1 = p4[b + 64], a2 = p4[b + 128], w = w4[g];
            r0.x = fmaf(a0.x, w.x, r0.x); r0.y = fmaf(a0.y, w.y, r0.y); r0.z = fmaf(a0.z, w.z, r0.z); r0.w = fmaf(a0.w, w.w, r0.w);
            r1.x = fmaf(a1.x, w.x, r1.x); r1.y = fmaf(a1.y, w.y, r1.y); r1.z = fmaf(a1.z, w.z, r1.z); r1.w = fmaf(a1.w, w.w, r1.w);
            r2.x = fmaf(a2.x, w.x, r2.x); r2.y = fmaf(a2.y, w.y, r2.y); r2.z = fmaf(a2.z, w.z, r2.z); r2.w = fmaf(a2.w, w.w, r2.w);
        }
        if (COPY) {
            // (every component changes: a store of an unchanged loaded value would be dropped by the compiler)
            float4 *r[3] = { &r0, &r1, &r2 };
#pragma unroll
            for (int q = 0; q < 3; ++q) { r[q]->x = fmaf(r[q]->x, 1.0000001f, 1e-9f); r[q]->y = fmaf(r[q]->y, 1.0000001f, 1e-9f); r[q]->z = fmaf(r[q]->z, 1.0000001f, 1e-9f); r[q]->w = fmaf(r[q]->w, 1.0000001f, 1e-9f); }
            stnt(f4 + b, r0); stnt(f4 + b + 64, r1); stnt(f4 + b + 128, r2);
        } else acc += fold(r0) + fold(r1) + fold(r2);
    }
    if (!COPY && acc == 12345.678f) out[0] = acc;
}

// "resident" copy: ONE launch, 256 workgroups x 1024 lanes = one 4-atom group per lane; every lane walks the frames of the
// launch with its own group (loads D frames ahead, in registers), so the frame is read once and written once however many
// passes look at it in between -- the shape of a fused sums + fit pass with the frame parked on chip (24 MB / frame).
template <int D>
__global__ __launch_bounds__(1024) void k_resident(float *frames, size_t stride, uint32_t nframes, uint32_t ngroups) {
    const uint32_t g = blockIdx.x * 1024 + threadIdx.x;
    if (g >= ngroups) return;
    const size_t b = (size_t)(g >> 6) * 192 + (g & 63);
    float4 buf[D][3];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if ((uint32_t)d < nframes) { const float4 *f4 = reinterpret_cast<const float4 *>(frames + (size_t)d * stride); buf[d][0] = ldnt(f4 + b); buf[d][1] = ldnt(f4 + b + 64); buf[d][2] = ldnt(f4 + b + 128); }
    for (uint32_t f0 = 0; f0 < nframes; f0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint32_t f = f0 + d;
            if (f >= nframes) break;
            float4 r0 = buf[d][0], r1 = buf[d][1], r2 = buf[d][2];
            if (f + D < nframes) { const float4 *n4 = reinterpret_cast<const float4 *>(frames + (size_t)(f + D) * stride); buf[d][0] = ldnt(n4 + b); buf[d][1] = ldnt(n4 + b + 64); buf[d][2] = ldnt(n4 + b + 128); }
            float4 *r[3] = { &r0, &r1, &r2 };
#pragma unroll
            for (int q = 0; q < 3; ++q) { r[q]->x = fmaf(r[q]->x, 1.0000001f, 1e-9f); r[q]->y = fmaf(r[q]->y, 1.0000001f, 1e-9f); r[q]->z = fmaf(r[q]->z, 1.0000001f, 1e-9f); r[q]->w = fmaf(r[q]->w, 1.0000001f, 1e-9f); }
            float4 *f4 = reinterpret_cast<float4 *>(frames + (size_t)f * stride);
            stnt(f4 + b, r0); stnt(f4 + b + 64, r1); stnt(f4 + b + 128, r2);
        }
    }
}

// How long does one word take from a workgroup on one XCD to a workgroup on another while the rest of the chip streams at
// full rate?  blocks [0, n_copy) run the resident copy; the last two blocks bounce a counter: A writes k (agent scope), B
// answers k in a second word, A waits for it, k + 1 ... -> round trips per launch = 2 hops each.
template <int D>
__global__ __launch_bounds__(1024) void k_pingpong(float *frames, size_t stride, uint32_t nframes, uint32_t ngroups, uint32_t n_copy,
                                                   unsigned long long *flags, uint32_t *trips_out) {
    if (blockIdx.x < n_copy) {
        const uint32_t g = blockIdx.x * 1024 + threadIdx.x;
        const size_t b = (size_t)(g >> 6) * 192 + (g & 63);
        for (uint32_t f = 0; f < nframes && g < ngroups; ++f) {
            float4 *f4 = reinterpret_cast<float4 *>(frames + (size_t)f * stride);
            float4 r0 = ldnt(f4 + b), r1 = ldnt(f4 + b + 64), r2 = ldnt(f4 + b + 128);
            float4 *r[3] = { &r0, &r1, &r2 };
#pragma unroll
            for (int q = 0; q < 3; ++q) { r[q]->x = fmaf(r[q]->x, 1.0000001f, 1e-9f); r[q]->y = fmaf(r[q]->y, 1.0000001f, 1e-9f); r[q]->z = fmaf(r[q]->z, 1.0000001f, 1e-9f); r[q]->w = fmaf(r[q]->w, 1.0000001f, 1e-9f); }
            stnt(f4 + b, r0); stnt(f4 + b + 64, r1); stnt(f4 + b + 128, r2);
        }
        if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + 32, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // copy blocks done
        return;
    }
    if (threadIdx.x >= 64) return;
    const bool a_side = blockIdx.x == n_copy;
    unsigned long long k = 1;
    for (; k < 20000000ull;) {      // (bounded whatever happens to the other side)
        if (a_side) {
            __hip_atomic_store(flags + 0, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t spin = 0; spin < 100000000u && __hip_atomic_load(flags + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != k; ++spin)
                if (__hip_atomic_load(flags + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_copy) break;
        } else {
            for (uint32_t spin = 0; spin < 100000000u && __hip_atomic_load(flags + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != k; ++spin)
                if (__hip_atomic_load(flags + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_copy) break;
            __hip_atomic_store(flags + 16, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__hip_atomic_load(flags + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_copy) break;
        ++k;
    }
    if (a_side && threadIdx.x == 0) trips_out[0] = (uint32_t)k;
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u, frames = argc > 2 ? (uint32_t)atoi(argv[2]) : 256u;
    const uint32_t ntiles = (n + 255) / 256, ngroups = ntiles * 64;
    const size_t stride = (size_t)ntiles * 768;              // floats per frame slot
    float *F, *out; float4 *p4, *w4;
    CHECK(hipMalloc(&F, stride * frames * sizeof(float)));
    CHECK(hipMalloc(&p4, stride * sizeof(float)));
    CHECK(hipMalloc(&w4, (size_t)ngroups * sizeof(float4)));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(F, 0, stride * frames * sizeof(float)));
    CHECK(hipMemset(p4, 0, stride * sizeof(float)));
    CHECK(hipMemset(w4, 0, (size_t)ngroups * sizeof(float4)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[4] = { "read", "read+pw", "copy", "copy+pw" };
    const double mb[4] = { 12.0, 12.0, 24.0, 24.0 };
    const int grids[] = { 128, 256, 512, (int)((ngroups + 255) / 256) & ~7 };   // multiples of 8: block x stays on XCD x % 8 for every frame
    printf("{\"n_atoms\": %u, \"frames_per_launch\": %u, \"results\": [\n", n, frames);
    bool first = true;
    for (int gi = 0; gi < 4; ++gi)
        for (int v = 0; v < 4; ++v) {
            const dim3 grid(grids[gi], frames);
            float best = 1e30f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipEventRecord(e0));
                switch (v) {
                case 0: k<false, false><<<grid, 256>>>(F, stride, p4, w4, ngroups, out); break;
                case 1: k<true, false><<<grid, 256>>>(F, stride, p4, w4, ngroups, out); break;
                case 2: k<false, true><<<grid, 256>>>(F, stride, p4, w4, ngroups, out); break;
                default: k<true, true><<<grid, 256>>>(F, stride, p4, w4, ngroups, out); break;
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double us = 1e3 * best / frames;
            printf("%s {\"kernel\": \"%s\", \"grid_x\": %d, \"us_per_frame\": %.3f, \"hbm_GBs\": %.0f}", first ? " " : ",\n ", names[v], grids[gi], us, mb[v] * 1e6 * (n / 1e6) / (us * 1e-6) / 1e9);
            first = false;
        }
    {   // out of place: 128 frames -> the other 128
        float best = 1e30f;
        const uint32_t half = frames / 2;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipEventRecord(e0));
            k_oop<<<dim3(976, half), 256>>>(F, F + (size_t)half * stride, stride, ngroups);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double us = 1e3 * best / half;
        printf(",\n  {\"kernel\": \"copy, out of place\", \"grid_x\": 976, \"us_per_frame\": %.3f, \"hbm_GBs\": %.0f}", us, 24.0 * n / (us * 1e-6) / 1e9);
    }
    for (int D = 2; D <= 6; D += 2) {
        float best = 1e30f;
        const dim3 grid((ngroups + 1023) / 1024);
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipEventRecord(e0));
            if (D == 2) k_resident<2><<<grid, 1024>>>(F, stride, frames, ngroups);
            else if (D == 4) k_resident<4><<<grid, 1024>>>(F, stride, frames, ngroups);
            else k_resident<6><<<grid, 1024>>>(F, stride, frames, ngroups);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double us = 1e3 * best / frames;
        printf(",\n  {\"kernel\": \"resident copy, %d frames ahead\", \"grid_x\": %d, \"us_per_frame\": %.3f, \"hbm_GBs\": %.0f}", D, (int)grid.x, us, 24.0 * n / (us * 1e-6) / 1e9);
    }
    {
        unsigned long long *flags; uint32_t *trips;
        CHECK(hipMalloc(&flags, 64 * 8)); CHECK(hipMalloc(&trips, 4));
        const uint32_t n_copy = (ngroups + 1023) / 1024;
        for (int loaded = 1; loaded >= 0; --loaded) {
            CHECK(hipMemset(flags, 0, 64 * 8));
            CHECK(hipEventRecord(e0));
            k_pingpong<1><<<dim3(n_copy + 2), 1024>>>(F, stride, frames, loaded ? ngroups : 64u, n_copy, flags, trips);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint32_t t = 0; CHECK(hipMemcpy(&t, trips, 4, hipMemcpyDeviceToHost));
            printf(",\n  {\"kernel\": \"word round trip between two workgroups, chip %s\", \"launch_ms\": %.3f, \"round_trips\": %u, \"us_per_round_trip\": %.3f}",
                   loaded ? "streaming (resident copy)" : "idle (one wave copies)", ms, t, t ? 1e3 * ms / t : 0.0);
        }
    }
    printf("\n]}\n");
    return 0;
}
