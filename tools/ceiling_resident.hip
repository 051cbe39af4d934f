// ceiling_resident.hip -- what bounds a PERSISTENT copy of the resident pass's shape?  (k_fit_resident sits at
// 0.97 of an arithmetic-free "resident copy" that is itself 9 % slower than a grid-launched copy of the same bytes.)
// One launch, W workgroups, every lane walks all frames of the launch with its own 4-atom groups: per iteration i it requests the
// rows of frame i + 1, waits for the rows of frame i and stores them into slot i - LAG (LAG = 6 mimics the pass: loads run one
// frame ahead of the sums stage, stores six frames behind it; the CONTENT is irrelevant for a memory floor, so nothing is parked).
// Variants (one JSON row each, us per 1e6 atoms and frame):
//   base        245 x 512 lanes x 2 groups, lockstep, LAG 6                (the pass's own shape)
//   lag0        ... stores into the frame just read (in place, no lag)
//   oop         ... stores into a second buffer (out of place)
//   wg256       256 x 512 x 2 on a 1 048 576-atom frame                   (every CU streams)
//   stagger-x   workgroup b walks the frames rotated by (b % 8) * nframes / 8      (8 different frames in flight: one per XCD)
//   stagger-w   ... rotated by b * nframes / W                                     (every workgroup in a different frame)
//   2wg         490 x 512 lanes x 1 group                                   (two workgroups per CU: phases may drift)
//   1024x1      245 x 1024 lanes x 1 group                                  (round 2's shape)
//   rd / wr     the loads alone / the stores alone at the pass's shape
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ceiling_resident tools/ceiling_resident.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float4 ldnt(const float4 *p) { float4 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w); return v; }
__device__ __forceinline__ void stnt(float4 *p, float4 v) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w); }
__device__ __forceinline__ void touch(float4 &r) { r.x = fmaf(r.x, 1.0000001f, 1e-9f); r.y = fmaf(r.y, 1.0000001f, 1e-9f); r.z = fmaf(r.z, 1.0000001f, 1e-9f); r.w = fmaf(r.w, 1.0000001f, 1e-9f); }

struct Rows { float4 a[3]; };

// MODE 0 copy, 1 loads only, 2 stores only
template <int LANES, int GROUPS, int MODE>
__global__ __launch_bounds__(LANES) void k_rcopy(const float *src, float *dst, size_t stride, uint32_t nframes, uint32_t ngroups, int lag, uint32_t rot_mod, uint32_t rot_div, float *sink) {
    const uint32_t base = blockIdx.x * LANES * GROUPS;
    uint32_t g[GROUPS]; size_t b[GROUPS]; bool ok[GROUPS];
#pragma unroll
    for (int q = 0; q < GROUPS; ++q) { g[q] = base + q * LANES + threadIdx.x; ok[q] = g[q] < ngroups; const uint32_t gg = ok[q] ? g[q] : 0u; b[q] = (size_t)(gg >> 6) * 192 + (gg & 63); }
    // frame order of this workgroup: rotated by rot (0 = lockstep)
    const uint32_t rot = rot_div ? (uint32_t)(((unsigned long long)(blockIdx.x % rot_mod) * nframes) / rot_div) : 0u;
    auto fr = [&](uint32_t i) { uint32_t f = i + rot; return f >= nframes ? f - nframes : f; };
    Rows cur[GROUPS], nxt[GROUPS];
    float acc = 0.f;
    auto request = [&](uint32_t i, Rows *r) {
        const float4 *f4 = reinterpret_cast<const float4 *>(src + (size_t)fr(i) * stride);
#pragma unroll
        for (int q = 0; q < GROUPS; ++q) if (ok[q]) { r[q].a[0] = ldnt(f4 + b[q]); r[q].a[1] = ldnt(f4 + b[q] + 64); r[q].a[2] = ldnt(f4 + b[q] + 128); }
    };
#pragma unroll
    for (int q = 0; q < GROUPS; ++q) { cur[q].a[0] = cur[q].a[1] = cur[q].a[2] = make_float4(1.f, 2.f, 3.f, 4.f); nxt[q] = cur[q]; }
    if (MODE != 2) request(0, cur);
    for (uint32_t i = 0; i < nframes; ++i) {
        if (MODE != 2 && i + 1 < nframes) request(i + 1, nxt);
        if (MODE != 1) {
            const uint32_t it = i >= (uint32_t)lag ? i - lag : i + nframes - lag;   // the slot written this iteration (wraps: every slot once)
            float4 *d4 = reinterpret_cast<float4 *>(dst + (size_t)fr(it) * stride);
#pragma unroll
            for (int q = 0; q < GROUPS; ++q) if (ok[q]) {
                touch(cur[q].a[0]); touch(cur[q].a[1]); touch(cur[q].a[2]);
                stnt(d4 + b[q], cur[q].a[0]); stnt(d4 + b[q] + 64, cur[q].a[1]); stnt(d4 + b[q] + 128, cur[q].a[2]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < GROUPS; ++q) acc += cur[q].a[0].x + cur[q].a[1].y + cur[q].a[2].z;
        }
#pragma unroll
        for (int q = 0; q < GROUPS; ++q) cur[q] = nxt[q];
    }
    if (MODE == 1 && acc == 12345.678f) sink[0] = acc;
}

// ---- the same walk with the waits the hardware actually needs.  The kernel above leaves the waits to the compiler, and the compiler --
// faced with `if (ok)` / `if (i + 1 < nframes)` around the loads -- drains the wave's memory queue (s_waitcnt vmcnt(0)) before the
// stores of every iteration: the rows requested for the NEXT frame are waited for at once, nothing is prefetched, and the "ceiling" is
// that of a loop that exposes a full memory latency per frame and wave (round 4 found the same drain in k_fit_resident itself).
// Here: buffer loads / stores whose out-of-range lanes and frames are suppressed by the buffer's own bounds check (num_records = 0
// for a frame that does not exist, offset 0xFFFFFFF0 for a group that does not exist) -- no branch, no exec mask, no phi copies --
// two named register sets, and the loop unrolled by two, so that the only wait before the stores of frame i is "all but the six
// loads of frame i + 1".
typedef int i4v __attribute__((ext_vector_type(4)));
#define GR_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr), 0, (int)(bytes), 0x00020000)
__device__ __forceinline__ i4v touch_i(i4v v) { v.x += 1; v.y ^= 3; v.z += 5; v.w ^= 7; return v; }
template <int LANES, int MODE, int STORE_AUX = 2>
__global__ __launch_bounds__(LANES) void k_rcopy_pipelined(float *frames, float *dst_frames, size_t stride, uint32_t nframes, uint32_t ngroups, int lag) {
    const uint32_t base = blockIdx.x * LANES * 2;
    uint32_t off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { const uint32_t g = base + q * LANES + threadIdx.x; off[q] = g < ngroups ? ((g >> 6) * 192 + (g & 63)) * 16u : 0xFFFFF000u; }
    const uint32_t fbytes = (uint32_t)(stride * 4);
    i4v A[2][3], B[2][3];
    auto load = [&](uint32_t f, i4v (&R)[2][3]) {
        const bool live = f < nframes;
        __amdgpu_buffer_rsrc_t s = GR_RSRC(frames + (size_t)(live ? f : 0u) * stride, live ? fbytes : 0u);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 3; ++r) R[q][r] = __builtin_amdgcn_raw_buffer_load_b128(s, off[q] + 1024 * r, 0, 2);
    };
    auto store = [&](uint32_t i, i4v (&R)[2][3]) {
        const uint32_t it = i >= (uint32_t)lag ? i - lag : i + nframes - lag;
        __amdgpu_buffer_rsrc_t d = GR_RSRC(dst_frames + (size_t)it * stride, fbytes);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 3; ++r) { R[q][r] = touch_i(R[q][r]); __builtin_amdgcn_raw_buffer_store_b128(R[q][r], d, off[q] + 1024 * r, 0, STORE_AUX); }
    };
    int acc = 0;
    auto fold = [&](i4v (&R)[2][3]) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 3; ++r) acc += R[q][r].x ^ R[q][r].w;
    };
    if (MODE != 2) load(0, A);
    else {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 3; ++r) A[q][r] = B[q][r] = i4v{ 1, 2, 3, (int)threadIdx.x };
    }
    for (uint32_t i = 0; i < nframes; i += 2) {       // nframes even
        if (MODE != 2) load(i + 1, B);
        if (MODE != 1) store(i, A); else fold(A);
        if (MODE != 2) load(i + 2, A);
        if (MODE != 1) store(i + 1, B); else fold(B);
    }
    if (MODE == 1 && acc == 0x12345678) dst_frames[0] = (float)acc;
}

// ---- deeper prefetch: D frames ahead with D + 1 named register sets, the loop unrolled by D + 1 (compile-time rotation, no copies).
// How much of the persistent shape's deficit against a grid-launched copy is bytes in flight per CU?  (8 waves x 6 KB one frame
// ahead = 48 KB per CU; a grid launch keeps 32 waves x 3 KB = 96 KB in flight.)
template <int D>
__global__ __launch_bounds__(512) void k_rcopy_deep(float *frames, size_t stride, uint32_t nframes, uint32_t ngroups, int lag) {
    const uint32_t base = blockIdx.x * 1024;
    uint32_t off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { const uint32_t g = base + q * 512 + threadIdx.x; off[q] = g < ngroups ? ((g >> 6) * 192 + (g & 63)) * 16u : 0xFFFFF000u; }
    const uint32_t fbytes = (uint32_t)(stride * 4);
    i4v R[D + 1][2][3];
    auto load = [&](uint32_t f, i4v (&X)[2][3]) {
        const bool live = f < nframes;
        __amdgpu_buffer_rsrc_t s = GR_RSRC(frames + (size_t)(live ? f : 0u) * stride, live ? fbytes : 0u);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 3; ++r) X[q][r] = __builtin_amdgcn_raw_buffer_load_b128(s, off[q] + 1024 * r, 0, 2);
    };
    auto store = [&](uint32_t i, i4v (&X)[2][3]) {
        if (i >= nframes) return;
        const uint32_t it = i >= (uint32_t)lag ? i - lag : i + nframes - lag;
        __amdgpu_buffer_rsrc_t d = GR_RSRC(frames + (size_t)it * stride, fbytes);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 3; ++r) { X[q][r] = touch_i(X[q][r]); __builtin_amdgcn_raw_buffer_store_b128(X[q][r], d, off[q] + 1024 * r, 0, 2); }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, R[d]);
    for (uint32_t i = 0; i < nframes; i += D + 1) {
#pragma unroll
        for (int u = 0; u <= D; ++u) {
            load(i + u + D, R[(u + D) % (D + 1)]);
            store(i + u, R[u]);
        }
    }
}

// --quick [atoms] : the two rows bench.py wants from the SAME box and process tree as its own line -- the persistent copy of the resident
// pass's shape (waits written out, one frame of prefetch, stores six frames behind) with the stores the pass uses (sc1 nt) and with plain
// nt stores; as many frames per launch as the bench's own launches walk (768 by default: the floor rises with the footprint), best of 5 launches
static int quick(uint32_t n, uint32_t frames) {
    const uint32_t ntiles = (n + 255) / 256, ngroups = ntiles * 64;
    const size_t stride = (size_t)ntiles * 768;
    float *F;
    CHECK(hipMalloc(&F, stride * frames * sizeof(float)));
    CHECK(hipMemset(F, 0, stride * frames * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const uint32_t wgs = (ngroups + 1023) / 1024;
    double us[2] = { 0, 0 };
    {   // a device that has been idle runs its first tenths of a second slow (clocks): warm up for ~0.6 s before anything is timed
        float total = 0.f;
        while (total < 600.f) {
            CHECK(hipEventRecord(e0));
            k_rcopy_pipelined<512, 0, 18><<<wgs, 512>>>(F, F, stride, frames, ngroups, 6);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            total += ms > 0.f ? ms : 1.f;
        }
    }
    for (int v = 0; v < 2; ++v) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipEventRecord(e0));
            if (v == 0) k_rcopy_pipelined<512, 0, 18><<<wgs, 512>>>(F, F, stride, frames, ngroups, 6);
            else k_rcopy_pipelined<512, 0, 2><<<wgs, 512>>>(F, F, stride, frames, ngroups, 6);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        us[v] = 1e3 * best / frames;
    }
    printf("{\"n_atoms\": %u, \"frames_per_launch\": %u, \"workgroups\": %u, \"persistent_copy_us_per_frame_stores_sc1_nt\": %.4f, \"persistent_copy_us_per_frame_stores_nt\": %.4f}\n", n, frames, wgs, us[0], us[1]);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "--quick")) return quick(argc > 2 ? (uint32_t)atoi(argv[2]) : 1000000u, argc > 3 ? (uint32_t)atoi(argv[3]) & ~1u : 768u);
    const uint32_t frames = argc > 1 ? (uint32_t)atoi(argv[1]) : 256u;
    const uint32_t n_big = 1048576u;
    const uint32_t ntiles_big = n_big / 256;
    const size_t stride = (size_t)ntiles_big * 768;
    float *F, *F2, *sink;
    CHECK(hipMalloc(&F, stride * frames * sizeof(float)));
    CHECK(hipMalloc(&F2, stride * frames * sizeof(float)));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(F, 0, stride * frames * sizeof(float)));
    CHECK(hipMemset(F2, 0, stride * frames * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct Case { const char *name; int kind; uint32_t n; uint32_t wgs; int lag; uint32_t rot_mod, rot_div; bool oop; };
    const uint32_t n1 = 1000000u;
    const Case cases[] = {
        { "base: 245 x 512 x 2, lockstep, lag 6", 0, n1, 245, 6, 1, 0, false },
        { "lag0: in place, no lag", 0, n1, 245, 0, 1, 0, false },
        { "oop: stores into a second buffer, lag 6", 0, n1, 245, 6, 1, 0, true },
        { "wg256: 256 x 512 x 2 on 1 048 576 atoms", 0, n_big, 256, 6, 1, 0, false },
        { "wg240: 240 x 512 x 2 on 983 040 atoms", 0, 983040u, 240, 6, 1, 0, false },
        { "stagger-x: 8 frames in flight (by XCD)", 0, n1, 245, 6, 8, 8, false },
        { "stagger-w: every workgroup in its own frame", 0, n1, 245, 6, 245, 245, false },
        { "stagger-2: two halves of the chip half a launch apart", 0, n1, 245, 6, 2, 2, false },
        { "2wg: 490 x 512 x 1", 1, n1, 490, 6, 1, 0, false },
        { "1024x1: 245 x 1024 x 1", 2, n1, 245, 6, 1, 0, false },
        { "4wg: 980 x 256 x 1", 5, n1, 980, 6, 1, 0, false },
        { "rd: loads alone", 3, n1, 245, 6, 1, 0, false },
        { "wr: stores alone", 4, n1, 245, 6, 1, 0, false },
    };
    printf("{\"frames_per_launch\": %u, \"results\": [\n", frames);
    bool first = true;
    for (const Case &c : cases) {
        const uint32_t ngroups = ((c.n + 255) / 256) * 64;
        float best = 1e30f, sum = 0.f; int cnt = 0;
        for (int rep = 0; rep < 7; ++rep) {
            CHECK(hipEventRecord(e0));
            float *dst = c.oop ? F2 : F;
            switch (c.kind) {
            case 0: k_rcopy<512, 2, 0><<<c.wgs, 512>>>(F, dst, stride, frames, ngroups, c.lag, c.rot_mod, c.rot_div, sink); break;
            case 1: k_rcopy<512, 1, 0><<<c.wgs, 512>>>(F, dst, stride, frames, ngroups, c.lag, c.rot_mod, c.rot_div, sink); break;
            case 2: k_rcopy<1024, 1, 0><<<c.wgs, 1024>>>(F, dst, stride, frames, ngroups, c.lag, c.rot_mod, c.rot_div, sink); break;
            case 3: k_rcopy<512, 2, 1><<<c.wgs, 512>>>(F, dst, stride, frames, ngroups, c.lag, c.rot_mod, c.rot_div, sink); break;
            case 4: k_rcopy<512, 2, 2><<<c.wgs, 512>>>(F, dst, stride, frames, ngroups, c.lag, c.rot_mod, c.rot_div, sink); break;
            default: k_rcopy<256, 1, 0><<<c.wgs, 256>>>(F, dst, stride, frames, ngroups, c.lag, c.rot_mod, c.rot_div, sink); break;
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) { if (ms < best) best = ms; sum += ms; ++cnt; }
        }
        const double us = 1e3 * best / frames, us_mean = 1e3 * (sum / cnt) / frames, per1e6 = us * 1e6 / c.n;
        const double bytes = (c.kind == 3 || c.kind == 4) ? 12.0 : 24.0;
        printf("%s {\"kernel\": \"%s\", \"n_atoms\": %u, \"workgroups\": %u, \"us_per_frame\": %.3f, \"us_per_frame_mean\": %.3f, \"us_per_1e6_atoms\": %.3f, \"hbm_GBs\": %.0f}",
               first ? " " : ",\n ", c.name, c.n, c.wgs, us, us_mean, per1e6, bytes * c.n / (us * 1e-6) / 1e9);
        first = false;
    }
    {   // the properly pipelined walk: copy (in place, lag 6 / lag 0 / out of place), loads alone, stores alone; 512 and 1024 lanes
        const uint32_t ngroups = ((n1 + 255) / 256) * 64;
        struct P { const char *name; int lanes, mode, lag; bool oop; double bytes; };
        const P ps[] = { { "pipelined copy: 245 x 512 x 2, lag 6", 512, 0, 6, false, 24.0 }, { "pipelined copy, lag 0", 512, 0, 0, false, 24.0 },
                         { "pipelined copy, out of place, lag 6", 512, 0, 6, true, 24.0 }, { "pipelined copy: 123 x 1024 x 2, lag 6", 1024, 0, 6, false, 24.0 },
                         { "pipelined loads alone", 512, 1, 6, false, 12.0 }, { "pipelined stores alone", 512, 2, 6, false, 12.0 } };
        for (const P &c : ps) {
            float best = 1e30f, sum = 0.f; int cnt = 0;
            const uint32_t wgs = (ngroups + c.lanes * 2 - 1) / (c.lanes * 2);
            for (int rep = 0; rep < 7; ++rep) {
                CHECK(hipEventRecord(e0));
                float *dst = c.oop ? F2 : F;
                if (c.lanes == 512) {
                    if (c.mode == 0) k_rcopy_pipelined<512, 0><<<wgs, 512>>>(F, dst, stride, frames, ngroups, c.lag);
                    else if (c.mode == 1) k_rcopy_pipelined<512, 1><<<wgs, 512>>>(F, dst, stride, frames, ngroups, c.lag);
                    else k_rcopy_pipelined<512, 2><<<wgs, 512>>>(F, dst, stride, frames, ngroups, c.lag);
                } else k_rcopy_pipelined<1024, 0><<<wgs, 1024>>>(F, dst, stride, frames, ngroups, c.lag);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0) { if (ms < best) best = ms; sum += ms; ++cnt; }
            }
            const double us = 1e3 * best / frames, us_mean = 1e3 * (sum / cnt) / frames;
            printf(",\n  {\"kernel\": \"%s\", \"n_atoms\": %u, \"workgroups\": %u, \"us_per_frame\": %.3f, \"us_per_frame_mean\": %.3f, \"us_per_1e6_atoms\": %.3f, \"hbm_GBs\": %.0f}",
                   c.name, n1, wgs, us, us_mean, us, c.bytes * n1 / (us * 1e-6) / 1e9);
        }
    }
    for (int D = 1; D <= 4; ++D) {
        const uint32_t ngroups = ((n1 + 255) / 256) * 64;
        float best = 1e30f;
        const uint32_t fr = frames - frames % (D + 1);
        for (int rep = 0; rep < 7; ++rep) {
            CHECK(hipEventRecord(e0));
            if (D == 1) k_rcopy_deep<1><<<245, 512>>>(F, stride, fr, ngroups, 6);
            else if (D == 2) k_rcopy_deep<2><<<245, 512>>>(F, stride, fr, ngroups, 6);
            else if (D == 3) k_rcopy_deep<3><<<245, 512>>>(F, stride, fr, ngroups, 6);
            else k_rcopy_deep<4><<<245, 512>>>(F, stride, fr, ngroups, 6);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double us = 1e3 * best / fr;
        printf(",\n  {\"kernel\": \"pipelined copy, %d frames ahead (245 x 512 x 2, lag 6)\", \"n_atoms\": %u, \"workgroups\": 245, \"us_per_frame\": %.3f, \"us_per_1e6_atoms\": %.3f, \"hbm_GBs\": %.0f}",
               D, n1, us, us, 24.0 * n1 / (us * 1e-6) / 1e9);
    }
    printf("\n]}\n");
    return 0;
}
