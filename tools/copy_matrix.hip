// copy_matrix.hip -- where does the distance between the guide's 6.29 TB/s float4 copy (MI355X_MICROARCH.md:36) and the 5.5-5.8 TB/s
// that every probe of the resident pass's own address stream reaches (profiles/r04_ceiling.json) come from?
// start from a PLAIN LINEAR float4 copy and change one thing at a time until the pass's shape is reached.
//
// Every row of the output is one configuration: best and median of 5 timed launches (one warm-up), bytes = read + written.
//   family "lin"     k_lin<U, LD, ST, PIPE>: a grid of W workgroups x 256 lanes; per turn a workgroup moves U x 4 KiB contiguous
//                    (lane l, piece u: float4 index base + u * 256 + l); U loads are in flight before the first store.
//                    walk = "stride": the grid sweeps the buffer together (turn t of workgroup b = chunk t * W + b)
//                    walk = "block":  workgroup b owns one contiguous 1/W of the buffer
//                    PIPE: the loads of the next turn are issued before the stores of this one (2 x U registers)
//                    LD: 0 plain, 1 nt.  ST: 0 plain, 1 nt, 2 sc1 nt.
//                    place: "oop" dst = a second allocation, "oop-adj" dst = second half of the same allocation, "inplace" dst = src
//   family "one"     one float4 per lane, one turn, grid = n4 / 256 (hipMemcpy's shape; no loop)
//   family "tile"    the pass's row tiling: a wave moves 3 KiB = rows b, b + 64, b + 128 of a 192-float4 tile (U = 3 by construction)
//   family "frame"   grid (x, frames): 12 MB frames, workgroup (x, f) grid-strides over frame f            (ceiling_bench's shape)
//   family "persist" W workgroups x 512 lanes x 2 groups, every lane keeps its offset and walks the frames, loads D frames ahead,
//                    stores LAG frames behind (LAG = 0: in place right away; LAG = 6: the pass)             (ceiling_resident's shape)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/copy_matrix tools/copy_matrix.hip
// Run:   tools/bin/copy_matrix [GiB per buffer = 16] > gpurun_out/copy_matrix.jsonl
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));

template <int LD> __device__ __forceinline__ v4f ld(const v4f *p) {
    if (LD == 1) return __builtin_nontemporal_load(p);
    return *p;
}
template <int ST> __device__ __forceinline__ void st(v4f *p, v4f v) {
    if (ST == 1) __builtin_nontemporal_store(v, p);
    else if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    else *p = v;
}
__device__ __forceinline__ v4f touch(v4f v) { return v * 1.0000001f + 1e-9f; }

// ---------------------------------------------------------------------------------------------------------------- lin
template <int U, int LD, int ST, bool PIPE>
__global__ __launch_bounds__(256) void k_lin(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n_chunks, int block_walk) {
    // chunk = U * 256 float4.  stride walk: chunks b, b + W, ...; block walk: chunks [b * per, (b + 1) * per)
    const size_t W = gridDim.x, b = blockIdx.x;
    const size_t per = (n_chunks + W - 1) / W;
    size_t c = block_walk ? b * per : b, end = block_walk ? std::min(n_chunks, (b + 1) * per) : n_chunks, step = block_walk ? 1 : W;
    if (c >= end) return;
    v4f r[U], q[U];
    if (PIPE) {
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = ld<LD>(src + c * (U * 256) + u * 256 + threadIdx.x);
    }
    for (; c < end; c += step) {
        const size_t base = c * (U * 256) + threadIdx.x;
        if (PIPE) {
            const size_t cn = c + step;
            if (cn < end) {
#pragma unroll
                for (int u = 0; u < U; ++u) q[u] = ld<LD>(src + cn * (U * 256) + u * 256 + threadIdx.x);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) st<ST>(dst + base + u * 256, touch(r[u]));
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = q[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = ld<LD>(src + base + u * 256);
#pragma unroll
            for (int u = 0; u < U; ++u) st<ST>(dst + base + u * 256, touch(r[u]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- one
template <int LD, int ST>
__global__ __launch_bounds__(256) void k_one(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) st<ST>(dst + i, touch(ld<LD>(src + i)));
}

// ---------------------------------------------------------------------------------------------------------------- tile
// a wave owns tiles of 192 float4 (3 KiB): rows b, b + 64, b + 128.  Workgroup = 4 waves = 4 consecutive tiles per turn (12 KiB).
template <int LD, int ST>
__global__ __launch_bounds__(256) void k_tile(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n_tiles4, int block_walk) {
    const size_t W = gridDim.x, b = blockIdx.x;
    const size_t per = (n_tiles4 + W - 1) / W;
    size_t c = block_walk ? b * per : b, end = block_walk ? std::min(n_tiles4, (b + 1) * per) : n_tiles4, step = block_walk ? 1 : W;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (; c < end; c += step) {
        const size_t t = (c * 4 + wave) * 192 + lane;
        v4f a = ld<LD>(src + t), bb = ld<LD>(src + t + 64), cc = ld<LD>(src + t + 128);
        st<ST>(dst + t, touch(a)); st<ST>(dst + t + 64, touch(bb)); st<ST>(dst + t + 128, touch(cc));
    }
}

// ---------------------------------------------------------------------------------------------------------------- frame
// grid (x, frames): frame f = 12 MB at src + f * stride4; workgroup (x, f) grid-strides over the frame's tiles
template <int LD, int ST>
__global__ __launch_bounds__(256) void k_frame(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t stride4, uint32_t ngroups) {
    const v4f *s = src + (size_t)blockIdx.y * stride4;
    v4f *d = dst + (size_t)blockIdx.y * stride4;
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += gridDim.x * 256) {
        const size_t t = (size_t)(g >> 6) * 192 + (g & 63);
        v4f a = ld<LD>(s + t), bb = ld<LD>(s + t + 64), cc = ld<LD>(s + t + 128);
        st<ST>(d + t, touch(a)); st<ST>(d + t + 64, touch(bb)); st<ST>(d + t + 128, touch(cc));
    }
}

// ---------------------------------------------------------------------------------------------------------------- persist
// W workgroups x 512 lanes, G groups per lane; buffer loads / stores (no branches: a frame that does not exist is a resource of
// zero records), D frames of loads in flight, stores LAG frames behind the loads' frame.  Frame order lockstep.
#define RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr), 0, (int)(bytes), 0x00020000)
template <int G, int D, int LDAUX, int STAUX>
__global__ __launch_bounds__(512) void k_persist(const float *src, float *dst, size_t stride_f, uint32_t nframes, uint32_t ngroups, int lag) {
    const uint32_t base = blockIdx.x * 512 * G;
    uint32_t off[G];
#pragma unroll
    for (int q = 0; q < G; ++q) { const uint32_t g = base + q * 512 + threadIdx.x; off[q] = g < ngroups ? ((g >> 6) * 192 + (g & 63)) * 16u : 0xFFFFF000u; }
    const uint32_t slot_bytes = (uint32_t)(stride_f * 4);
    i4v buf[D][G][3];
    auto request = [&](uint32_t f, i4v (&r)[G][3]) {
        const __amdgpu_buffer_rsrc_t rs = RSRC(src + (size_t)(f < nframes ? f : 0) * stride_f, f < nframes ? slot_bytes : 0u);
#pragma unroll
        for (int q = 0; q < G; ++q) {
            r[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[q], 0, LDAUX);
            r[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[q] + 1024, 0, LDAUX);
            r[q][2] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[q] + 2048, 0, LDAUX);
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) request(d, buf[d]);
    for (uint32_t f0 = 0; f0 < nframes; f0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint32_t f = f0 + d;
            i4v cur[G][3];
#pragma unroll
            for (int q = 0; q < G; ++q) { cur[q][0] = buf[d][q][0]; cur[q][1] = buf[d][q][1]; cur[q][2] = buf[d][q][2]; }
            request(f + D, buf[d]);
            const uint32_t ft = f >= (uint32_t)lag ? f - lag : f + nframes - lag;
            const __amdgpu_buffer_rsrc_t rd = RSRC(dst + (size_t)(f < nframes ? ft : 0) * stride_f, f < nframes ? slot_bytes : 0u);
#pragma unroll
            for (int q = 0; q < G; ++q) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { i4v v = cur[q][k]; v.x += 1; v.y ^= 3; v.z += 5; v.w ^= 7; __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)off[q] + 1024 * k, 0, STAUX); }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- host
static hipEvent_t e0, e1;
template <class F> static void run(const char *family, const char *desc, double bytes, F f) {
    std::vector<float> t;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        f();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) t.push_back(ms);
    }
    CHECK(hipGetLastError());
    std::sort(t.begin(), t.end());
    printf("{\"family\": \"%s\", %s, \"GB\": %.3f, \"best_ms\": %.4f, \"median_ms\": %.4f, \"best_TBs\": %.3f, \"median_TBs\": %.3f}\n", family, desc, bytes / 1e9, t[0], t[t.size() / 2],
           bytes / (t[0] * 1e-3) / 1e12, bytes / (t[t.size() / 2] * 1e-3) / 1e12);
    fflush(stdout);
}

template <int U, int LD, int ST, bool PIPE>
static void lin(const v4f *src, v4f *dst, size_t n4, int W, int block_walk, const char *place, double gib) {
    const size_t n_chunks = n4 / (U * 256);
    char d[256];
    snprintf(d, sizeof d, "\"U\": %d, \"ld\": \"%s\", \"st\": \"%s\", \"pipe\": %d, \"wgs\": %d, \"walk\": \"%s\", \"place\": \"%s\", \"window_GiB\": %.1f", U, LD ? "nt" : "plain",
             ST == 0 ? "plain" : ST == 1 ? "nt" : "sc1 nt", (int)PIPE, W, block_walk ? "block" : "stride", place, gib);
    run("lin", d, 2.0 * n_chunks * U * 256 * 16, [&] { k_lin<U, LD, ST, PIPE><<<dim3(W), dim3(256)>>>(src, dst, n_chunks, block_walk); });
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const size_t bytes = (size_t)(gib * 1024.0) << 20, n4 = bytes / 16;
    v4f *A, *B;
    CHECK(hipMalloc(&A, bytes)); CHECK(hipMalloc(&B, bytes));
    CHECK(hipMemset(A, 0, bytes)); CHECK(hipMemset(B, 0, bytes));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"family\": \"meta\", \"device\": \"%s\", \"cus\": %d, \"mem_clock_khz\": %d, \"bus_bits\": %d, \"buffer_GiB\": %.1f}\n", prop.name, prop.multiProcessorCount, prop.memoryClockRate, prop.memoryBusWidth, gib);

    // hipMemcpyDtoD as the vendor's own answer
    run("memcpy", "\"what\": \"hipMemcpyDtoDAsync\"", 2.0 * bytes, [&] { CHECK(hipMemcpyAsync(B, A, bytes, hipMemcpyDeviceToDevice, 0)); });

    // ---- step 0: one float4 per lane, no loop
    run("one", "\"ld\": \"plain\", \"st\": \"plain\"", 2.0 * n4 * 16, [&] { k_one<0, 0><<<dim3((unsigned)(n4 / 256)), dim3(256)>>>(A, B, n4); });
    run("one", "\"ld\": \"nt\", \"st\": \"nt\"", 2.0 * n4 * 16, [&] { k_one<1, 1><<<dim3((unsigned)(n4 / 256)), dim3(256)>>>(A, B, n4); });
    run("one", "\"ld\": \"nt\", \"st\": \"sc1 nt\"", 2.0 * n4 * 16, [&] { k_one<1, 2><<<dim3((unsigned)(n4 / 256)), dim3(256)>>>(A, B, n4); });
    run("one", "\"ld\": \"nt\", \"st\": \"nt\", \"place\": \"inplace\"", 2.0 * n4 * 16, [&] { k_one<1, 1><<<dim3((unsigned)(n4 / 256)), dim3(256)>>>(A, A, n4); });

    // ---- step 1: linear grid-stride loop; loads in flight x workgroups x walk, nt both ways, out of place
    const int Ws[] = { 256, 512, 1024, 2048, 4096, 8192 };
    for (int W : Ws)
        for (int bw = 0; bw < 2; ++bw) {
            lin<1, 1, 1, false>(A, B, n4, W, bw, "oop", gib);
            lin<2, 1, 1, false>(A, B, n4, W, bw, "oop", gib);
            lin<4, 1, 1, false>(A, B, n4, W, bw, "oop", gib);
            lin<8, 1, 1, false>(A, B, n4, W, bw, "oop", gib);
            lin<16, 1, 1, false>(A, B, n4, W, bw, "oop", gib);
        }
    // ---- step 2: next turn's loads before this turn's stores
    for (int W : { 1024, 2048, 4096 })
        for (int bw = 0; bw < 2; ++bw) {
            lin<1, 1, 1, true>(A, B, n4, W, bw, "oop", gib);
            lin<3, 1, 1, true>(A, B, n4, W, bw, "oop", gib);
            lin<4, 1, 1, true>(A, B, n4, W, bw, "oop", gib);
            lin<8, 1, 1, true>(A, B, n4, W, bw, "oop", gib);
        }
    // ---- step 3: cache policies at U = 4, 2048 workgroups
    for (int bw = 0; bw < 2; ++bw) {
        lin<4, 0, 0, false>(A, B, n4, 2048, bw, "oop", gib);
        lin<4, 1, 0, false>(A, B, n4, 2048, bw, "oop", gib);
        lin<4, 0, 1, false>(A, B, n4, 2048, bw, "oop", gib);
        lin<4, 1, 2, false>(A, B, n4, 2048, bw, "oop", gib);
        lin<4, 0, 2, false>(A, B, n4, 2048, bw, "oop", gib);
    }
    // ---- step 4: placement: same allocation (adjacent halves), in place
    for (int bw = 0; bw < 2; ++bw) {
        lin<4, 1, 1, false>(A, A + n4 / 2, n4 / 2, 2048, bw, "oop-adj", gib / 2);
        lin<4, 1, 1, false>(A, A, n4, 2048, bw, "inplace", gib);
        lin<3, 1, 1, false>(A, A, n4, 2048, bw, "inplace", gib);
        lin<3, 1, 2, false>(A, A, n4, 2048, bw, "inplace", gib);
        lin<3, 1, 1, true>(A, A, n4, 2048, bw, "inplace", gib);
    }
    // ---- step 5: window size (the grid-launched probes of r04 walked 3 GiB): 1.5 / 3 / 6 GiB of the same buffers
    for (double g : { 0.75, 1.5, 3.0, 6.0 }) {
        if (g > gib) break;
        const size_t m4 = ((size_t)(g * 1024.0) << 20) / 16;
        lin<4, 1, 1, false>(A, B, m4, 2048, 0, "oop", g);
        lin<4, 1, 1, false>(A, A, m4, 2048, 0, "inplace", g);
    }
    // ---- step 6: the pass's row tiling (wave = 3 KiB tile), linear buffer
    for (int W : { 976, 2048, 4096, 8192 })
        for (int bw = 0; bw < 2; ++bw) {
            char d[256];
            const size_t nt4 = n4 / (192 * 4);
            snprintf(d, sizeof d, "\"ld\": \"nt\", \"st\": \"nt\", \"wgs\": %d, \"walk\": \"%s\", \"place\": \"oop\", \"window_GiB\": %.1f", W, bw ? "block" : "stride", gib);
            run("tile", d, 2.0 * nt4 * 4 * 192 * 16, [&] { k_tile<1, 1><<<dim3(W), dim3(256)>>>(A, B, nt4, bw); });
            snprintf(d, sizeof d, "\"ld\": \"nt\", \"st\": \"nt\", \"wgs\": %d, \"walk\": \"%s\", \"place\": \"inplace\", \"window_GiB\": %.1f", W, bw ? "block" : "stride", gib);
            run("tile", d, 2.0 * nt4 * 4 * 192 * 16, [&] { k_tile<1, 1><<<dim3(W), dim3(256)>>>(A, A, nt4, bw); });
        }
    // ---- step 7: frames of 1e6 atoms (12 MB), grid (x, frame)
    const uint32_t n_atoms = 1000000u, ntiles = (n_atoms + 255) / 256, ngroups = ntiles * 64;
    const size_t stride4 = (size_t)ntiles * 192, stride_f = stride4 * 4;
    const uint32_t fmax = (uint32_t)(n4 / stride4);
    for (uint32_t F : { 256u, 768u, fmax })
        for (int gx : { 128, 488, 976 }) {
            if (F > fmax) continue;
            char d[256];
            snprintf(d, sizeof d, "\"ld\": \"nt\", \"st\": \"nt\", \"grid_x\": %d, \"frames\": %u, \"place\": \"inplace\", \"window_GiB\": %.2f", gx, F, F * stride4 * 16 / 1073741824.0);
            run("frame", d, 2.0 * F * (double)stride4 * 16, [&] { k_frame<1, 1><<<dim3(gx, F), dim3(256)>>>(A, A, stride4, ngroups); });
            snprintf(d, sizeof d, "\"ld\": \"nt\", \"st\": \"nt\", \"grid_x\": %d, \"frames\": %u, \"place\": \"oop\", \"window_GiB\": %.2f", gx, F, F * stride4 * 16 / 1073741824.0);
            run("frame", d, 2.0 * F * (double)stride4 * 16, [&] { k_frame<1, 1><<<dim3(gx, F), dim3(256)>>>(A, B, stride4, ngroups); });
        }
    // ---- step 8: persistent walk of the same frames (245 workgroups x 512 lanes x 2 groups); prefetch depth, lag, store policy
    const int Wp = (int)((ngroups + 1023) / 1024);
    for (uint32_t F : { 768u, fmax }) {
        if (F > fmax) continue;
        auto row = [&](const char *what, int D, int lag, const char *place, auto kern, const float *s, float *t) {
            char d[256];
            snprintf(d, sizeof d, "\"what\": \"%s\", \"wgs\": %d, \"D\": %d, \"lag\": %d, \"frames\": %u, \"place\": \"%s\", \"window_GiB\": %.2f", what, Wp, D, lag, F, place, F * stride4 * 16 / 1073741824.0);
            run("persist", d, 2.0 * F * (double)stride4 * 16, [&] { kern<<<dim3(Wp), dim3(512)>>>(s, t, stride_f, F, ngroups, lag); });
        };
        const float *a = (const float *)A; float *am = (float *)A, *bm = (float *)B;
        row("ld nt, st nt", 1, 0, "inplace", k_persist<2, 1, 2, 2>, a, am);
        row("ld nt, st nt", 2, 0, "inplace", k_persist<2, 2, 2, 2>, a, am);
        row("ld nt, st nt", 4, 0, "inplace", k_persist<2, 4, 2, 2>, a, am);
        row("ld nt, st sc1 nt", 1, 0, "inplace", k_persist<2, 1, 2, 18>, a, am);
        row("ld nt, st sc1 nt", 2, 0, "inplace", k_persist<2, 2, 2, 18>, a, am);
        row("ld nt, st sc1 nt", 1, 6, "inplace", k_persist<2, 1, 2, 18>, a, am);
        row("ld nt, st sc1 nt", 2, 6, "inplace", k_persist<2, 2, 2, 18>, a, am);
        row("ld nt, st sc1 nt", 2, 0, "oop", k_persist<2, 2, 2, 18>, a, bm);
        row("ld nt, st nt", 2, 0, "oop", k_persist<2, 2, 2, 2>, a, bm);
        row("ld plain, st plain", 2, 0, "oop", k_persist<2, 2, 0, 0>, a, bm);
    }
    return 0;
}
