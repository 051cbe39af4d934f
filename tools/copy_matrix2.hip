// copy_matrix2.hip -- second sheet of the copy matrix (tools/copy_matrix.hip found: a copy WITHOUT a loop -- one float4 per lane, 4 M
// workgroups handed out by the dispatcher -- runs at 6.5-6.7 TB/s over 16 + 16 GiB, every LOOP form at 4.9-5.3 whatever its depth).
// What is it about the loop?  One thing at a time:
//   family "oneU"   no loop, U float4 per lane (grid = n4 / (LANES * U)): does the 6.5 survive more bytes per wave?  (+ the pass's tile
//                   addressing at U = 3; + 512 / 1024-lane workgroups)
//   family "dyn"    W resident workgroups, every WAVE takes its next chunk (U KiB) from one atomic counter, the next index requested
//                   before the current chunk is moved: a loop with the dispatcher's load balance
//   family "ts"     the static loop (lin, stride / block walk) and the persistent frame walk with every workgroup's start and end stamped
//                   (wall_clock64, 100 MHz): is the launch as long as its slowest workgroups (spread) or slow everywhere (no spread)?
//   family "occ"    the static loop at 1 / 2 / 4 / 8 workgroups per CU (dynamic LDS as ballast)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/copy_matrix2 tools/copy_matrix2.hip ;  run: tools/bin/copy_matrix2 [GiB = 16]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f ldnt(const v4f *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(v4f *p, v4f v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ v4f touch(v4f v) { return v * 1.0000001f + 1e-9f; }

template <int LANES, int U, bool TILE>
__global__ __launch_bounds__(LANES) void k_oneU(const v4f *__restrict__ src, v4f *__restrict__ dst) {
    size_t base;
    int pitch;
    if (TILE) { base = ((size_t)blockIdx.x * (LANES / 64) + (threadIdx.x >> 6)) * (64 * U) + (threadIdx.x & 63); pitch = 64; }   // a wave owns U KiB
    else { base = (size_t)blockIdx.x * (LANES * U) + threadIdx.x; pitch = LANES; }                                               // a workgroup owns U x LANES x 16 B
    v4f r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = ldnt(src + base + (size_t)u * pitch);
#pragma unroll
    for (int u = 0; u < U; ++u) stnt(dst + base + (size_t)u * pitch, touch(r[u]));
}

// every wave pulls chunks of U KiB (64 lanes x U float4, contiguous) from a counter
template <int U>
__global__ __launch_bounds__(256) void k_dyn(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n_chunks, unsigned long long *head) {
    const uint32_t lane = threadIdx.x & 63;
    auto pull = [&]() -> unsigned long long {
        unsigned long long c = 0;
        if (lane == 0) c = __hip_atomic_fetch_add(head, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return c;
    };
    unsigned long long nxt = pull();
    for (;;) {
        const unsigned long long c = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nxt)) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(nxt >> 32)) << 32);
        if (c >= n_chunks) break;
        nxt = pull();
        const size_t base = (size_t)c * (64 * U) + lane;
        v4f r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = ldnt(src + base + u * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) stnt(dst + base + u * 64, touch(r[u]));
    }
}

// static loop with stamps: stamps[2 b] = start, stamps[2 b + 1] = end of workgroup b
template <int U>
__global__ __launch_bounds__(256) void k_lin_ts(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n_chunks, int block_walk, unsigned long long *stamps) {
    extern __shared__ char ballast[];
    const size_t W = gridDim.x, b = blockIdx.x;
    const size_t per = (n_chunks + W - 1) / W;
    size_t c = block_walk ? b * per : b, end = block_walk ? std::min(n_chunks, (b + 1) * per) : n_chunks, step = block_walk ? 1 : W;
    if (threadIdx.x == 0 && stamps) stamps[2 * b] = wall_clock64();
    for (; c < end; c += step) {
        const size_t base = c * (U * 256) + threadIdx.x;
        v4f r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = ldnt(src + base + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) stnt(dst + base + u * 256, touch(r[u]));
    }
    __syncthreads();
    if (threadIdx.x == 0 && stamps) stamps[2 * b + 1] = wall_clock64();
}

// the persistent frame walk (copy_matrix.hip's k_persist at G = 2, D = 2, lag 0), stamped, + XCC id
typedef int i4v __attribute__((ext_vector_type(4)));
#define RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr), 0, (int)(bytes), 0x00020000)
template <int LANES, int G>
__global__ __launch_bounds__(LANES) void k_persist_ts(const float *src, float *dst, size_t stride_f, uint32_t nframes, uint32_t ngroups, unsigned long long *stamps) {
    extern __shared__ char ballast[];
    constexpr int D = 2;
    const uint32_t base = blockIdx.x * LANES * G;
    uint32_t off[G];
#pragma unroll
    for (int q = 0; q < G; ++q) { const uint32_t g = base + q * LANES + threadIdx.x; off[q] = g < ngroups ? ((g >> 6) * 192 + (g & 63)) * 16u : 0xFFFFF000u; }
    const uint32_t slot_bytes = (uint32_t)(stride_f * 4);
    if (threadIdx.x == 0 && stamps) { stamps[3 * blockIdx.x] = wall_clock64(); stamps[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) /* XCC_ID */; }
    i4v buf[D][G][3];
    auto request = [&](uint32_t f, i4v (&r)[G][3]) {
        const __amdgpu_buffer_rsrc_t rs = RSRC(src + (size_t)(f < nframes ? f : 0) * stride_f, f < nframes ? slot_bytes : 0u);
#pragma unroll
        for (int q = 0; q < G; ++q) {
            r[q][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[q], 0, 2);
            r[q][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[q] + 1024, 0, 2);
            r[q][2] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[q] + 2048, 0, 2);
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
            const __amdgpu_buffer_rsrc_t rd = RSRC(dst + (size_t)(f < nframes ? f : 0) * stride_f, f < nframes ? slot_bytes : 0u);
#pragma unroll
            for (int q = 0; q < G; ++q) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { i4v v = cur[q][k]; v.x += 1; v.y ^= 3; v.z += 5; v.w ^= 7; __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)off[q] + 1024 * k, 0, 18); }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && stamps) stamps[3 * blockIdx.x + 1] = wall_clock64();
}

static hipEvent_t e0, e1;
template <class F> static void run(const char *family, const char *desc, double bytes, F f, const char *extra = nullptr, std::function<std::string()> after = nullptr);
#include <functional>
#include <string>
template <class F> static void run(const char *family, const char *desc, double bytes, F f, const char *extra, std::function<std::string()> after) {
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
    std::string tail = after ? after() : std::string();
    printf("{\"family\": \"%s\", %s, \"GB\": %.3f, \"best_ms\": %.4f, \"median_ms\": %.4f, \"best_TBs\": %.3f, \"median_TBs\": %.3f%s}\n", family, desc, bytes / 1e9, t[0], t[t.size() / 2],
           bytes / (t[0] * 1e-3) / 1e12, bytes / (t[t.size() / 2] * 1e-3) / 1e12, tail.c_str());
    fflush(stdout);
}

// spread of the stamps of the LAST launch: start / end percentiles in us relative to the earliest start (100 MHz clock)
static std::string spread(unsigned long long *dev, int n_wg, int words, bool xcc) {
    std::vector<unsigned long long> h((size_t)n_wg * words);
    CHECK(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < n_wg; ++b) t0 = std::min(t0, h[(size_t)b * words]);
    std::vector<double> s, e;
    for (int b = 0; b < n_wg; ++b) { s.push_back((h[(size_t)b * words] - t0) * 0.01); e.push_back((h[(size_t)b * words + 1] - t0) * 0.01); }
    std::vector<double> es = e; std::sort(es.begin(), es.end()); std::sort(s.begin(), s.end());
    auto pc = [&](const std::vector<double> &v, double p) { return v[(size_t)(p * (v.size() - 1))]; };
    char buf[512];
    int k = snprintf(buf, sizeof buf, ", \"start_us_p50_max\": [%.1f, %.1f], \"end_us_min_p10_p50_p90_max\": [%.1f, %.1f, %.1f, %.1f, %.1f]", pc(s, 0.5), s.back(), es.front(), pc(es, 0.1), pc(es, 0.5), pc(es, 0.9), es.back());
    if (xcc) {
        double sum[8] = { 0 }; int cnt[8] = { 0 };
        for (int b = 0; b < n_wg; ++b) { const int x = (int)(h[(size_t)b * words + 2] & 7); sum[x] += e[b]; ++cnt[x]; }
        k += snprintf(buf + k, sizeof buf - k, ", \"mean_end_us_by_xcc\": [");
        for (int x = 0; x < 8; ++x) k += snprintf(buf + k, sizeof buf - k, "%s%.1f", x ? ", " : "", cnt[x] ? sum[x] / cnt[x] : 0.0);
        k += snprintf(buf + k, sizeof buf - k, "], \"wgs_by_xcc\": [");
        for (int x = 0; x < 8; ++x) k += snprintf(buf + k, sizeof buf - k, "%s%d", x ? ", " : "", cnt[x]);
        k += snprintf(buf + k, sizeof buf - k, "]");
    }
    return std::string(buf);
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 16.0;
    const size_t bytes = (size_t)(gib * 1024.0) << 20, n4 = bytes / 16;
    v4f *A, *B; unsigned long long *head, *stamps;
    CHECK(hipMalloc(&A, bytes)); CHECK(hipMalloc(&B, bytes)); CHECK(hipMalloc(&head, 64)); CHECK(hipMalloc(&stamps, 8 * 3 * 16384));
    CHECK(hipMemset(A, 0, bytes)); CHECK(hipMemset(B, 0, bytes));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    char d[256];

    // ---- oneU
#define ONEU(LANES, U, TILE) do { snprintf(d, sizeof d, "\"lanes\": %d, \"U\": %d, \"tile\": %d, \"place\": \"oop\", \"window_GiB\": %.1f", LANES, U, (int)TILE, gib); \
        const size_t nwg = n4 / ((size_t)LANES * U); run("oneU", d, 2.0 * nwg * LANES * U * 16, [&] { k_oneU<LANES, U, TILE><<<dim3((unsigned)nwg), dim3(LANES)>>>(A, B); }); } while (0)
    ONEU(256, 1, false); ONEU(256, 2, false); ONEU(256, 3, false); ONEU(256, 4, false); ONEU(256, 6, false); ONEU(256, 8, false); ONEU(256, 16, false);
    ONEU(256, 3, true); ONEU(256, 6, true);
    ONEU(512, 1, false); ONEU(512, 3, true); ONEU(512, 6, true); ONEU(1024, 1, false); ONEU(1024, 3, true); ONEU(64, 1, false); ONEU(64, 3, true); ONEU(128, 3, true);
    // in place
    { snprintf(d, sizeof d, "\"lanes\": 256, \"U\": 3, \"tile\": 1, \"place\": \"inplace\", \"window_GiB\": %.1f", gib);
      const size_t nwg = n4 / (256 * 3); run("oneU", d, 2.0 * nwg * 256 * 3 * 16, [&] { k_oneU<256, 3, true><<<dim3((unsigned)nwg), dim3(256)>>>(A, A); }); }

    // ---- dyn
#define DYN(U, W) do { snprintf(d, sizeof d, "\"U\": %d, \"wgs\": %d, \"place\": \"oop\", \"window_GiB\": %.1f", U, W, gib); const size_t nc = n4 / (64 * U); \
        run("dyn", d, 2.0 * nc * 64 * U * 16, [&] { CHECK(hipMemsetAsync(head, 0, 8, 0)); k_dyn<U><<<dim3(W), dim3(256)>>>(A, B, nc, head); }); } while (0)
    DYN(1, 2048); DYN(2, 2048); DYN(3, 2048); DYN(4, 2048); DYN(6, 2048); DYN(8, 2048);
    DYN(3, 256); DYN(3, 512); DYN(3, 1024); DYN(6, 512); DYN(6, 1024);

    // ---- ts: static loops stamped
    for (int W : { 2048, 1024, 256 })
        for (int bw = 0; bw < 2; ++bw) {
            snprintf(d, sizeof d, "\"what\": \"lin U=4\", \"wgs\": %d, \"walk\": \"%s\", \"window_GiB\": %.1f", W, bw ? "block" : "stride", gib);
            const size_t nc = n4 / (4 * 256);
            run("ts", d, 2.0 * nc * 4 * 256 * 16, [&] { k_lin_ts<4><<<dim3(W), dim3(256)>>>(A, B, nc, bw, stamps); }, nullptr, [&] { return spread(stamps, W, 2, false); });
        }
    // ---- occ: the static loop at fewer workgroups per CU (LDS ballast: 160 KiB / k)
    for (int per_cu : { 1, 2, 4, 8 }) {
        const int W = 256 * per_cu, lds = per_cu == 8 ? 0 : (160 * 1024 / per_cu - 1024);
        for (int bw = 0; bw < 2; ++bw) {
            snprintf(d, sizeof d, "\"what\": \"lin U=4\", \"wgs\": %d, \"wgs_per_cu\": %d, \"walk\": \"%s\", \"window_GiB\": %.1f", W, per_cu, bw ? "block" : "stride", gib);
            const size_t nc = n4 / (4 * 256);
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lin_ts<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
            run("occ", d, 2.0 * nc * 4 * 256 * 16, [&] { k_lin_ts<4><<<dim3(W), dim3(256), lds>>>(A, B, nc, bw, stamps); }, nullptr, [&] { return spread(stamps, W, 2, false); });
        }
    }
    // ---- persistent frame walk, stamped (245 x 512 x 2 = the pass; 489 x 512 x 1 two per CU; 489 x 256 x 2; 977 x 256 x 1)
    const uint32_t n_atoms = 1000000u, ntiles = (n_atoms + 255) / 256, ngroups = ntiles * 64;
    const size_t stride4 = (size_t)ntiles * 192, stride_f = stride4 * 4;
    const uint32_t F = std::min<uint32_t>(768u, (uint32_t)(n4 / stride4));
    auto prow = [&](const char *what, int lanes, int G, int lds, auto kern) {
        const int W = (int)((ngroups + lanes * G - 1) / (lanes * G));
        snprintf(d, sizeof d, "\"what\": \"%s\", \"wgs\": %d, \"lanes\": %d, \"groups_per_lane\": %d, \"lds_ballast\": %d, \"frames\": %u, \"place\": \"oop\"", what, W, lanes, G, lds, F);
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        run("ts", d, 2.0 * F * (double)stride4 * 16, [&] { kern<<<dim3(W), dim3(lanes), lds>>>((const float *)A, (float *)B, stride_f, F, ngroups, stamps); }, nullptr, [&] { return spread(stamps, W, 3, true); });
    };
    prow("persist 512x2, one per CU", 512, 2, 150 * 1024, k_persist_ts<512, 2>);
    prow("persist 512x2", 512, 2, 0, k_persist_ts<512, 2>);
    prow("persist 512x1", 512, 1, 0, k_persist_ts<512, 1>);
    prow("persist 256x2", 256, 2, 0, k_persist_ts<256, 2>);
    prow("persist 256x1", 256, 1, 0, k_persist_ts<256, 1>);
    prow("persist 1024x1", 1024, 1, 0, k_persist_ts<1024, 1>);
    return 0;
}
