// store_variants.hip -- which cache policy / launch shape writes a 12 MB frame fastest?  (round 4: the persistent copy of the resident
// pass's shape pays 2.26 us for the stores of a 1e6-atom frame where a grid-launched store stream pays ~1.9.)
// Rows: policy bits of buffer_store_dwordx4 (sc0 / sc1 / nt) x { persistent 245 x 512 x 2 store-only, persistent copy, grid store-only,
// grid copy }; loads nt unless stated.  Buffer instructions through the compiler's builtins, so that every load and store is
// counted by the compiler's own s_waitcnt bookkeeping (hand-written asm loads are not: a first version of this file faulted).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/store_variants tools/store_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
#define RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void *)(ptr), 0, (int)(bytes), 0x00020000)

__device__ __forceinline__ i4 touch(i4 v) { v.x += 1; v.y ^= 3; v.z += 5; v.w ^= 7; return v; }

// persistent: 512 lanes x 2 groups, every lane walks all frames.  COPY: loads (policy LP, one frame ahead) + stores (policy P).
template <int P, int LP, bool COPY>
__global__ __launch_bounds__(512) void k_persist(float *frames, size_t stride, uint32_t nframes, uint32_t ngroups) {
    const uint32_t base = blockIdx.x * 1024;
    uint32_t b[2]; bool ok[2];
    for (int q = 0; q < 2; ++q) { const uint32_t g = base + q * 512 + threadIdx.x; ok[q] = g < ngroups; const uint32_t gg = ok[q] ? g : 0u; b[q] = ((gg >> 6) * 192 + (gg & 63)) * 16u; }
    const uint32_t fbytes = (uint32_t)(stride * 4);
    i4 cur[2][3], nxt[2][3];
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 3; ++r) { cur[q][r] = i4{ 1, 2, 3, (int)threadIdx.x }; nxt[q][r] = cur[q][r]; }
    if (COPY) { __amdgpu_buffer_rsrc_t s = RSRC(frames, fbytes); for (int q = 0; q < 2; ++q) if (ok[q]) for (int r = 0; r < 3; ++r) cur[q][r] = __builtin_amdgcn_raw_buffer_load_b128(s, b[q] + 1024 * r, 0, LP); }
    for (uint32_t i = 0; i < nframes; ++i) {
        if (COPY && i + 1 < nframes) { __amdgpu_buffer_rsrc_t s = RSRC(frames + (size_t)(i + 1) * stride, fbytes); for (int q = 0; q < 2; ++q) if (ok[q]) for (int r = 0; r < 3; ++r) nxt[q][r] = __builtin_amdgcn_raw_buffer_load_b128(s, b[q] + 1024 * r, 0, LP); }
        __amdgpu_buffer_rsrc_t d = RSRC(frames + (size_t)i * stride, fbytes);
        for (int q = 0; q < 2; ++q) if (ok[q]) for (int r = 0; r < 3; ++r) { cur[q][r] = touch(cur[q][r]); __builtin_amdgcn_raw_buffer_store_b128(cur[q][r], d, b[q] + 1024 * r, 0, P); }
        if (COPY) for (int q = 0; q < 2; ++q) for (int r = 0; r < 3; ++r) cur[q][r] = nxt[q][r];
    }
}
// grid-launched: (x, frame), 256 lanes, grid-stride over the frame's groups
template <int P, int LP, bool COPY>
__global__ __launch_bounds__(256) void k_grid(float *frames, size_t stride, uint32_t ngroups) {
    __amdgpu_buffer_rsrc_t d = RSRC(frames + (size_t)blockIdx.y * stride, (uint32_t)(stride * 4));
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += gridDim.x * 256) {
        const uint32_t b = ((g >> 6) * 192 + (g & 63)) * 16u;
        i4 v[3];
        for (int r = 0; r < 3; ++r) v[r] = COPY ? __builtin_amdgcn_raw_buffer_load_b128(d, b + 1024 * r, 0, LP) : i4{ 1, 2, 3, (int)g };
        for (int r = 0; r < 3; ++r) { v[r] = touch(v[r]); __builtin_amdgcn_raw_buffer_store_b128(v[r], d, b + 1024 * r, 0, P); }
    }
}

template <int P, int LP> int run(const char *pname, float *F, size_t stride, uint32_t frames, uint32_t ngroups, hipEvent_t e0, hipEvent_t e1, bool &first) {
    for (int kind = 0; kind < 4; ++kind) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipEventRecord(e0));
            if (kind == 0) k_persist<P, LP, false><<<245, 512>>>(F, stride, frames, ngroups);
            else if (kind == 1) k_persist<P, LP, true><<<245, 512>>>(F, stride, frames, ngroups);
            else if (kind == 2) k_grid<P, LP, false><<<dim3(512, frames), 256>>>(F, stride, ngroups);
            else k_grid<P, LP, true><<<dim3(512, frames), 256>>>(F, stride, ngroups);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        static const char *kn[4] = { "persistent store-only", "persistent copy", "grid store-only", "grid copy" };
        printf("%s {\"stores\": \"%s\", \"shape\": \"%s\", \"us_per_frame\": %.3f}", first ? " " : ",\n ", pname, kn[kind], 1e3 * best / frames);
        fflush(stdout);
        first = false;
    }
    return 0;
}

int main(int argc, char **argv) {
    const uint32_t n = 1000000u, frames = argc > 1 ? (uint32_t)atoi(argv[1]) : 256u;
    const uint32_t ntiles = (n + 255) / 256, ngroups = ntiles * 64;
    const size_t stride = (size_t)ntiles * 768;
    float *F;
    CHECK(hipMalloc(&F, stride * frames * sizeof(float)));
    CHECK(hipMemset(F, 0, stride * frames * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bool first = true;
    printf("{\"n_atoms\": %u, \"frames_per_launch\": %u, \"loads\": \"nt unless stated\", \"results\": [\n", n, frames);
    // aux bits: 1 sc0, 2 nt, 16 sc1
    if (run<0, 2>("plain", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<2, 2>("nt", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<16, 2>("sc1", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<17, 2>("sc0 sc1", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<19, 2>("sc0 sc1 nt", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<1, 2>("sc0", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<18, 2>("sc1 nt", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<3, 2>("sc0 nt", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<18, 18>("sc1 nt (loads sc1 nt)", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<18, 19>("sc1 nt (loads sc0 sc1 nt)", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<18, 3>("sc1 nt (loads sc0 nt)", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<18, 16>("sc1 nt (loads sc1)", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<2, 0>("nt (loads plain)", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<2, 17>("nt (loads sc0 sc1)", F, stride, frames, ngroups, e0, e1, first)) return 1;
    if (run<2, 19>("nt (loads sc0 sc1 nt)", F, stride, frames, ngroups, e0, e1, first)) return 1;
    printf("\n]}\n");
    return 0;
}
