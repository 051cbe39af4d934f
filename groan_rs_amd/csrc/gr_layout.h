// gr_layout.h -- the "pair-tiled" layout of frame slots and RMSD plans in HBM, and the accessors every kernel goes through.
//
// Atoms are taken in tiles of 256 (a slot holds n_pad = n_atoms rounded up to 256 of them); lane L of a wave owns atoms
// 4L .. 4L+3 of a tile, and the tile is three ROWS of 64 float4 -- lane L's float4 of row r sits at float4 index
// tile * 192 + r * 64 + L:
//       row 0 = (x0, x1, y0, y1)     row 1 = (z0, z1, x2, x3)     row 2 = (y2, y3, z2, z3)
// (see gr_kernels.h for why).  The reference stores positions inside ~230-byte Atom records (atom.rs:23-71) and the
// decoders deliver packed rvec[n] (molly_xtc.rs:294-307); k_tile / k_untile are the only places that format appears on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// float index of component c (0 x, 1 y, 2 z) of atom i inside a slot / a plan (see the header of this file)
__host__ __device__ __forceinline__ size_t gr_tile_index(size_t i, int c) {
    const size_t s = ((i >> 1) & 1u) * 6u + 2u * (size_t)c + (i & 1u);
    return (i >> 8) * 768u + (s >> 2) * 256u + ((i >> 2) & 63u) * 4u + (s & 3u);
}
__device__ __forceinline__ void gr_pos_load(const float *__restrict__ xyz, size_t i, float &x, float &y, float &z) {
    const size_t b = gr_tile_index(i, 0);            // y: + 2 floats inside the row or + 250 into the next, z likewise: one index computation
    x = xyz[b]; y = xyz[gr_tile_index(i, 1)]; z = xyz[gr_tile_index(i, 2)];
}
__device__ __forceinline__ void gr_pos_store(float *__restrict__ xyz, size_t i, float x, float y, float z) {
    xyz[gr_tile_index(i, 0)] = x; xyz[gr_tile_index(i, 1)] = y; xyz[gr_tile_index(i, 2)] = z;
}
// float4 index of row r of the 4-atom group g (atoms 4g .. 4g+3)
__device__ __forceinline__ size_t gr_row_index(size_t g, int r) { return (g >> 6) * 192u + (size_t)r * 64u + (g & 63u); }
typedef float gr_f4 __attribute__((ext_vector_type(4)));
// Frame data is streamed (each byte is touched once per pass): the non-temporal hint keeps it from pushing the reference
// coordinates / masses, which every frame re-reads, out of the 4 MiB L2 of the XCD.
__device__ __forceinline__ float4 gr_stream_load(const float4 *p) {
    const gr_f4 v = __builtin_nontemporal_load(reinterpret_cast<const gr_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gr_stream_store(float4 *p, const float4 &v) {
    gr_f4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<gr_f4 *>(p));
}
template <bool NT = false>
__device__ __forceinline__ void gr_rows_load(const float4 *__restrict__ f4, size_t g, float4 &r0, float4 &r1, float4 &r2) {
    const size_t b = gr_row_index(g, 0);
    if (NT) { r0 = gr_stream_load(f4 + b); r1 = gr_stream_load(f4 + b + 64); r2 = gr_stream_load(f4 + b + 128); }
    else { r0 = f4[b]; r1 = f4[b + 64]; r2 = f4[b + 128]; }
}
template <bool NT = false>
__device__ __forceinline__ void gr_rows_store(float4 *__restrict__ f4, size_t g, const float4 &r0, const float4 &r1, const float4 &r2) {
    const size_t b = gr_row_index(g, 0);
    if (NT) { gr_stream_store(f4 + b, r0); gr_stream_store(f4 + b + 64, r1); gr_stream_store(f4 + b + 128, r2); }
    else { f4[b] = r0; f4[b + 64] = r1; f4[b + 128] = r2; }
}
// Frame rows through BUFFER loads: the frame's slot is a buffer resource (its base a wave-uniform pointer, its size the slot's),
// the lane's rows 32-bit byte offsets into it.  What that buys the streaming loop is loads WITHOUT CONDITIONS: a turn that does not
// exist gets a resource of zero records, a group that does not exist an offset beyond every slot -- the hardware's bounds check
// returns zeros and touches no memory.  With `if (turn exists) load` / `if (group B exists) load` the compiler has to merge loaded
// and not-loaded values where the branches join, does it with register copies, and waits for the loads it has just issued in
// order to copy them (round 4 found `s_waitcnt vmcnt(3)` + six moves right behind the row request of every other turn).
typedef int gr_i4 __attribute__((ext_vector_type(4)));
#define GR_BUF_NOWHERE 0xFFFFF000u      /* byte offset of a group that does not exist: out of range of every slot (+ 2 KiB of row offsets) */
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gr_buf_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 gr_buf_load_stream(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    const gr_i4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 2 /* nt */);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ float gr_buf_load_f32(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
// ... with a wave-uniform part of the address in an SGPR (`soff`, bytes): a walk whose lanes keep their place inside a tile and move from
// tile to tile spends no vector instruction on addresses (k_sums_pk)
template <int AUX /* 0 plain, 2 nt */>
__device__ __forceinline__ float4 gr_buf_load_f4(__amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t soff) {
    const gr_i4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)soff, AUX);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
// rows <-> the four atoms of the group
__device__ __forceinline__ void gr_rows_unpack(const float4 &r0, const float4 &r1, const float4 &r2, float (&x)[4], float (&y)[4], float (&z)[4]) {
    x[0] = r0.x; x[1] = r0.y; y[0] = r0.z; y[1] = r0.w; z[0] = r1.x; z[1] = r1.y; x[2] = r1.z; x[3] = r1.w; y[2] = r2.x; y[3] = r2.y; z[2] = r2.z; z[3] = r2.w;
}
__device__ __forceinline__ void gr_rows_pack(const float (&x)[4], const float (&y)[4], const float (&z)[4], float4 &r0, float4 &r1, float4 &r2) {
    r0 = make_float4(x[0], x[1], y[0], y[1]); r1 = make_float4(z[0], z[1], x[2], x[3]); r2 = make_float4(y[2], y[3], z[2], z[3]);
}

// packed rvec[n] records (the C ABI's frame format) <-> a slot; one lane per 4-atom group, `aos` holds n_pad records
__global__ __launch_bounds__(256) void k_tile(const float *__restrict__ aos, float *__restrict__ slot, uint32_t n_groups) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_groups) return;
    const float4 *q = reinterpret_cast<const float4 *>(aos) + 3 * (size_t)g;
    float4 a = q[0], b = q[1], c = q[2];
    // an atom WITHOUT POSITION is NaN in x (groan_hip.h); inside the library its y and z are NaN as well, whatever the caller sent: kernels that
    // treat the three coordinates of an atom in different lanes (k_translate_wrap_rows) then leave such an atom alone without asking for its x
    const float qn = __uint_as_float(0x7fc00000u);
    if (a.x != a.x) { a.y = qn; a.z = qn; }     // atom 0: a.x a.y a.z
    if (a.w != a.w) { b.x = qn; b.y = qn; }     // atom 1: a.w b.x b.y
    if (b.z != b.z) { b.w = qn; c.x = qn; }     // atom 2: b.z b.w c.x
    if (c.y != c.y) { c.z = qn; c.w = qn; }     // atom 3: c.y c.z c.w
    gr_rows_store<true>(reinterpret_cast<float4 *>(slot), g, make_float4(a.x, a.w, a.y, b.x), make_float4(a.z, b.y, b.z, c.y), make_float4(b.w, c.z, c.x, c.w));
}
__global__ __launch_bounds__(256) void k_untile(const float *__restrict__ slot, float *__restrict__ aos, uint32_t n_groups) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_groups) return;
    float4 r0, r1, r2;
    gr_rows_load<true>(reinterpret_cast<const float4 *>(slot), g, r0, r1, r2);
    float4 *q = reinterpret_cast<float4 *>(aos) + 3 * (size_t)g;
    q[0] = make_float4(r0.x, r0.z, r1.x, r0.y); q[1] = make_float4(r0.w, r1.y, r1.z, r2.x); q[2] = make_float4(r2.z, r1.w, r2.y, r2.w);
}

