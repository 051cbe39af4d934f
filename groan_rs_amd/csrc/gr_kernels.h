// gr_kernels.h -- hand-written HIP kernels for gfx950 (MI355X / CDNA4): 64-wide wavefronts,
// HBM-bound vector math (no MFMA: nothing here is a dense contraction).
//
// Data layout in HBM
//   frame slot : "pair-tiled" structure of arrays.  Atoms are taken in tiles of 256 (n_pad = n_atoms rounded up to 256); lane L
//                of a wave owns atoms 4L .. 4L+3 of a tile, and the tile is three ROWS of 64 float4 -- lane L's float4 of
//                row r sits at float4 index tile * 192 + r * 64 + L:
//                    row 0 = (x0, x1, y0, y1)     row 1 = (z0, z1, x2, x3)     row 2 = (y2, y3, z2, z3)
//                A wave moves a tile as three fully coalesced 1-KiB loads / stores (one 16-byte element per lane and
//                instruction: the access shape that streams fastest on this chip), every 128-byte line is touched by exactly
//                one instruction, and each register pair a lane receives is (coordinate of atom a, same coordinate of atom
//                a + 1) -- the operand form of the packed-f32 instructions (v_pk_fma_f32 ...) the hot kernels are written in.
//                The C ABI still takes and returns the packed rvec[n] records the xtc / trr decoders deliver
//                (molly_xtc.rs:294-307): k_tile / k_untile convert on the way in and out (one extra 24 B/atom pass on the copy
//                stream, hidden behind the 50 GB/s PCIe copy); the device-side xtc / trr unpackers write the layout directly.
//                Round 1 kept the packed records in HBM: lanes striding 48 B made every line pass the L1 three times (the
//                sums pass ran at 4.9 TB/s however few instructions it issued), and the LDS transpose that repairs the access
//                pattern costs more than it saves (measured round 2: 2.3 vs 1.7 us per 1e6-atom frame).
//   masses     : float m[n_pad]       (separate array; NaN = no mass)
//   selection  : one contiguous block  -> {start, n}            (row path, 4 atoms / lane / trip)
//                anything else         -> uint32 idx[n] in HBM  (gather path, 1 atom / lane / trip)
//   plan       : reference coordinates minus the reference box centre in the same pair-tiled layout, indexed by
//                ordinal + offset so that the plan's tiles line up with the frame's; float w[s_pad]
//
// Reductions: per-lane fp64 accumulators -> wave __shfl_down tree -> LDS across the 4 waves of a
// workgroup -> one partial record per workgroup in HBM -> a one-workgroup finalize kernel sums the
// records in a fixed order (no atomics: results are bitwise reproducible run to run).
#pragma once
#include <hip/hip_runtime.h>
#include "gr_math.h"
#include "gr_layout.h"
#include "gr_rotation.h"

#define GR_WG 256
#define GR_NOIDX 0xFFFFFFFFu
#define GR_ST_FALLBACK 100 /* internal: frame must be redone on the multi-pass exact path */
#define GR_ST_REDO_EXACT 103 /* internal (RMSD without fit, f32 chains): rmsd too close to the rounding of its own sums -> the exact-product pass */
#define GR_ST_AMBIG 101    /* internal (one-pass centre): images proven, but the periodic copy needs the Bai-Breen estimate itself */

struct GrSel {
    uint32_t n;            // atoms in the selection
    uint32_t contiguous;   // 1: atoms start .. start+n-1
    uint32_t start;        // first atom (contiguous) / first atom of idx (gather)
    uint32_t g0;           // contiguous: first 256-atom tile (start / 256)
    const uint32_t *idx;   // gather list (device), NULL when contiguous
    // a DENSE non-contiguous selection (every third atom, a few large blocks: at least an eighth of the atoms between its first and its
    // last one): besides the gather list it carries one bit per atom of the system, and the streaming kernels that know about it
    // (k_sums_pk<.., MASK>) walk the whole SPAN start .. start + span - 1 with coalesced row loads, turning the atoms whose bit is
    // clear into what a ragged end's atoms become -- copies of the first atom with zero mass and zero reference
    uint32_t masked;       // bit 0: `mask` is set (contiguous is 0: every other kernel takes the gather list); bit 1: translate / wrap walk the span too
    uint32_t span;         // atoms from the first to the last selected one (contiguous: n)
    const uint32_t *mask;  // bit (a & 31) of word a >> 5: atom a is selected
};

// per-frame state shared by the stages of one analysis (lives in HBM, one record per frame of a batch)
struct GrFrameState {
    float center[3];   // centre used for unwrapping (Bai-Breen estimate)
    float com[3];      // result centre / centre of mass
    float shift[3];    // box centre - com
    float R[9];        // optimal rotation, column-major
    float rmsd;
    int status;
    uint32_t err_index;
    uint32_t pad;
};

// constants of an RMSD plan (device copy)
struct GrPlanDev {
    const float *p;    // [s_pad][3]
    const float *w;    // [s_pad]
    double sp[3];      // sum p
    double swp[3];     // sum w p
    double swpp;       // sum w |p|^2
    double sw;         // sum w
    float ref_com[3];  // reference.group_get_com(group)
    uint32_t n;        // atoms in the reference group
    uint32_t w_is_mass; // weights equal the target masses of the group: one load serves both
    float fast_sigmas;  // k_sums_pk<false, true>: the closing step keeps a frame when this many sigma of its rounding estimate move the rmsd by < 2.5e-6 nm (0: keeps every frame -- calibration runs only)
};

// ------------------------------------------------------------------------------------------ reductions
template <int K>
__device__ __forceinline__ void gr_block_sum(double (&v)[K], double *lds /* [GR_WG/64][K] */) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        v[k] = x;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double s = lds[k];
            for (int wv = 1; wv < GR_WG / 64; ++wv) s += lds[wv * K + k];
            v[k] = s;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void gr_wave_sync() {
    // LDS operations of one wavefront execute in order; this only stops the compiler from moving
    // LDS accesses across the point where lanes exchange data
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double gr_wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;   // lane 0 holds the total
}

__device__ __forceinline__ uint32_t gr_block_min_u32(uint32_t x, uint32_t *lds /* [GR_WG/64] */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { uint32_t y = __shfl_down(x, off, 64); x = y < x ? y : x; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = x;
    __syncthreads();
    if (threadIdx.x == 0) for (int wv = 1; wv < GR_WG / 64; ++wv) x = lds[wv] < x ? lds[wv] : x;
    __syncthreads();
    return x;
}

__device__ __forceinline__ float gr_block_min_f32(float x, float *lds) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = gr_fminf(x, __shfl_down(x, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = x;
    __syncthreads();
    if (threadIdx.x == 0) for (int wv = 1; wv < GR_WG / 64; ++wv) x = gr_fminf(x, lds[wv]);
    __syncthreads();
    return x;
}

// Stage the frame's box into LDS once per workgroup (box + image table = 316 bytes).
__device__ __forceinline__ void gr_stage_box(GrBox *lds_box, const GrBox *g) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(g);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds_box);
    for (int i = threadIdx.x; i < (int)(sizeof(GrBox) / 4); i += GR_WG) dst[i] = src[i];
    __syncthreads();
}

// Wave reduce-scatter of 32 floats: afterwards lane l holds the wave total of value (l >> 1).  Each step halves the
// values a lane still carries (it sends the half its partner keeps): 16+8+4+2+1+1 exchanges instead of 32 x 6.
// Steps are template instances so every register-array index is a compile-time constant.
// the value of lane (l ^ M) for M = 1, 2, 4, 8 as data-parallel-primitive moves inside the VALU (quad permutes, a row rotation by
// 8, two masked row shifts for 4): no trip through the LDS crossbar (ds_bpermute: ~100 cycles, and the reductions chain 4-5 of them)
template <int M> __device__ __forceinline__ float gr_xor_lane(float v) {
    const int x = __float_as_int(v);
    int r;
#ifdef GR_NO_DPP
    return __int_as_float(__shfl_xor(x, M, 64));
#endif
    if (M == 1) r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);                 // quad_perm [1,0,3,2]
    else if (M == 2) r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);            // quad_perm [2,3,0,1]
    else if (M == 4) { r = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);         // row_shl:4 into banks 0, 2
                       r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xf, 0xA, false); }       // row_shr:4 into banks 1, 3
    else if (M == 8) r = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);           // row_ror:8
    else r = __shfl_xor(x, M, 64);
    return __int_as_float(r);
}
// sum of a float over the 64 lanes, in every lane: four DPP steps, then the two lane swaps (swap(x, x) puts the two halves /
// the odd and even rows side by side) -- no LDS crossbar anywhere
__device__ __forceinline__ float gr_wave_allsum_f32(float x) {
    x += gr_xor_lane<1>(x); x += gr_xor_lane<2>(x); x += gr_xor_lane<4>(x); x += gr_xor_lane<8>(x);
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}
template <int HALF, int MASK, bool MAX>
__device__ __forceinline__ void gr_rs_step(float (&a)[32], const uint32_t lane) {
    const bool hi = (lane & MASK) != 0;
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
        const float send = hi ? a[k] : a[k + HALF];
        const float keep = hi ? a[k + HALF] : a[k];
        const float got = gr_xor_lane<MASK>(send);
        a[k] = MAX ? gr_fmaxf(keep, got) : keep + got;
    }
}
// the two widest steps (partner 32 / 16 lanes away) as ONE lane-swap per pair of registers: v_permlane32_swap exchanges the
// upper half of a[k] with the lower half of a[k + HALF] (v_permlane16_swap: the odd 16-lane rows of a[k] with the even rows of
// a[k + HALF]), after which a[k] op a[k + HALF] is, in every lane, exactly "what I keep + what my partner sent" -- the same two
// operands as gr_rs_step adds, so the results are bit-identical; 2 instructions per exchange instead of 4 (and no LDS crossbar)
template <int HALF, int MASK, bool MAX>
__device__ __forceinline__ void gr_rs_step_swap(float (&a)[32]) {
    static_assert(MASK == 32 || MASK == 16, "lane swaps exist for 32 and 16");
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
        const auto r = MASK == 32 ? __builtin_amdgcn_permlane32_swap(__float_as_uint(a[k]), __float_as_uint(a[k + HALF]), false, false)
                                  : __builtin_amdgcn_permlane16_swap(__float_as_uint(a[k]), __float_as_uint(a[k + HALF]), false, false);
        const float x = __uint_as_float(r[0]), y = __uint_as_float(r[1]);
        a[k] = MAX ? gr_fmaxf(x, y) : x + y;
    }
}
template <int HALF, int MASK>
__device__ __forceinline__ void gr_rs_step_f64(double (&a)[32], const uint32_t lane) {
    const bool hi = (lane & MASK) != 0;
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
        const double send = hi ? a[k] : a[k + HALF];
        const double keep = hi ? a[k + HALF] : a[k];
        a[k] = keep + __shfl_xor(send, MASK, 64);
    }
}
// 32 doubles: afterwards lane l holds the wave total of value (l >> 1)
__device__ __forceinline__ double gr_wave_sum_scatter32_f64(double (&a)[32], const uint32_t lane) {
    gr_rs_step_f64<16, 32>(a, lane); gr_rs_step_f64<8, 16>(a, lane); gr_rs_step_f64<4, 8>(a, lane);
    gr_rs_step_f64<2, 4>(a, lane); gr_rs_step_f64<1, 2>(a, lane);
    return a[0] + __shfl_xor(a[0], 1, 64);
}
__device__ __forceinline__ float gr_wave_sum_scatter32(float (&a)[32], const uint32_t lane) {
    gr_rs_step_swap<16, 32, false>(a); gr_rs_step_swap<8, 16, false>(a); gr_rs_step<4, 8, false>(a, lane);
    gr_rs_step<2, 4, false>(a, lane); gr_rs_step<1, 2, false>(a, lane);
    return a[0] + gr_xor_lane<1>(a[0]);
}
// 8 doubles, without the LDS crossbar (the steps above move a double as two ds_bpermute: ~34 of them per 16 values, and eight waves of a
// workgroup queue for the one crossbar): lane swaps for the partners 32 and 16 lanes away and DPP moves for 8, 4, 2, 1, a double as its two
// 32-bit halves.  Afterwards lane l holds the wave total of value (l >> 3) (the lanes of a group of 8 all hold it).
__device__ __forceinline__ double gr_f64_from_halves(uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
template <int M> __device__ __forceinline__ double gr_xor_lane_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const float lo = gr_xor_lane<M>(__uint_as_float((uint32_t)b)), hi = gr_xor_lane<M>(__uint_as_float((uint32_t)(b >> 32)));
    return gr_f64_from_halves(__float_as_uint(lo), __float_as_uint(hi));
}
template <int HALF, int MASK>
__device__ __forceinline__ void gr_rs_step_swap_f64(double (&a)[8]) {
    static_assert(MASK == 32 || MASK == 16, "lane swaps exist for 32 and 16");
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
        const unsigned long long p = (unsigned long long)__double_as_longlong(a[k]), q = (unsigned long long)__double_as_longlong(a[k + HALF]);
        const auto rl = MASK == 32 ? __builtin_amdgcn_permlane32_swap((uint32_t)p, (uint32_t)q, false, false) : __builtin_amdgcn_permlane16_swap((uint32_t)p, (uint32_t)q, false, false);
        const auto rh = MASK == 32 ? __builtin_amdgcn_permlane32_swap((uint32_t)(p >> 32), (uint32_t)(q >> 32), false, false) : __builtin_amdgcn_permlane16_swap((uint32_t)(p >> 32), (uint32_t)(q >> 32), false, false);
        a[k] = gr_f64_from_halves(rl[0], rh[0]) + gr_f64_from_halves(rl[1], rh[1]);
    }
}
__device__ __forceinline__ double gr_wave_sum_scatter8_f64(double (&a)[8], const uint32_t lane) {
    gr_rs_step_swap_f64<4, 32>(a);                 // lanes 0-31: values 0..3, lanes 32-63: values 4..7
    gr_rs_step_swap_f64<2, 16>(a);                 // ... of which the even 16-lane rows keep the first two, the odd rows the other two
    {                                              // partner 8 lanes away: the lower half of a row keeps a[0], the upper half a[1]
        const bool hi = (lane & 8u) != 0;
        const double send = hi ? a[0] : a[1], keep = hi ? a[1] : a[0];
        a[0] = keep + gr_xor_lane_f64<8>(send);
    }
    double v = a[0];
    v += gr_xor_lane_f64<4>(v); v += gr_xor_lane_f64<2>(v); v += gr_xor_lane_f64<1>(v);
    return v;
}
// 16 doubles (a[0..15]): afterwards lane l holds the wave total of value (l >> 2)
__device__ __forceinline__ double gr_wave_sum_scatter16_f64(double (&a)[32], const uint32_t lane) {
    gr_rs_step_f64<8, 32>(a, lane); gr_rs_step_f64<4, 16>(a, lane); gr_rs_step_f64<2, 8>(a, lane); gr_rs_step_f64<1, 4>(a, lane);
    const double m = a[0] + __shfl_xor(a[0], 2, 64);
    return m + __shfl_xor(m, 1, 64);
}
// 16 floats: afterwards lane l holds the wave total of value (l >> 2)
__device__ __forceinline__ float gr_wave_sum_scatter16(float (&a)[32], const uint32_t lane) {
    gr_rs_step_swap<8, 32, false>(a); gr_rs_step_swap<4, 16, false>(a); gr_rs_step<2, 8, false>(a, lane); gr_rs_step<1, 4, false>(a, lane);
    const float m = a[0] + gr_xor_lane<2>(a[0]);
    return m + gr_xor_lane<1>(m);
}
// the same with max over the first 16 floats: lane l ends with the wave maximum of value (l >> 2)
__device__ __forceinline__ float gr_wave_max_scatter16(float (&a)[32], const uint32_t lane) {
    gr_rs_step_swap<8, 32, true>(a); gr_rs_step_swap<4, 16, true>(a); gr_rs_step<2, 8, true>(a, lane); gr_rs_step<1, 4, true>(a, lane);
    const float m = gr_fmaxf(a[0], gr_xor_lane<2>(a[0]));
    return gr_fmaxf(m, gr_xor_lane<1>(m));
}

// ------------------------------------------------------------------------------------------ centres
// two f32 values in one register pair: gfx950 issues v_pk_add / mul / fma_f32 at full rate, i.e. two atoms per instruction
typedef float gr_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gr_v2f gr_v2(float a) { gr_v2f r = { a, a }; return r; }
__device__ __forceinline__ gr_v2f gr_v2_fma(gr_v2f a, gr_v2f b, gr_v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float gr_min3f(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// The Bai-Breen terms of TWO atoms at a time (gr_center_atom<1> below, operation for operation -- the same IEEE operations in the same
// order, hence the same bits): fractional ("u") coordinates in a non-orthogonal cell (reciprocal + one Newton step), theta = x (2 pi / L),
// sine and cosine from the hardware in revolutions with the first-order correction for the rounding of theta / 2 pi.  The positions must
// lie inside the cell already (the caller wraps those that do not).  Box constants are wave-uniform (SGPR operands).
struct GrBbBox { float by, cz, bx, cx, cy, iby, icz, scx, scy, scz; bool tric; };
__device__ __forceinline__ void gr_bb_angles_pair(gr_v2f x, gr_v2f y, gr_v2f z, const GrBbBox &b, gr_v2f (&sn)[3], gr_v2f (&cs)[3]) {
    if (b.tric) {
        gr_v2f sc = z * gr_v2(b.icz);
        sc = gr_v2_fma(gr_v2_fma(-sc, gr_v2(b.cz), z), gr_v2(b.icz), sc);
        const gr_v2f uy = gr_v2_fma(-sc, gr_v2(b.cy), y);
        gr_v2f sb = uy * gr_v2(b.iby);
        sb = gr_v2_fma(gr_v2_fma(-sb, gr_v2(b.by), uy), gr_v2(b.iby), sb);
        x = gr_v2_fma(-sc, gr_v2(b.cx), gr_v2_fma(-sb, gr_v2(b.bx), x));
        y = uy;
    }
    const float IH = 0.15915493667125702f, IL = 6.4206382432985265e-09f, TWO_PI = 6.283185307179586f;
    const gr_v2f th[3] = { x * gr_v2(b.scx), y * gr_v2(b.scy), z * gr_v2(b.scz) };
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const gr_v2f u = th[a] * gr_v2(IH);
        const gr_v2f e = gr_v2_fma(th[a], gr_v2(IL), gr_v2_fma(th[a], gr_v2(IH), -u));
        gr_v2f s, c;
        s.x = __builtin_amdgcn_sinf(u.x); s.y = __builtin_amdgcn_sinf(u.y); c.x = __builtin_amdgcn_cosf(u.x); c.y = __builtin_amdgcn_cosf(u.y);
        const gr_v2f k = gr_v2(TWO_PI) * e;
        sn[a] = gr_v2_fma(k, c, s); cs[a] = gr_v2_fma(-k, s, c);
    }
}
// kind 0: naive sums  sum(m x), sum(m) [or count]            iterators.rs:886-903,946-967
// kind 1: Bai-Breen   sum(m cos th), sum(m sin th) per axis  iterators.rs:1152-1191,1314-1357
// kind 2: unwrapped about state.center: sum(m (c + vector_to(c, x))), sum(m)   :1237-1266,1404-1438
// Partial record per workgroup: 8 doubles + {min idx without position, min idx without mass}.
#define GR_CEN_K 8
struct GrCenPartial { double s[GR_CEN_K]; uint32_t bad_pos, bad_mass; };

// one atom's terms of a centre, added to the f32 partials p[0..6] (4 atoms per partial on k_center_sums' contiguous path); shared by
// k_center_sums and the single-wave kernel of gr_small.h (sc* = 2 pi / box length for the Bai-Breen angles, c* = the centre to unwrap about)
template <int KIND>
__device__ __forceinline__ void gr_center_atom(uint32_t i, float x, float y, float z, float m, const int weighted, const GrBox &box,
                                               const float scx, const float scy, const float scz, const float cx, const float cy, const float cz,
                                               uint32_t &bad_pos, uint32_t &bad_mass, float (&p)[7]) {
    if (weighted) { if (m != m) { bad_mass = min(bad_mass, i); m = 0.0f; } } else m = 1.0f;
    if (x != x) { bad_pos = min(bad_pos, i); return; }
    if (KIND == 0) {
        p[0] = fmaf(x, m, p[0]); p[1] = fmaf(y, m, p[1]); p[2] = fmaf(z, m, p[2]); p[3] += m;
    } else if (KIND == 1) {
        // position.wrap(simbox): an atom inside the cell is left as it is (every stage's k is 0: the reference's loops do
        // not turn), so the closed form (~25 VALU slots per axis) only runs for the lanes that need it
        if (!(x >= 0.0f && x <= box.ax && y >= 0.0f && y <= box.by && z >= 0.0f && z <= box.cz)) gr_wrap(x, y, z, box);
        if (!box.ortho) {   // fractional ("u") coordinates, scaled by the box diagonal
            // (z / cz and uy / by as reciprocal x one Newton correction: the quotient to within an ulp -- nearly always the
            // correctly rounded one -- in 3 instructions instead of the ~12 of an IEEE division; this branch has no reference
            // arithmetic to match, and the pass was VALU-bound on non-orthogonal cells: 3.5 us against 3.0)
            float sc = z * box.icz;
            sc = fmaf(fmaf(-sc, box.cz, z), box.icz, sc);
            const float uy = y - sc * box.cy;
            float sb = uy * box.iby;
            sb = fmaf(fmaf(-sb, box.by, uy), box.iby, sb);
            const float ux = x - sb * box.bx - sc * box.cx;
            x = ux; y = uy;
        }
        // the reference's own f32 angle theta = wrap(x) * (2 pi / L) (auxiliary.rs:59-84), bit for bit -- for a group spread
        // over the whole box the resultant is short and the estimate amplifies every 1e-7 in theta
        // and its sine and cosine from the hardware: v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS, so theta / 2 pi --
        // a number in [0, 1] -- needs no range reduction at all.  Measured on gfx950 over 4.2 M arguments in [0, 1)
        // (tools/microbench/vsin_accuracy.hip): max |error| 1.25e-7, rms 3.5e-8, mean 1e-13 -- libm's sinf quality, at 2
        // quarter-rate instructions per pair instead of the ~25 of a Cody-Waite reduction + two polynomials (round 2; the pass
        // was VALU-bound: 3.9 -> 2.x us per 1e6-atom frame).  u = theta / 2 pi is formed with its rounding error e (the exact
        // residual of the product + the low part of 1 / 2 pi) and the two results are corrected to first order, sin(2 pi (u + e))
        // = s + 2 pi e c: without that the rounding of u (up to 3.7e-7 rad near a full turn) is the largest error in the sum, and
        // a group spread evenly over the whole box -- the water of example.gro: a resultant of ~1 from 10 399 unit vectors --
        // turns every 1e-7 per atom into 1e-5 nm of the estimate
        float s0, c0, s1, c1, s2, c2;
        auto sincos_rev = [](float theta, float &sn, float &cs) {
            const float IH = 0.15915493667125702f, IL = 6.4206382432985265e-09f, TWO_PI = 6.283185307179586f;
            const float u = theta * IH;
            const float e = fmaf(theta, IL, fmaf(theta, IH, -u));
            const float s = __builtin_amdgcn_sinf(u), c = __builtin_amdgcn_cosf(u), k = TWO_PI * e;
            sn = fmaf(k, c, s); cs = fmaf(-k, s, c);
        };
        sincos_rev(x * scx, s0, c0); sincos_rev(y * scy, s1, c1); sincos_rev(z * scz, s2, c2);
        p[0] = fmaf(m, c0, p[0]); p[1] = fmaf(m, c1, p[1]); p[2] = fmaf(m, c2, p[2]);
        p[3] = fmaf(m, s0, p[3]); p[4] = fmaf(m, s1, p[4]); p[5] = fmaf(m, s2, p[5]);
        p[6] += 1.0f;
    } else {
        float vx, vy, vz;
        gr_vector_to(cx, cy, cz, x, y, z, box, vx, vy, vz);
        p[0] = fmaf(cx + vx, m, p[0]); p[1] = fmaf(cy + vy, m, p[1]); p[2] = fmaf(cz + vz, m, p[2]);
        p[3] += m;
    }
}

template <int KIND>   // 0 naive, 1 Bai-Breen, 2 unwrapped about state.center (compile-time: the naive sums need neither box nor branches)
__global__ __launch_bounds__(GR_WG) void k_center_sums(
    const float *__restrict__ frames, size_t frame_stride, uint32_t first_slot,
    const float *__restrict__ masses, GrSel sel, const GrBox *__restrict__ boxes,
    const GrFrameState *__restrict__ state, int weighted, GrCenPartial *__restrict__ partials, int only_status = 0) {
    __shared__ GrBox box;
    __shared__ double lds[(GR_WG / 64) * GR_CEN_K];
    __shared__ uint32_t ldsu[GR_WG / 64];
    const uint32_t frame = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    // only_status: a launch over a whole batch that works on the frames carrying this status only (the frames the one-pass
    // centre handed back, GR_ST_FALLBACK); the workgroups of every other frame leave at once
    if (only_status && state[frame].status != only_status) return;
    const float *xyz = frames + (size_t)(first_slot + frame) * frame_stride;
    if (KIND != 0) gr_stage_box(&box, boxes + first_slot + frame);
    double acc[GR_CEN_K];
#pragma unroll
    for (int k = 0; k < GR_CEN_K; ++k) acc[k] = 0.0;
    uint32_t bad_pos = GR_NOIDX, bad_mass = GR_NOIDX;
    const float PI_X2 = 3.14159265358979323846f * 2.0f;   // auxiliary.rs:15
    const float scx = KIND == 1 ? PI_X2 / box.ax : 0.f, scy = KIND == 1 ? PI_X2 / box.by : 0.f, scz = KIND == 1 ? PI_X2 / box.cz : 0.f;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (KIND == 2) { cx = state[frame].center[0]; cy = state[frame].center[1]; cz = state[frame].center[2]; }
    auto atom = [&](uint32_t i, float x, float y, float z, float m, float (&p)[7]) {
        gr_center_atom<KIND>(i, x, y, z, m, weighted, box, scx, scy, scz, cx, cy, cz, bad_pos, bad_mass, p);
    };
    // the frame's box once more as wave-uniform values (scalar loads through the kernel argument: SGPR operands of the packed arithmetic)
    GrBbBox bb;
    float bb_ax = 0.f;
    if (KIND == 1) {
        const GrBox *bp = boxes + first_slot + frame;
        bb_ax = bp->ax; bb.by = bp->by; bb.cz = bp->cz; bb.bx = bp->bx; bb.cx = bp->cx; bb.cy = bp->cy; bb.iby = bp->iby; bb.icz = bp->icz;
        bb.scx = PI_X2 / bp->ax; bb.scy = PI_X2 / bp->by; bb.scz = PI_X2 / bp->cz; bb.tric = !bp->ortho;
    }
    if (sel.contiguous) {
        // read-only stream: a lane's 4 atoms are its three row loads (coalesced: consecutive lanes, consecutive 16 bytes) + one of
        // masses; their terms form 4-atom f32 partials that go into the lane's fp64 accumulators
        const float4 *f4 = reinterpret_cast<const float4 *>(xyz);
        const float4 *m4 = reinterpret_cast<const float4 *>(masses);
        const uint32_t first = sel.start, last = sel.start + sel.n;
        const uint32_t g0 = first >> 2, g1 = (last + 3u) >> 2;
        // (rows of trip k + 1 are requested before the arithmetic of trip k: a lane has one trip's loads in flight at any time;
        // non-temporal: the frame is read once, the masses that every frame re-reads stay in the L2)
        const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f);
        uint32_t g = g0 + chunk * GR_WG + threadIdx.x;
        float4 n0 = one4, n1 = one4, n2 = one4, nm = one4;
        if (g < g1) { gr_rows_load<true>(f4, g, n0, n1, n2); if (weighted) nm = m4[g]; }
        for (; g < g1; g += nchunks * GR_WG) {
            const float4 r0 = n0, r1 = n1, r2 = n2, mm = nm;
            const uint32_t gn = g + nchunks * GR_WG;
            if (gn < g1) { gr_rows_load<true>(f4, gn, n0, n1, n2); if (weighted) nm = m4[gn]; }
            float x[4], y[4], z[4];
            gr_rows_unpack(r0, r1, r2, x, y, z);
            const float m[4] = { mm.x, mm.y, mm.z, mm.w };
            const uint32_t i = g << 2;
            float p[7] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
            if (KIND == 1) {
                // the whole wave's trip at once when nothing special happens in it (round 5): every lane's four atoms in the selection, with
                // position and mass, inside the cell -- then the terms are formed two atoms per instruction, without a branch, and added in
                // the order of the loop below (the same bits); any other trip takes that loop as before
                bool plain = i >= first && i + 3u < last && x[0] == x[0] && x[1] == x[1] && x[2] == x[2] && x[3] == x[3];
                if (weighted) plain = plain && m[0] == m[0] && m[1] == m[1] && m[2] == m[2] && m[3] == m[3];
                {
                    const float xl = fminf(fminf(x[0], x[1]), fminf(x[2], x[3])), xh = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
                    const float yl = fminf(fminf(y[0], y[1]), fminf(y[2], y[3])), yh = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
                    const float zl = fminf(fminf(z[0], z[1]), fminf(z[2], z[3])), zh = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
                    plain = plain && xl >= 0.0f && xh <= bb_ax && yl >= 0.0f && yh <= bb.by && zl >= 0.0f && zh <= bb.cz;
                }
#ifdef GR_EXP_NO_BBFAST
                plain = false;
#endif
                if (__builtin_amdgcn_ballot_w64(!plain) == 0ull) {
                    gr_v2f s01[3], c01[3], s23[3], c23[3];
                    const gr_v2f x01 = { x[0], x[1] }, y01 = { y[0], y[1] }, z01 = { z[0], z[1] }, x23 = { x[2], x[3] }, y23 = { y[2], y[3] }, z23 = { z[2], z[3] };
                    gr_bb_angles_pair(x01, y01, z01, bb, s01, c01);
                    gr_bb_angles_pair(x23, y23, z23, bb, s23, c23);
                    const float w0 = weighted ? m[0] : 1.0f, w1 = weighted ? m[1] : 1.0f, w2 = weighted ? m[2] : 1.0f, w3 = weighted ? m[3] : 1.0f;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        p[a] = fmaf(w3, c23[a].y, fmaf(w2, c23[a].x, fmaf(w1, c01[a].y, fmaf(w0, c01[a].x, 0.0f))));
                        p[3 + a] = fmaf(w3, s23[a].y, fmaf(w2, s23[a].x, fmaf(w1, s01[a].y, fmaf(w0, s01[a].x, 0.0f))));
                    }
                    p[6] = 4.0f;
#pragma unroll
                    for (int k = 0; k < 7; ++k) acc[k] += (double)p[k];
                    continue;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (i + k >= first && i + k < last) atom(i + k, x[k], y[k], z[k], m[k], p);
#pragma unroll
            for (int k = 0; k < (KIND == 1 ? 7 : 4); ++k) acc[k] += (double)p[k];
        }
    } else {
        for (uint32_t j = chunk * GR_WG + threadIdx.x; j < sel.n; j += nchunks * GR_WG) {
            const uint32_t i = sel.idx[j];
            float rx, ry, rz;
            gr_pos_load(xyz, i, rx, ry, rz);
            float p[7] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
            atom(i, rx, ry, rz, weighted ? masses[i] : 1.0f, p);
#pragma unroll
            for (int k = 0; k < (KIND == 1 ? 7 : 4); ++k) acc[k] += (double)p[k];
        }
    }
    gr_block_sum<GR_CEN_K>(acc, lds);
    bad_pos = gr_block_min_u32(bad_pos, ldsu);
    bad_mass = gr_block_min_u32(bad_mass, ldsu);
    if (threadIdx.x == 0) {
        GrCenPartial &o = partials[(size_t)frame * nchunks + chunk];
#pragma unroll
        for (int k = 0; k < GR_CEN_K; ++k) o.s[k] = acc[k];
        o.bad_pos = bad_pos; o.bad_mass = bad_mass;
    }
}

// closing step of one centre stage for one frame (one lane): error precedence, the Bai-Breen angles back to a position, targets;
// shared by k_center_finalize and the single-wave kernel of gr_small.h
__device__ __forceinline__ void gr_center_close(const double *acc, const uint32_t bad_pos, const uint32_t bad_mass, const GrBox &b, const int kind, const int weighted,
                                                const int mass_first, const int target, const uint32_t n_sel, GrFrameState &st, const int only_status) {
    // error precedence of the reference loops: estimate_com / get_com_naive test per atom
    // (mass first: iterators.rs:1324-1339; position first: :946-958); get_com runs the unweighted
    // estimate over all atoms before any mass is read (:1405-1422)
    if (bad_pos != GR_NOIDX || bad_mass != GR_NOIDX) {
        bool mass_err;
        if (bad_mass == GR_NOIDX) mass_err = false;
        else if (bad_pos == GR_NOIDX) mass_err = true;
        else mass_err = mass_first ? (bad_mass <= bad_pos) : (bad_mass < bad_pos);
        st.status = mass_err ? 7 /*GR_E_NO_MASS*/ : 6 /*GR_E_NO_POSITION*/;
        st.err_index = mass_err ? bad_mass : bad_pos;
        return;
    }
    float r[3];
    if (kind == 1) {
        const float PI_F = 3.14159265358979323846f, PI_X2 = PI_F * 2.0f;
        const float sc[3] = { PI_X2 / b.ax, PI_X2 / b.by, PI_X2 / b.cz };
        float t[3];
        for (int a = 0; a < 3; ++a) t[a] = (atan2f(-(float)acc[3 + a], -(float)acc[a]) + PI_F) / sc[a];   // auxiliary.rs:87-99
        if (b.ortho) { r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; }
        else {
            const float s_c = t[2] / b.cz, s_b = t[1] / b.by;
            r[2] = t[2]; r[1] = t[1] + s_c * b.cy; r[0] = t[0] + s_b * b.bx + s_c * b.cx;
        }
        if (only_status == GR_ST_AMBIG) {
            // the one-pass centre left its centre / com in SOME periodic copy: the reference's copy is the one about c' (= r),
            // i.e. the copy whose centre is the minimum image of the stored one as seen from c' (they are within eps of each other)
            float vx, vy, vz;
            gr_vector_to(r[0], r[1], r[2], st.center[0], st.center[1], st.center[2], b, vx, vy, vz);
            const double lat[3] = { (double)r[0] + vx - st.center[0], (double)r[1] + vy - st.center[1], (double)r[2] + vz - st.center[2] };
            for (int a = 0; a < 3; ++a) { st.center[a] = (float)((double)st.center[a] + lat[a]); st.com[a] = (float)((double)st.com[a] + lat[a]); }
            st.status = 0;
            return;
        }
    } else {
        const double div = weighted ? acc[3] : (double)n_sel;
        r[0] = (float)(acc[0] / div); r[1] = (float)(acc[1] / div); r[2] = (float)(acc[2] / div);
    }
    if (target == 0) { st.center[0] = r[0]; st.center[1] = r[1]; st.center[2] = r[2]; }
    else {
        st.com[0] = r[0]; st.com[1] = r[1]; st.com[2] = r[2];
        st.shift[0] = b.bcx - r[0]; st.shift[1] = b.bcy - r[1]; st.shift[2] = b.bcz - r[2];
        if (only_status) st.status = 0;   // the frame's last stage: done
    }
}

// one workgroup per frame; target: 0 -> state.center, 1 -> state.com (+ shift = box centre - com)
__global__ __launch_bounds__(GR_WG) void k_center_finalize(
    const GrCenPartial *__restrict__ partials, uint32_t nchunks, const GrBox *__restrict__ boxes,
    uint32_t first_slot, int kind, int weighted, int mass_first, int target, uint32_t n_sel,
    GrFrameState *__restrict__ state, int only_status = 0) {
    __shared__ double lds[(GR_WG / 64) * GR_CEN_K];
    __shared__ uint32_t ldsu[GR_WG / 64];
    const uint32_t frame = blockIdx.x;
    if (only_status && state[frame].status != only_status) return;
    double acc[GR_CEN_K];
#pragma unroll
    for (int k = 0; k < GR_CEN_K; ++k) acc[k] = 0.0;
    uint32_t bad_pos = GR_NOIDX, bad_mass = GR_NOIDX;
    for (uint32_t c = threadIdx.x; c < nchunks; c += GR_WG) {
        const GrCenPartial &p = partials[(size_t)frame * nchunks + c];
#pragma unroll
        for (int k = 0; k < GR_CEN_K; ++k) acc[k] += p.s[k];
        bad_pos = min(bad_pos, p.bad_pos); bad_mass = min(bad_mass, p.bad_mass);
    }
    gr_block_sum<GR_CEN_K>(acc, lds);
    bad_pos = gr_block_min_u32(bad_pos, ldsu);
    bad_mass = gr_block_min_u32(bad_mass, ldsu);
    if (threadIdx.x != 0) return;
    GrFrameState &st = state[frame];
    if (st.status != only_status) return;   // an earlier stage of this frame already failed
    gr_center_close(acc, bad_pos, bad_mass, boxes[first_slot + frame], kind, weighted, mass_first, target, n_sel, st, only_status);
}

__global__ void k_state_reset(GrFrameState *state, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { GrFrameState z = {}; z.err_index = GR_NOIDX; state[i] = z; }
}

// ------------------------------------------------------------------------------------------ RMSD accumulate
// One pass over the group: everything kabsch_rmsd (rmsd.rs:547-603) and get_com (iterators.rs:1404-1438)
// need, as sums that do not depend on the centre of mass:
//   v_i   = image of x_i nearest to the provisional centre g (= first atom of the group)    [MODE 0]
//         = wrap(x_i + shift) - box_centre, the reference's own q_i (rmsd.rs:479-492)       [MODE 1]
//   [0] sum m        [1..3] sum m v      [4..12] A = sum p v^T (unweighted, rmsd.rs:567-570)
//   [13..21] B = sum w p v^T             [22] sum w |v|^2      [23..25] sum w v
//   [26..31] first and second moments of the fractional coordinates of v (MODE 0 only): with the fractional
//            extent they BOUND where the reference's Bai-Breen centre estimate can lie, which is what proves
//            that the images chosen about g are the images the reference chooses about its own centre
//   min/max of v per axis (Cartesian and fractional), first atom without position / mass.
#define GR_ACC_K 32
struct GrAccPartial { double s[GR_ACC_K]; float vmin[3], vmax[3], fmin[3], fmax[3]; uint32_t bad_pos, bad_mass;
                      float rd; uint32_t pad; };   // rd (k_sums_pk<.., RMSD> only): this chunk's MEASURED rounding of its f32 chains, in units of what the random-walk model expects

// ---- per-lane state and per-atom arithmetic of the closed-form single pass
struct GrA4 { float x[4], y[4], z[4], m[4], px[4], py[4], pz[4], w[4]; uint32_t i[4]; bool ok[4]; };   // four atoms of one lane

struct GrLaneAcc {
    double acc[GR_ACC_K];
    float fsum[6];                       // first / second moments of the fractional coordinates
    float mn[3], mx[3], fmn[3], fmx[3];  // Cartesian and fractional extent of v
    uint32_t bad_pos, bad_mass;
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int k = 0; k < GR_ACC_K; ++k) acc[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) fsum[k] = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fmn[a] = 3.0e38f; mx[a] = fmx[a] = -3.0e38f; }
        bad_pos = bad_mass = GR_NOIDX;
    }
    // fold the f32 moment sums into their fp64 slots (call once, before the cross-lane reduction)
    __device__ __forceinline__ void close(bool w_is_mass) {
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[26 + k] = (double)fsum[k];
        if (w_is_mass) { acc[23] = acc[1]; acc[24] = acc[2]; acc[25] = acc[3]; }   // sum w v == sum m v
    }
};

struct GrFrameConst {   // wave-uniform per-frame constants
    float gx, gy, gz;    // provisional centre (MODE 0)
    float sx, sy, sz;    // shift = box centre - COM (MODE 1)
    float iax, iby, icz, rws2;
    bool tric, wm;
};

// v = the image of (x - g) nearest to the provisional centre g, and its fractional coordinates (single pass, MODE 0)
__device__ __forceinline__ void gr_image_about(float &vx, float &vy, float &vz, float &f_a, float &f_b, float &f_c,
                                               float x, float y, float z, const GrBox &box, const GrFrameConst &fc) {
    // closed-form brick reduction along c, b, a ...
    vx = x - fc.gx; vy = y - fc.gy; vz = z - fc.gz;
    const float kc = rintf(vz * fc.icz);
    vx = fmaf(-kc, box.cx, vx); vy = fmaf(-kc, box.cy, vy); vz = fmaf(-kc, box.cz, vz);
    const float kb = rintf(vy * fc.iby);
    vx = fmaf(-kb, box.bx, vx); vy = fmaf(-kb, box.by, vy);
    const float ka = rintf(vx * fc.iax);
    vx = fmaf(-ka, box.ax, vx);
    // ... which is already THE minimum image whenever |v| < r_ws; otherwise search the table
    if (fc.tric && vx * vx + vy * vy + vz * vz >= fc.rws2) gr_tric_refine(vx, vy, vz, box);
    // fractional coordinates of v: first and second moments feed the image proof (see gr_finalize_math)
    f_c = vz * fc.icz;
    const float uy = fmaf(-f_c, box.cy, vy);
    f_b = uy * fc.iby;
    f_a = (vx - f_b * box.bx - f_c * box.cx) * fc.iax;
}

// Precision plan.  rmsd^2 is the small difference of sums of size W r^2 (it must come out ~0 for a frame that is a
// rigid copy of the reference: the reference's own tests ask |rmsd| <= 1e-4 there, i.e. 1e-9 relative on those sums),
// so everything that enters it -- B = sum (w p) v^T, sum w|v|^2, sum w v, sum m, sum m v -- uses exact products
// (f32 x f32 is exact in fp64) and fp64 sums.  The unweighted covariance A only steers the rotation; it is formed as
// a 4-atom f32 partial (operands are centred, ~1e-7 relative, random in sign) that is then added to its fp64
// accumulator: 2.25 instead of 9 fp64 adds per atom.
// `checked` = the group straddles the ends of the selection (or is the gather path's tail): per-atom validity + NaN
// tests.  Interior groups skip them: a NaN position or mass then simply poisons the fp64 sums, which the finalize step
// detects and answers by sending the frame to the multi-pass path, whose kernels report the first atom without
// position / mass in the reference's order.
// (RMSD-fit of a contiguous selection does not come through here: its sums pass is k_sums_pk of gr_hot.h, and the fit pass
// evaluates sum w |R q - p|^2 directly.)
template <int MODE>
__device__ __forceinline__ void gr_flush4(GrLaneAcc &L, const GrA4 &a, const bool checked, const GrBox &box, const GrFrameConst &fc) {
    float part[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) part[k] = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float m = a.m[q];
        if (checked) {
            if (!a.ok[q]) continue;
            if (m != m) { L.bad_mass = min(L.bad_mass, a.i[q]); m = 0.0f; }
            if (a.x[q] != a.x[q]) { L.bad_pos = min(L.bad_pos, a.i[q]); continue; }
        }
        float vx, vy, vz;
        if (MODE == 0) {
            float f_a, f_b, f_c;
            gr_image_about(vx, vy, vz, f_a, f_b, f_c, a.x[q], a.y[q], a.z[q], box, fc);
            L.fsum[0] += f_a; L.fsum[1] += f_b; L.fsum[2] += f_c;
            L.fsum[3] = fmaf(f_a, f_a, L.fsum[3]); L.fsum[4] = fmaf(f_b, f_b, L.fsum[4]); L.fsum[5] = fmaf(f_c, f_c, L.fsum[5]);
            L.fmn[0] = gr_fminf(L.fmn[0], f_a); L.fmn[1] = gr_fminf(L.fmn[1], f_b); L.fmn[2] = gr_fminf(L.fmn[2], f_c);
            L.fmx[0] = gr_fmaxf(L.fmx[0], f_a); L.fmx[1] = gr_fmaxf(L.fmx[1], f_b); L.fmx[2] = gr_fmaxf(L.fmx[2], f_c);
        } else {
            vx = a.x[q] + fc.sx; vy = a.y[q] + fc.sy; vz = a.z[q] + fc.sz;
            gr_wrap(vx, vy, vz, box);
            vx -= box.bcx; vy -= box.bcy; vz -= box.bcz;
        }
        L.mn[0] = gr_fminf(L.mn[0], vx); L.mn[1] = gr_fminf(L.mn[1], vy); L.mn[2] = gr_fminf(L.mn[2], vz);
        L.mx[0] = gr_fmaxf(L.mx[0], vx); L.mx[1] = gr_fmaxf(L.mx[1], vy); L.mx[2] = gr_fmaxf(L.mx[2], vz);
        const float px = a.px[q], py = a.py[q], pz = a.pz[q];
        part[0] = fmaf(px, vx, part[0]); part[1] = fmaf(px, vy, part[1]); part[2] = fmaf(px, vz, part[2]);
        part[3] = fmaf(py, vx, part[3]); part[4] = fmaf(py, vy, part[4]); part[5] = fmaf(py, vz, part[5]);
        part[6] = fmaf(pz, vx, part[6]); part[7] = fmaf(pz, vy, part[7]); part[8] = fmaf(pz, vz, part[8]);
        const double dvx = vx, dvy = vy, dvz = vz, dm = m, dw = a.w[q];
        const double wpx = dw * (double)px, wpy = dw * (double)py, wpz = dw * (double)pz;
        double *acc = L.acc;
        acc[0] += dm;
        acc[1] = fma(dm, dvx, acc[1]); acc[2] = fma(dm, dvy, acc[2]); acc[3] = fma(dm, dvz, acc[3]);
        acc[13] = fma(wpx, dvx, acc[13]); acc[14] = fma(wpx, dvy, acc[14]); acc[15] = fma(wpx, dvz, acc[15]);
        acc[16] = fma(wpy, dvx, acc[16]); acc[17] = fma(wpy, dvy, acc[17]); acc[18] = fma(wpy, dvz, acc[18]);
        acc[19] = fma(wpz, dvx, acc[19]); acc[20] = fma(wpz, dvy, acc[20]); acc[21] = fma(wpz, dvz, acc[21]);
        acc[22] = fma(dw, fma(dvx, dvx, fma(dvy, dvy, dvz * dvz)), acc[22]);
        if (!fc.wm) { acc[23] = fma(dw, dvx, acc[23]); acc[24] = fma(dw, dvy, acc[24]); acc[25] = fma(dw, dvz, acc[25]); }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) L.acc[4 + k] += (double)part[k];
}

template <bool NOREF = false, bool RMSD = false>
__device__ __forceinline__ void gr_finalize_frame_lite(const GrAccPartial *partials, uint32_t nchunks, uint32_t frame, const float *frames, size_t frame_stride,
                                                       uint32_t first_slot, const GrSel &sel, const GrBox *boxes, const GrPlanDev &plan, GrFrameState *state,
                                                       double *tot, float *ext, uint32_t lane);

// agent-scope (sc1, write-through) stores for records another workgroup of the same launch will read
template <typename T> __device__ __forceinline__ void gr_st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The closed-form single pass (MODE 0: RMSD without fit, and non-contiguous selections) and the literal multi-pass sums
// (MODE 1: q = wrap(x + shift) - box centre with the shift of the frame's state, the exact path).
template <int MODE>
__global__ __launch_bounds__(GR_WG) void k_rmsd_accum(
    const float *__restrict__ frames, size_t frame_stride, uint32_t first_slot,
    const float *__restrict__ masses, GrSel sel, const GrBox *__restrict__ boxes,
    GrPlanDev plan, const GrFrameState *state, GrAccPartial *partials) {
    __shared__ GrBox box;
    __shared__ double lds[(GR_WG / 64) * GR_ACC_K];
    __shared__ uint32_t ldsu[GR_WG / 64];
    __shared__ float ldsf[GR_WG / 64];
    const uint32_t frame = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const float *xyz = frames + (size_t)(first_slot + frame) * frame_stride;
    gr_stage_box(&box, boxes + first_slot + frame);
    GrLaneAcc L;
    L.reset();
    GrFrameConst fc;
    gr_pos_load(xyz, sel.contiguous ? sel.start : sel.idx[0], fc.gx, fc.gy, fc.gz);
    fc.sx = fc.sy = fc.sz = 0.f;
    if (MODE == 1) { fc.sx = state[frame].shift[0]; fc.sy = state[frame].shift[1]; fc.sz = state[frame].shift[2]; }
    fc.iax = box.iax; fc.iby = box.iby; fc.icz = box.icz;
    fc.rws2 = box.r_ws * box.r_ws;
    fc.tric = !box.ortho;
    fc.wm = plan.w_is_mass != 0;
    const bool wm = fc.wm;

    if (sel.contiguous) {
        // 4 atoms per lane per trip: 3 rows of positions, 3 rows of reference coordinates, 1 float4 of masses
        // (+1 of weights when they differ from the masses), all coalesced
        const uint32_t first = sel.start, last = sel.start + sel.n;
        const uint32_t g0 = sel.g0 << 6, g1 = (last + 3u) >> 2;      // float4 groups [g0, g1): g0 = first group of the first tile
        const float4 *f4 = reinterpret_cast<const float4 *>(xyz);
        const float4 *p4 = reinterpret_cast<const float4 *>(plan.p);
        const float4 *m4 = reinterpret_cast<const float4 *>(masses);
        const float4 *w4 = reinterpret_cast<const float4 *>(plan.w);
        // (a depth-one prefetch of the next trip's rows, as in k_sums_pk / k_center_sums, costs this kernel a wave per SIMD in
        // registers: measured 3.9 instead of 3.5 us per 1e6-atom frame)
        for (uint32_t g = g0 + chunk * GR_WG + threadIdx.x; g < g1; g += nchunks * GR_WG) {
            float4 r0, r1, r2, q0, q1, q2;
            gr_rows_load(f4, g, r0, r1, r2);
            const size_t pg = (size_t)(g - g0);
            gr_rows_load(p4, pg, q0, q1, q2);
            const float4 mm = m4[g];
            const float4 ww = wm ? mm : w4[pg];
            const uint32_t i = g << 2;
            GrA4 q;
            gr_rows_unpack(r0, r1, r2, q.x, q.y, q.z);
            gr_rows_unpack(q0, q1, q2, q.px, q.py, q.pz);
            q.m[0] = mm.x; q.m[1] = mm.y; q.m[2] = mm.z; q.m[3] = mm.w;
            q.w[0] = ww.x; q.w[1] = ww.y; q.w[2] = ww.z; q.w[3] = ww.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) { q.i[k] = i + k; q.ok[k] = (i + k >= first) && (i + k < last); }
            if (MODE == 0 && i >= first && i + 3 < last) gr_flush4<MODE>(L, q, false, box, fc); else gr_flush4<MODE>(L, q, true, box, fc);
        }
    } else {
        const uint32_t n4 = (sel.n + 3u) >> 2;
        for (uint32_t j4 = chunk * GR_WG + threadIdx.x; j4 < n4; j4 += nchunks * GR_WG) {
            GrA4 t;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t j = j4 * 4 + q;
                t.ok[q] = j < sel.n;
                const uint32_t jj = t.ok[q] ? j : 0u;
                const uint32_t i = sel.idx[jj];
                t.i[q] = i; gr_pos_load(xyz, i, t.x[q], t.y[q], t.z[q]); t.m[q] = masses[i];
                gr_pos_load(plan.p, jj, t.px[q], t.py[q], t.pz[q]);
                t.w[q] = wm ? t.m[q] : plan.w[jj];
            }
            if (MODE == 0 && j4 * 4 + 3 < sel.n) gr_flush4<MODE>(L, t, false, box, fc); else gr_flush4<MODE>(L, t, true, box, fc);
        }
    }
    L.close(wm);
    gr_block_sum<GR_ACC_K>(L.acc, lds);
    const uint32_t bad_pos = gr_block_min_u32(L.bad_pos, ldsu);
    const uint32_t bad_mass = gr_block_min_u32(L.bad_mass, ldsu);
    float rmn[3], rmx[3], rfmn[3], rfmx[3];
    for (int a = 0; a < 3; ++a) {
        rmn[a] = gr_block_min_f32(L.mn[a], ldsf); rmx[a] = -gr_block_min_f32(-L.mx[a], ldsf);
        rfmn[a] = gr_block_min_f32(L.fmn[a], ldsf); rfmx[a] = -gr_block_min_f32(-L.fmx[a], ldsf);
    }
    if (threadIdx.x == 0) {
        GrAccPartial &o = partials[(size_t)frame * nchunks + chunk];
#pragma unroll
        for (int k = 0; k < GR_ACC_K; ++k) o.s[k] = L.acc[k];
        for (int a = 0; a < 3; ++a) { o.vmin[a] = rmn[a]; o.vmax[a] = rmx[a]; o.fmin[a] = rfmn[a]; o.fmax[a] = rfmx[a]; }
        o.bad_pos = bad_pos; o.bad_mass = bad_mass;
    }
}

// 3x3 rotation helpers (Jacobi eigen-solver, Kabsch rotation, Newton polar iteration): gr_rotation.h

// One workgroup per frame: sum the partial records, then lane 0 closes the algebra:
//   cv = sum(m v)/sum(m)                     (COM relative to the provisional centre; MODE 1: v is already q)
//   H  = A - (sum p) cv^T,  Hw = B - (sum w p) cv^T,  sum w|q|^2 = sum w|v|^2 - 2 cv.sum(w v) + W |cv|^2
//   R  from H;  rmsd^2 = (sum w|p|^2 + sum w|q|^2 - 2 sum_ab R_ab Hw_ab) / W      (= rmsd.rs:592-599 expanded)
// MODE 0 also proves the single-pass images equal the reference's (see DESIGN.md "image proof"):
// every v_i must lie strictly inside the minimum-image cell about BOTH the reference's Bai-Breen centre estimate
// (wherever in its rigorously bounded region it is) and the COM; otherwise the frame is flagged GR_ST_FALLBACK
// and redone by the multi-pass path.
// Closing algebra of the single pass for one frame (one lane).  `st` receives centre / com / shift / R / rmsd / status.
// FAST (k_sums_pk<false, true>): the RMSD sums (acc[13..25]) come from f32 products in 8-atom chains, widened to fp64; M is the plan's
// own fp64 sum of the weights (they ARE the masses: the caller checked element by element), and the rounding left in rmsd^2 is estimated
// from the magnitudes of the sums -- a frame whose rmsd is not well above that estimate goes to the exact-product pass (GR_ST_REDO_EXACT).
template <int MODE, bool LITE = false, bool NOREF = false, bool FAST = false>
__device__ inline void gr_finalize_math(const double *acc, const float mn[3], const float mx[3], const float fmn[3], const float fmx[3],
                                        uint32_t bad_pos, uint32_t bad_mass, const GrBox &b, const GrPlanDev &plan,
                                        const double g[3], uint32_t n_sel, GrFrameState &st, float rd = 0.0f) {
    if (MODE == 0) {
        // get_com: positions of the whole group are checked before any mass (iterators.rs:1405-1422)
        if (bad_pos != GR_NOIDX) { st.status = 6; st.err_index = bad_pos; return; }
        if (bad_mass != GR_NOIDX) { st.status = 7; st.err_index = bad_mass; return; }
        // interior groups are not NaN-tested atom by atom: a missing position / mass shows up here as a NaN sum
        // and the frame goes to the multi-pass path, which names the atom
        bool poisoned = false;
        for (int k = 0; k < 26; ++k) if (acc[k] != acc[k]) poisoned = true;
        if (poisoned) { st.status = GR_ST_FALLBACK; return; }
    }
    const double M = FAST ? plan.sw : acc[0];
    double cv[3] = { 0, 0, 0 };
    if (MODE == 0) { cv[0] = acc[1] / M; cv[1] = acc[2] / M; cv[2] = acc[3] / M; }
    if (MODE == 0) {
        // Where can the reference's Bai-Breen centre estimate c' lie?  Per fractional axis it is the circular mean
        // of the f_i.  With mu = mean f, theta_i = 2 pi (f_i - mu) (sum theta_i = 0), |theta_i| <= Theta = 2 pi E
        // (E = fractional extent, must be < 1/2), T = sum theta_i^2:
        //   |sum sin theta_i| = |sum (sin theta_i - theta_i)| <= Theta T / 6,   sum cos theta_i >= n - T / 2
        //   => |c'_a - mu_a| <= atan((Theta T / 6) / (n - T / 2)) / (2 pi) <= (Theta T / 6) / (n - T / 2) / (2 pi) =: eps_a
        // so c' = mu + e with |e| <= sum_a eps_a |box_a|.
        const double TWO_PI = 6.283185307179586;
        const double nsel = (double)n_sel;
        double mu[3], eps[3];
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
            mu[a] = acc[26 + a] / nsel;
            const double E = (double)fmx[a] - (double)fmn[a];
            double T = TWO_PI * TWO_PI * (acc[29 + a] - nsel * mu[a] * mu[a]);
            if (T < 0) T = 0;
            T *= 1.0 + 1e-4; T += 1e-3;                       // f32 rounding of the two moments
            // |theta_i| = 2 pi |f_i - mu| <= 2 pi max(fmax - mu, mu - fmin): the larger distance of mu from the two ends of the extent, not the
            // whole extent (rounds 1-3 used the extent: twice the bound for a symmetric group, hence twice the radius eps and -- in
            // get_com / get_center below -- twice as many frames whose periodic copy had to be settled by the estimate pass)
            const double dev = fmax((double)fmx[a] - mu[a], mu[a] - (double)fmn[a]) + 1.0e-6;
            const double Theta = TWO_PI * (dev < E ? dev : E) * (1.0 + 1e-6);
            const double den = nsel - 0.5 * T;
            if (!(E < 0.4999) || !(den > 0.0)) { ok = false; eps[a] = 0; continue; }
            eps[a] = (Theta * T / 6.0) / den / TWO_PI;        // atan(x) <= x for x >= 0: a (slightly) larger, still rigorous radius
            if (NOREF && LITE) {
                // get_center / get_com carry the THIRD central moment too (k_sums_pk<true>): sin x = x - x^3 / 6 + r, |r| <= |x|^5 / 120, so
                //   |sum sin theta_i| <= (2 pi)^3 |M3| / 6 + Theta^3 T / 120,     M3 = sum (f_i - mu)^3
                // -- for a roughly symmetric group M3 ~ 0 and the radius shrinks from ~Theta T / 6 to ~Theta^3 T / 120 (a blob of 0.4
                // cell widths: 1.3e-2 -> 1e-3), and with it the share of frames whose periodic copy the estimate pass has to settle.
                // The moments are f32 lane sums: 1e-4 of every raw term is added as their rounding (generous by orders of magnitude).
                const double s1 = acc[26 + a], s2 = acc[29 + a], s3 = acc[13 + a], m = mu[a];
                const double m3 = s3 - 3.0 * m * s2 + 3.0 * m * m * s1 - nsel * m * m * m;
                const double m3e = 1.0e-4 * (fabs(s3) + 3.0 * fabs(m) * s2 + 3.0 * m * m * fabs(s1) + nsel * fabs(m * m * m)) + 1.0e-6;
                const double e3 = (TWO_PI * TWO_PI * TWO_PI * (fabs(m3) + m3e) / 6.0 + Theta * Theta * Theta * T / 120.0) / den / TWO_PI;
                if (e3 < eps[a]) eps[a] = e3;
            }
        }
        // centre candidate in Cartesian coordinates and the radius of the region c' is confined to
        const double ce[3] = { mu[0] * b.ax + mu[1] * b.bx + mu[2] * b.cx, mu[1] * b.by + mu[2] * b.cy, mu[2] * b.cz };
        const double la = b.ax, lb = sqrt((double)b.bx * b.bx + (double)b.by * b.by), lc = sqrt((double)b.cx * b.cx + (double)b.cy * b.cy + (double)b.cz * b.cz);
        const double slack_tric = eps[0] * la + eps[1] * lb + eps[2] * lc;
        const double slack_ortho[3] = { eps[0] * b.ax, eps[1] * b.by, eps[2] * b.cz };
        const double margin = 1.0e-3;   // nm
        const double *cen[2] = { ce, cv };
        for (int t = 0; t < 2; ++t) {
            if (b.ortho) {
                const double L[3] = { b.ax, b.by, b.cz };
                for (int a = 0; a < 3; ++a) {
                    const double far = fmax(fabs((double)mx[a] - cen[t][a]), fabs((double)mn[a] - cen[t][a])) + (t == 0 ? slack_ortho[a] : 0.0);
                    if (!(far < 0.5 * L[a] - margin)) ok = false;
                }
            } else {
                double r2 = 0;
                const double L[3] = { b.ax, b.by, b.cz };
                for (int a = 0; a < 3; ++a) {
                    const double far = fmax(fabs((double)mx[a] - cen[t][a]), fabs((double)mn[a] - cen[t][a]));
                    r2 += far * far;
                    // about the COM the reference's q is wrap(x + shift) - box centre (rmsd.rs:479-492), and the triclinic wrap
                    // maps into the BRICK 0..ax, 0..by, 0..cz about the box centre: a group that fits the Wigner-Seitz sphere
                    // but sticks out of the brick (cz / 2 < r_ws in a rhombic dodecahedron) is broken by that wrap -- the
                    // images chosen here are then not the path's q, and the frame must take the literal path
                    if (!NOREF && t == 1 && !(far < 0.5 * L[a] - margin)) ok = false;
                }
                if (!(sqrt(r2) + (t == 0 ? slack_tric : 0.0) < (double)b.r_ws - margin)) ok = false;
            }
        }
        if (!ok) { st.status = GR_ST_FALLBACK; return; }
        if (NOREF) {
            // get_center / get_com (iterators.rs:1237-1266,1404-1438) unwrap the group about c' = the Bai-Breen estimate, which
            // lies in the cell: the reference's copy of the group is the one whose mean fractional coordinate mu lies in
            // [0, 1) -- c' is within eps of it.  With mu closer than that to a cell face the copy depends on where exactly
            // c' fell: those frames take the two-pass path.
            const double gfc = g[2] / (double)b.cz, guy = g[1] - gfc * (double)b.cy, gfb = guy / (double)b.by,
                         gfa = (g[0] - gfb * (double)b.bx - gfc * (double)b.cx) / (double)b.ax;
            const double gf[3] = { gfa, gfb, gfc };
            double nl[3];
            bool face = false;
            for (int a = 0; a < 3; ++a) {
                const double m = gf[a] + mu[a], fl = floor(m), fr = m - fl, guard = eps[a] + 1.0e-4;
                if (!(fr > guard && fr < 1.0 - guard)) face = true;
                nl[a] = -fl;
            }
            if (face) {   // some copy of (centre, com): the masked estimate pass places it (k_center_finalize, GR_ST_AMBIG)
                for (int a = 0; a < 3; ++a) { st.center[a] = (float)(g[a] + ce[a]); st.com[a] = (float)(g[a] + cv[a]); }
                st.status = GR_ST_AMBIG;
                return;
            }
            const double lat[3] = { nl[0] * b.ax + nl[1] * b.bx + nl[2] * b.cx, nl[1] * b.by + nl[2] * b.cy, nl[2] * b.cz };
            for (int a = 0; a < 3; ++a) { st.center[a] = (float)(g[a] + ce[a] + lat[a]); st.com[a] = (float)(g[a] + cv[a] + lat[a]); }
            return;
        }
        st.center[0] = (float)(g[0] + ce[0]); st.center[1] = (float)(g[1] + ce[1]); st.center[2] = (float)(g[2] + ce[2]);
        const float com[3] = { (float)(g[0] + cv[0]), (float)(g[1] + cv[1]), (float)(g[2] + cv[2]) };
        st.com[0] = com[0]; st.com[1] = com[1]; st.com[2] = com[2];
        st.shift[0] = b.bcx - com[0]; st.shift[1] = b.bcy - com[1]; st.shift[2] = b.bcz - com[2];
    }
    double H[3][3], Hw[3][3];
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
            H[a][c] = acc[4 + 3 * a + c] - plan.sp[a] * cv[c];
            Hw[a][c] = acc[13 + 3 * a + c] - plan.swp[a] * cv[c];
        }
    const double swqq = acc[22] - 2.0 * (cv[0] * acc[23] + cv[1] * acc[24] + cv[2] * acc[25]) +
                        plan.sw * (cv[0] * cv[0] + cv[1] * cv[1] + cv[2] * cv[2]);
    double R[3][3];
    gr_best_rotation(H, R);
    if (!LITE) {
        double tr = 0;
        for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) tr += R[a][c] * Hw[a][c];
        double r2 = (plan.swpp + swqq - 2.0 * tr) / plan.sw;
        if (FAST) {
            // What the f32 chains leave in r2.  Every term w p_a v_b / w v_a^2 / w v_a carries ~2 roundings of relative size <= 2^-24
            // (product, one add of a chain whose partial is at most 8 terms long); they are independent from atom to atom, so the sums
            // pick up a random walk.  With S = (sum w|p|^2 + sum w|v|^2) / W (nm^2) bounding every such product,
            //     sigma(r2) := 6e-8 * S * sqrt(20 / n)
            // CALIBRATED against the exact-product pass (tools/rmsd_calibrate.py -> profiles/r04_rmsd_calibration.json: 2e4 .. 1e6 atoms,
            // two cells, rmsd 0.003 .. 0.35 nm, 2 300 frames with the guard off): the observed rms of r2_fast - r2_exact is 0.3 .. 0.65
            // sigma, the largest deviation 2.1 sigma (beyond rmsd ~0.3 nm the f32 quantisation of the result itself, 2e-8 in r2, is
            // larger than either).  A frame is kept when plan.fast_sigmas (default 6) sigma move its rmsd by less than 2.5e-6 nm:
            // d(rmsd) = d(r2) / (2 rmsd); at that limit the largest deviation seen so far would be 9e-7 nm.
            const double S = (plan.swpp + acc[22]) / plan.sw;
            // THE MODEL IS ONLY A MODEL (round 5): coordinates on a lattice -- a crystal, or simply every xtc frame's multiples of 0.001 nm
            // against a regular reference -- repeat the same values, and with them the same roundings, thousands of times: the chains' errors
            // then add up instead of walking (measured, tests/test_gpu_rmsd_fast.py: a simple-cubic lattice of 1e6 atoms against its displaced,
            // quantised copy: |fast - exact| = 4.7e-6 nm where the model said 1.3e-7).  So the pass MEASURES its own rounding: every lane sums
            // sum m |v|^2 twice, as one f32 chain of 8 terms and as two chains of 4 (a different association of the same terms), and each
            // chunk reports the difference of its two totals in units of what the model expects of it (`rd`, the largest chunk's).  Random
            // roundings give rd ~ 1 (at most ~3 over a frame's chunks); coherent ones gave 30-100: sigma is scaled by rd / 3 beyond that.
            const double sigma = 6.0e-8 * S * sqrt(20.0 / (double)n_sel) * ((double)rd > 3.0 ? (double)rd / 3.0 : 1.0);
            if (plan.fast_sigmas > 0.0f && (!(r2 > 0.0) || !((double)plan.fast_sigmas * sigma < 5.0e-6 * sqrt(r2)))) { st.status = GR_ST_REDO_EXACT; return; }
        }
        if (r2 < 0.0) r2 = 0.0;
        st.rmsd = (float)sqrt(r2);
    }   // LITE: the fit pass sums w |R q - p|^2 and k_rmsd_close writes the rmsd
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) st.R[3 * c + a] = (float)R[a][c];   // column-major
}

template <int MODE>
__global__ __launch_bounds__(GR_WG) void k_rmsd_finalize(
    const GrAccPartial *__restrict__ partials, uint32_t nchunks,
    const float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, GrSel sel,
    const GrBox *__restrict__ boxes, GrPlanDev plan, GrFrameState *__restrict__ state) {
    __shared__ double lds[(GR_WG / 64) * GR_ACC_K];
    __shared__ uint32_t ldsu[GR_WG / 64];
    __shared__ float ldsf[GR_WG / 64];
    const uint32_t frame = blockIdx.x;
    double acc[GR_ACC_K];
#pragma unroll
    for (int k = 0; k < GR_ACC_K; ++k) acc[k] = 0.0;
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    float fmn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, fmx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    uint32_t bad_pos = GR_NOIDX, bad_mass = GR_NOIDX;
    for (uint32_t c = threadIdx.x; c < nchunks; c += GR_WG) {
        const GrAccPartial &p = partials[(size_t)frame * nchunks + c];
#pragma unroll
        for (int k = 0; k < GR_ACC_K; ++k) acc[k] += p.s[k];
        for (int a = 0; a < 3; ++a) {
            mn[a] = gr_fminf(mn[a], p.vmin[a]); mx[a] = gr_fmaxf(mx[a], p.vmax[a]);
            fmn[a] = gr_fminf(fmn[a], p.fmin[a]); fmx[a] = gr_fmaxf(fmx[a], p.fmax[a]);
        }
        bad_pos = min(bad_pos, p.bad_pos); bad_mass = min(bad_mass, p.bad_mass);
    }
    gr_block_sum<GR_ACC_K>(acc, lds);
    bad_pos = gr_block_min_u32(bad_pos, ldsu);
    bad_mass = gr_block_min_u32(bad_mass, ldsu);
    for (int a = 0; a < 3; ++a) {
        mn[a] = gr_block_min_f32(mn[a], ldsf); mx[a] = -gr_block_min_f32(-mx[a], ldsf);
        fmn[a] = gr_block_min_f32(fmn[a], ldsf); fmx[a] = -gr_block_min_f32(-fmx[a], ldsf);
    }
    if (threadIdx.x != 0) return;
    GrFrameState &st = state[frame];
    if (st.status != 0) return;
    const float *xyz = frames + (size_t)(first_slot + frame) * frame_stride;
    float g0x, g0y, g0z;
    gr_pos_load(xyz, sel.contiguous ? sel.start : sel.idx[0], g0x, g0y, g0z);
    const double g[3] = { g0x, g0y, g0z };
    gr_finalize_math<MODE>(acc, mn, mx, fmn, fmx, bad_pos, bad_mass, boxes[first_slot + frame], plan, g, sel.n, st);
}

// The same for the two-pass sums records, ONE WAVE per frame and no barrier: lane c sums the records c, c + 64, ..., a
// reduce-scatter leaves the 19 totals / 12 extents spread over the lanes, LDS hands them to lane 0.
// (tot: 32 doubles, ext: 16 floats of LDS owned by the calling wave)
// RMSD: the records also carry the closed-form RMSD's sums (slots 13..25, k_sums_pk<false, true>): values 19..31 of the scatter
template <bool NOREF, bool RMSD>
__device__ __forceinline__ void gr_finalize_frame_lite(const GrAccPartial *partials, uint32_t nchunks, uint32_t frame, const float *frames, size_t frame_stride,
                                                       uint32_t first_slot, const GrSel &sel, const GrBox *boxes, const GrPlanDev &plan, GrFrameState *state,
                                                       double *tot, float *ext, uint32_t lane) {
    double s[32];
    float e[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) { s[k] = 0.0; e[k] = -3.0e38f; }
    uint32_t bad_pos = GR_NOIDX, bad_mass = GR_NOIDX;
    for (uint32_t c = lane; c < nchunks; c += 64) {
        const GrAccPartial &p = partials[(size_t)frame * nchunks + c];
#pragma unroll
        for (int k = 0; k < 13; ++k) s[k] += p.s[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) s[13 + k] += p.s[26 + k];
        if (RMSD) {
#pragma unroll
            for (int k = 0; k < 13; ++k) s[19 + k] += p.s[13 + k];
        } else if (NOREF) {
#pragma unroll
            for (int k = 0; k < 3; ++k) s[19 + k] += p.s[13 + k];      // third moments of the fractional coordinates
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            e[a] = gr_fmaxf(e[a], -p.vmin[a]); e[3 + a] = gr_fmaxf(e[3 + a], p.vmax[a]);
            e[6 + a] = gr_fmaxf(e[6 + a], -p.fmin[a]); e[9 + a] = gr_fmaxf(e[9 + a], p.fmax[a]);
        }
        if (RMSD) e[12] = gr_fmaxf(e[12], p.rd);
        bad_pos = min(bad_pos, p.bad_pos); bad_mass = min(bad_mass, p.bad_mass);
    }
    const double t = gr_wave_sum_scatter32_f64(s, lane);
    const float m = gr_wave_max_scatter16(e, lane);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { bad_pos = min(bad_pos, (uint32_t)__shfl_xor((int)bad_pos, off, 64)); bad_mass = min(bad_mass, (uint32_t)__shfl_xor((int)bad_mass, off, 64)); }
    if ((lane & 1u) == 0) tot[lane >> 1] = t;
    if ((lane & 3u) == 0) ext[lane >> 2] = m;
    gr_wave_sync();
    if (lane != 0) return;
    GrFrameState &st = state[frame];
    if (st.status != 0) return;
    double acc[GR_ACC_K];
    for (int k = 0; k < GR_ACC_K; ++k) acc[k] = 0.0;
    for (int k = 0; k < 13; ++k) acc[k] = tot[k];
    for (int k = 0; k < 6; ++k) acc[26 + k] = tot[13 + k];
    if (RMSD) for (int k = 0; k < 13; ++k) acc[13 + k] = tot[19 + k];
    else if (NOREF) for (int k = 0; k < 3; ++k) acc[13 + k] = tot[19 + k];
    const float mn[3] = { -ext[0], -ext[1], -ext[2] }, mx[3] = { ext[3], ext[4], ext[5] };
    const float fmn[3] = { -ext[6], -ext[7], -ext[8] }, fmx[3] = { ext[9], ext[10], ext[11] };
    const float *xyz = frames + (size_t)(first_slot + frame) * frame_stride;
    float g0x, g0y, g0z;
    gr_pos_load(xyz, sel.contiguous ? sel.start : sel.idx[0], g0x, g0y, g0z);
    const double g[3] = { g0x, g0y, g0z };
    gr_finalize_math<0, !RMSD, NOREF, RMSD>(acc, mn, mx, fmn, fmx, bad_pos, bad_mass, boxes[first_slot + frame], plan, g, sel.n, st, RMSD ? ext[12] : 0.0f);
}

template <bool NOREF = false, bool RMSD = false>
__global__ __launch_bounds__(64) void k_rmsd_finalize_lite(
    const GrAccPartial *__restrict__ partials, uint32_t nchunks,
    const float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, GrSel sel,
    const GrBox *__restrict__ boxes, GrPlanDev plan, GrFrameState *__restrict__ state) {
    __shared__ double tot[32];
    __shared__ float ext[16];
    gr_finalize_frame_lite<NOREF, RMSD>(partials, nchunks, blockIdx.x, frames, frame_stride, first_slot, sel, boxes, plan, state, tot, ext, threadIdx.x);
}

// rmsd = sqrt(sum of the fit pass's workgroup partials / sum w)  (rmsd.rs:599); one wave per frame, fixed summation order
__global__ __launch_bounds__(64) void k_rmsd_close(const double *__restrict__ fit_partials, uint32_t nparts, double sum_w, GrFrameState *__restrict__ state) {
    const uint32_t frame = blockIdx.x;
    if (state[frame].status != 0) return;
    // four independent chains per lane: the loads of a lane do not wait for one another
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double *p = fit_partials + (size_t)frame * nparts;
    uint32_t k = threadIdx.x;
    for (; k + 192 < nparts; k += 256) { s0 += p[k]; s1 += p[k + 64]; s2 += p[k + 128]; s3 += p[k + 192]; }
    for (; k < nparts; k += 64) s0 += p[k];
    const double s = gr_wave_sum((s0 + s1) + (s2 + s3));
    if (threadIdx.x == 0) state[frame].rmsd = (float)sqrt(fmax(s, 0.0) / sum_w);
}

// ------------------------------------------------------------------------------------------ plan extraction
// extract_data_from_system on the reference (rmsd.rs:425-446): p_i = wrap(x_i + shift) - box_centre,
// stored at ordinal j + pofs; plus the frame-independent sums [0..2] sum p, [3..5] sum w p, [6] sum w|p|^2, [7] sum w.
__global__ __launch_bounds__(GR_WG) void k_plan_extract(
    const float *__restrict__ xyz, const float *__restrict__ masses, GrSel sel, const GrBox *__restrict__ boxp,
    const GrFrameState *__restrict__ state, uint32_t pofs, float *__restrict__ p_out, float *__restrict__ w_out,
    GrCenPartial *__restrict__ partials, float *__restrict__ p_span = nullptr) {
    __shared__ GrBox box;
    __shared__ double lds[(GR_WG / 64) * GR_CEN_K];
    gr_stage_box(&box, boxp);
    const float sx = state->shift[0], sy = state->shift[1], sz = state->shift[2];
    double acc[GR_CEN_K];
#pragma unroll
    for (int k = 0; k < GR_CEN_K; ++k) acc[k] = 0.0;
    // once per plan: one atom per lane and trip, component loads / stores through the layout's index function
    for (uint32_t j = blockIdx.x * GR_WG + threadIdx.x; j < sel.n; j += gridDim.x * GR_WG) {
        const uint32_t i = sel.contiguous ? sel.start + j : sel.idx[j];
        float x, y, z;
        gr_pos_load(xyz, i, x, y, z);
        x += sx; y += sy; z += sz;
        gr_wrap(x, y, z, box);
        x -= box.bcx; y -= box.bcy; z -= box.bcz;
        const float w = masses[i];
        const size_t jj = (size_t)j + pofs;
        gr_pos_store(p_out, jj, x, y, z); w_out[jj] = w;
        if (p_span) gr_pos_store(p_span, (size_t)i - ((size_t)sel.g0 << 8), x, y, z);      // masked selections: a second copy laid out by ATOM (tiles line up with the frame's)
        const double dx = x, dy = y, dz = z, dw = w;
        acc[0] += dx; acc[1] += dy; acc[2] += dz;
        acc[3] += dw * dx; acc[4] += dw * dy; acc[5] += dw * dz;
        acc[6] += dw * (dx * dx + dy * dy + dz * dz); acc[7] += dw;
    }
    gr_block_sum<GR_CEN_K>(acc, lds);
    if (threadIdx.x == 0) {
        GrCenPartial &o = partials[blockIdx.x];
#pragma unroll
        for (int k = 0; k < GR_CEN_K; ++k) o.s[k] = acc[k];
        o.bad_pos = GR_NOIDX; o.bad_mass = GR_NOIDX;
    }
}

// ------------------------------------------------------------------------------------------ translate / wrap
// MutAtomIteratorWithBox::translate / wrap (iterators.rs:1520-1553; atom.rs:498-545): x <- wrap(x + v)
// One frame per blockIdx.y (frames `frame_stride` floats apart; boxes, states and the 4-word "first atom without position"
// records follow the frame index).  use_state_shift: 0 = plain translate, 1 = atoms_center (shift from the frame's state),
// 2 = plain translate of the frames whose state carries no earlier error (batched calls: frames without a box are skipped).
__global__ __launch_bounds__(GR_WG) void k_translate_wrap(
    float *__restrict__ xyz, size_t frame_stride, GrSel sel, const GrBox *__restrict__ boxp, const GrFrameState *__restrict__ state,
    int use_state_shift, int dim_mask, float tx, float ty, float tz, uint32_t *__restrict__ bad_out) {
    __shared__ GrBox box;
    xyz += (size_t)blockIdx.y * frame_stride; boxp += blockIdx.y; state += blockIdx.y; bad_out += 4 * blockIdx.y;
    gr_stage_box(&box, boxp);
    if (use_state_shift == 2 && state->status != 0) return;
    if (use_state_shift == 1) {   // atoms_center: shift = filter(box centre - estimated centre, dim) (utility.rs:116-119)
        if (state->status != 0) return;
        tx = (dim_mask & 1) ? box.bcx - state->center[0] : 0.0f;
        ty = (dim_mask & 2) ? box.bcy - state->center[1] : 0.0f;
        tz = (dim_mask & 4) ? box.bcz - state->center[2] : 0.0f;
    }
    uint32_t bad = GR_NOIDX;
    auto tf = [&](uint32_t i, float &x, float &y, float &z) {
        if (x != x) { bad = min(bad, i); return; }
        x += tx; y += ty; z += tz;
        gr_wrap(x, y, z, box);
    };
    if (sel.contiguous) {
        // read-modify-write of the selection's 4-atom groups: three coalesced row loads, three coalesced row stores per lane
        // (atoms of a ragged first / last group that lie outside the selection are written back unchanged)
        const uint32_t first = sel.start, last = sel.start + sel.n;
        const uint32_t g0 = first >> 2, g1 = (last + 3u) >> 2;
        float4 *f4 = reinterpret_cast<float4 *>(xyz);
        for (uint32_t g = g0 + blockIdx.x * GR_WG + threadIdx.x; g < g1; g += gridDim.x * GR_WG) {
            float4 r0, r1, r2;
            gr_rows_load<true>(f4, g, r0, r1, r2);
            float x[4], y[4], z[4];
            gr_rows_unpack(r0, r1, r2, x, y, z);
            const uint32_t i = g << 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (i + k >= first && i + k < last) tf(i + k, x[k], y[k], z[k]);
            gr_rows_pack(x, y, z, r0, r1, r2);
            gr_rows_store<true>(f4, g, r0, r1, r2);
        }
    } else if (sel.masked & 2u) {
        // a scattered selection whose span walk moves fewer bytes than its list (GrSel::masked bit 1, decided by the host from the number
        // of 4-atom groups the selection touches): the same read-modify-write of whole groups over the span, an atom's membership from
        // its bit, untouched groups skipped unread.  Measured at 1e6 atoms (us per frame, list -> span): nine atoms in ten 7.2 -> 4.1,
        // two blocks of a sixth 2.7 -> 1.5 (= one block of a third), two blocks of 45 % 6.7 -> 3.7
        const uint32_t first = sel.start, last = sel.start + sel.span;
        const uint32_t g0 = first >> 2, g1 = (last + 3u) >> 2;
        float4 *f4 = reinterpret_cast<float4 *>(xyz);
        for (uint32_t g = g0 + blockIdx.x * GR_WG + threadIdx.x; g < g1; g += gridDim.x * GR_WG) {
            const uint32_t nib = (sel.mask[g >> 3] >> ((g & 7u) * 4u)) & 15u;
            if (nib == 0u) continue;                       // (nothing of the group is selected: not even read)
            float4 r0, r1, r2;
            gr_rows_load<true>(f4, g, r0, r1, r2);
            float x[4], y[4], z[4];
            gr_rows_unpack(r0, r1, r2, x, y, z);
            const uint32_t i = g << 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) if ((nib >> k) & 1u) tf(i + k, x[k], y[k], z[k]);
            gr_rows_pack(x, y, z, r0, r1, r2);
            gr_rows_store<true>(f4, g, r0, r1, r2);
        }
    } else {
        for (uint32_t j = blockIdx.x * GR_WG + threadIdx.x; j < sel.n; j += gridDim.x * GR_WG) {
            const uint32_t i = sel.idx[j];
            float x, y, z;
            gr_pos_load(xyz, i, x, y, z);
            tf(i, x, y, z);
            gr_pos_store(xyz, i, x, y, z);
        }
    }
    // first atom without position: rare -- a wave that has one reduces and reports it, the others do nothing (at the default grid a workgroup
    // is ONE trip: a block-wide minimum through LDS and a barrier per trip cost more than the test they serve; round 5)
    if (__builtin_amdgcn_ballot_w64(bad != GR_NOIDX) != 0ull) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bad = min(bad, (uint32_t)__shfl_xor((int)bad, off, 64));
        if ((threadIdx.x & 63u) == 0u) atomicMin(bad_out, bad);
    }
}

// The same operation on an ORTHORHOMBIC cell as one 16-byte load and one 16-byte store per lane (round 5).  There every coordinate wraps on
// its own (x <- wrap(x + tx) along a, y along b, z along c: gr_wrap's three independent gr_wrap_value), so a lane needs no whole atom --
// and a copy whose waves are fresh, hold one float4 per lane and are handed out by the dispatcher in address order streams at 6.3-6.5 TB/s
// where the three-rows-per-lane walk above reaches 5.8 (profiles/r05_copy_matrix.md, "U = 3").  A workgroup is ONE WAVE and one row of
// a 256-atom tile (gr_layout.h: row 0 = x0 x1 y0 y1, row 1 = z0 z1 x2 x3, row 2 = y2 y3 z2 z3 of the lane's four atoms): three waves of
// one workgroup with a barrier between them -- tried first -- ran at the walk's speed.  An atom without position is NaN in x AND in y and z
// inside the library (k_tile), so every coordinate knows by itself that it must stay as it is.  Contiguous selections only.
__global__ __launch_bounds__(64) void k_translate_wrap_rows(
    float *__restrict__ xyz, size_t frame_stride, GrSel sel, const GrBox *__restrict__ boxp, const GrFrameState *__restrict__ state,
    int use_state_shift, int dim_mask, float tx, float ty, float tz, uint32_t *__restrict__ bad_out) {
    xyz += (size_t)blockIdx.y * frame_stride; boxp += blockIdx.y; state += blockIdx.y; bad_out += 4 * blockIdx.y;
    if (use_state_shift == 2 && state->status != 0) return;
    if (use_state_shift == 1) {   // atoms_center: shift = filter(box centre - estimated centre, dim) (utility.rs:116-119)
        if (state->status != 0) return;
        tx = (dim_mask & 1) ? boxp->bcx - state->center[0] : 0.0f;
        ty = (dim_mask & 2) ? boxp->bcy - state->center[1] : 0.0f;
        tz = (dim_mask & 4) ? boxp->bcz - state->center[2] : 0.0f;
    }
    const uint32_t row = blockIdx.x % 3u, lane = threadIdx.x;
    const uint32_t tile = (sel.start >> 8) + blockIdx.x / 3u, i0 = (tile * 64u + lane) << 2;
    const uint32_t first = sel.start, last = sel.start + sel.n;
    float4 *p = reinterpret_cast<float4 *>(xyz) + (size_t)tile * 192u + row * 64u + lane;
    const float4 v = gr_stream_load(p);
    // element e of the float4: which atom of the lane's four, which axis (wave-uniform per row)
    //   row 0: atoms 0 1 0 1, axes x x y y     row 1: atoms 0 1 2 3, axes z z x x     row 2: atoms 2 3 2 3, axes y y z z
    const uint32_t a0 = row == 2u ? 2u : 0u, a1 = row == 2u ? 3u : 1u, a2 = row == 0u ? 0u : 2u, a3 = row == 0u ? 1u : 3u;
    const float L01 = row == 0u ? boxp->ax : row == 1u ? boxp->cz : boxp->by, L23 = row == 0u ? boxp->by : row == 1u ? boxp->ax : boxp->cz;
    const float iL01 = row == 0u ? boxp->iax : row == 1u ? boxp->icz : boxp->iby, iL23 = row == 0u ? boxp->iby : row == 1u ? boxp->iax : boxp->icz;
    const float t01 = row == 0u ? tx : row == 1u ? tz : ty, t23 = row == 0u ? ty : row == 1u ? tx : tz;
    const bool in0 = i0 + a0 >= first && i0 + a0 < last, in1 = i0 + a1 >= first && i0 + a1 < last;
    const bool in2 = i0 + a2 >= first && i0 + a2 < last, in3 = i0 + a3 >= first && i0 + a3 < last;
    float4 o = v;
    if (in0 && v.x == v.x) o.x = gr_wrap_value(v.x + t01, L01, iL01);
    if (in1 && v.y == v.y) o.y = gr_wrap_value(v.y + t01, L01, iL01);
    if (in2 && v.z == v.z) o.z = gr_wrap_value(v.z + t23, L23, iL23);
    if (in3 && v.w == v.w) o.w = gr_wrap_value(v.w + t23, L23, iL23);
    gr_stream_store(p, o);
    // first atom without position of the frame: the x coordinates are the first two elements of row 0 and the last two of row 1
    // (rare: the reduction is behind a wave-wide test)
    uint32_t bad = GR_NOIDX;
    if (row == 0u) { if (in0 && v.x != v.x) bad = i0; if (in1 && v.y != v.y) bad = min(bad, i0 + 1u); }
    else if (row == 1u) { if (in2 && v.z != v.z) bad = i0 + 2u; if (in3 && v.w != v.w) bad = min(bad, i0 + 3u); }
    if (__builtin_amdgcn_ballot_w64(bad != GR_NOIDX) != 0ull) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bad = min(bad, (uint32_t)__shfl_xor((int)bad, off, 64));
        if (lane == 0) atomicMin(bad_out, bad);
    }
}

// ------------------------------------------------------------------------------------------ pair distances
// group_all_distances (analysis.rs:401-427): D[i][j] = distance(x_i, x_j, dim), row-major n1 x n2.
// Workgroup tile: GR_PD_TI rows x 1024 columns.  Each lane keeps 4 consecutive j atoms in registers,
// the i atoms of the tile sit in LDS (broadcast reads), and every row is written with one 16-byte
// store per lane = 1 KiB contiguous per wavefront instruction.  The kernel is bound by the HBM write
// of the matrix (4 B/pair); orthorhombic boxes need ~15 flop/pair.
#ifndef GR_PD_TI
#define GR_PD_TI 8      // 8 rows: 94.6 us per 1e4 x 1e4 orthorhombic call, 16: 98.0, 32: 108 (more, smaller workgroups keep more stores in flight)
#endif
#ifndef GR_PD_NT
#define GR_PD_NT 1
#endif
// One frame per blockIdx.z (frames `frame_stride` floats, matrices `out_stride` floats, boxes one GrBox, error words 4 apart).
// The frame's box (with its image table) is read through a uniform, read-only pointer: scalar loads, so the table entries
// are SGPR operands of the FMAs instead of per-lane LDS reads.
//
// XYZ distances (the matrix the reference's users ask for) are VALU-bound, not write-bound, on this chip: a store-only
// kernel of the same shape writes the 400 MB matrix of config 3 in 63 us (tools/microbench/write_bw.hip, 6.5 TB/s), the
// generic distance needs ~40 VALU slots per orthorhombic pair (~100 us) and ~80 per triclinic pair (~200 us).  So XYZ has
// its own inner loops, two columns at a time in packed-f32 registers (v_pk_add/mul/fma_f32):
//   orthorhombic, every atom of the tile within [-L/4, 5L/4] (checked once per workgroup): |d| <= 1.5 L, so the
//     reference's loops (vector3d.rs:575-592) run at most once and `|d| > L/2 ? d - copysign(L, d) : d` is the same single
//     f32 rounding -- bit-identical, 5 slots per axis instead of 11; anything else takes the generic closed form;
//   triclinic (our extension): brick reduction with k = rint(d/L) per axis (an ulp outside the brick is harmless, the
//     image table covers it), then gain = |t|^2 - 2|d.t| per table entry with two entries per v_min3.
// (gr_v2f and its helpers: defined with the centres, above)
__device__ __forceinline__ float gr_mi_near(float d, float L, float h) {
    const float s = d - __builtin_copysignf(L, d);
    return __builtin_fabsf(d) > h ? s : d;
}
__device__ __forceinline__ gr_v2f gr_mi_near2(gr_v2f d, float L, float h) {
    gr_v2f cs = { __builtin_copysignf(L, d.x), __builtin_copysignf(L, d.y) };
    const gr_v2f s = d - cs;
    gr_v2f r = { __builtin_fabsf(d.x) > h ? s.x : d.x, __builtin_fabsf(d.y) > h ? s.y : d.y };
    return r;
}
template <int DIM>   // Dimension (src/structures/dimension.rs:13-23): only the requested components are min-imaged (vector3d.rs:458-486)
__device__ __forceinline__ gr_v2f gr_pd_ortho_near2(const float4 t, gr_v2f jx, gr_v2f jy, gr_v2f jz, const GrBox &b, float hx, float hy, float hz) {
    gr_v2f mx = { 0.0f, 0.0f }, my = mx, mz = mx;
    if (DIM == 1 || DIM == 4 || DIM == 5 || DIM == 7) mx = gr_mi_near2(gr_v2(t.x) - jx, b.ax, hx);
    if (DIM == 2 || DIM == 4 || DIM == 6 || DIM == 7) my = gr_mi_near2(gr_v2(t.y) - jy, b.by, hy);
    if (DIM == 3 || DIM == 5 || DIM == 6 || DIM == 7) mz = gr_mi_near2(gr_v2(t.z) - jz, b.cz, hz);
    if (DIM == 1) return mx;
    if (DIM == 2) return my;
    if (DIM == 3) return mz;
    const gr_v2f r2 = mx * mx + my * my + mz * mz;   // contracted like gr_mag3's x*x + y*y + z*z
    gr_v2f r = { __builtin_amdgcn_sqrtf(r2.x), __builtin_amdgcn_sqrtf(r2.y) };
    return r;
}
template <int NC>
__device__ __forceinline__ gr_v2f gr_pd_tric2(const float4 t, gr_v2f jx, gr_v2f jy, gr_v2f jz, const GrBox &b) {
    gr_v2f dx = gr_v2(t.x) - jx, dy = gr_v2(t.y) - jy, dz = gr_v2(t.z) - jz;
    gr_v2f q = dz * gr_v2(b.icz);
    gr_v2f k = { -rintf(q.x), -rintf(q.y) };
    dx = gr_v2_fma(k, gr_v2(b.cx), dx); dy = gr_v2_fma(k, gr_v2(b.cy), dy); dz = gr_v2_fma(k, gr_v2(b.cz), dz);
    q = dy * gr_v2(b.iby); k.x = -rintf(q.x); k.y = -rintf(q.y);
    dx = gr_v2_fma(k, gr_v2(b.bx), dx); dy = gr_v2_fma(k, gr_v2(b.by), dy);
    q = dx * gr_v2(b.iax); k.x = -rintf(q.x); k.y = -rintf(q.y);
    dx = gr_v2_fma(k, gr_v2(b.ax), dx);
    const gr_v2f r2 = dx * dx + dy * dy + dz * dz;
    gr_v2f best = { 0.0f, 0.0f };
    static_assert(NC % 2 == 0, "image table is searched two entries at a time");
    if (b.cand_pairs) {            // entry m + 1 = entry m + a (or a pad): d.(t + a) = d.t + ax dx   (gr_box_setup)
        const gr_v2f pa = gr_v2(b.ax) * dx;
#pragma unroll
        for (int m = 0; m < NC; m += 2) {
            const gr_v2f d0 = gr_v2_fma(gr_v2(b.cand[m][0]), dx, gr_v2_fma(gr_v2(b.cand[m][1]), dy, gr_v2(b.cand[m][2]) * dz));
            const gr_v2f d1 = d0 + pa;
            best.x = gr_min3f(best.x, fmaf(-2.0f, __builtin_fabsf(d0.x), b.cand_t2[m]), fmaf(-2.0f, __builtin_fabsf(d1.x), b.cand_t2[m + 1]));
            best.y = gr_min3f(best.y, fmaf(-2.0f, __builtin_fabsf(d0.y), b.cand_t2[m]), fmaf(-2.0f, __builtin_fabsf(d1.y), b.cand_t2[m + 1]));
        }
    } else {
#pragma unroll
        for (int m = 0; m < NC; m += 2) {
            const gr_v2f d0 = gr_v2_fma(gr_v2(b.cand[m][0]), dx, gr_v2_fma(gr_v2(b.cand[m][1]), dy, gr_v2(b.cand[m][2]) * dz));
            const gr_v2f d1 = gr_v2_fma(gr_v2(b.cand[m + 1][0]), dx, gr_v2_fma(gr_v2(b.cand[m + 1][1]), dy, gr_v2(b.cand[m + 1][2]) * dz));
            best.x = gr_min3f(best.x, fmaf(-2.0f, __builtin_fabsf(d0.x), b.cand_t2[m]), fmaf(-2.0f, __builtin_fabsf(d1.x), b.cand_t2[m + 1]));
            best.y = gr_min3f(best.y, fmaf(-2.0f, __builtin_fabsf(d0.y), b.cand_t2[m]), fmaf(-2.0f, __builtin_fabsf(d1.y), b.cand_t2[m + 1]));
        }
    }
    const gr_v2f s = r2 + best;
    gr_v2f r = { __builtin_amdgcn_sqrtf(fmaxf(s.x, 0.0f)), __builtin_amdgcn_sqrtf(fmaxf(s.y, 0.0f)) };
    return r;
}

// The same search when the VECTOR is needed (1-D / 2-D Dimensions of a triclinic cell are components of the 3-D minimum image,
// vector3d.rs:458-486): all NC gains are kept, their minimum is found with the same v_min3 chain, and a reverse scan names the
// FIRST entry that attains it -- the entry the sequential search of gr_tric_refine (strictly smaller wins) ends on; the entry's
// vector comes from an LDS copy of the table (one 16-byte read per pair: a per-lane index cannot address the SGPR copy), and its
// sign from one more dot product.  ~60 lane-instructions per pair against 124-180 for the rolled per-pair search.
// `tab` = float4 {tx, ty, tz, 0} x NC in LDS, entry NC = zeros (no image wins).
template <int NC, int DIM>
__device__ __forceinline__ gr_v2f gr_pd_tric_vec2(const float4 t, gr_v2f jx, gr_v2f jy, gr_v2f jz, const GrBox &b, const float4 *tab) {
    gr_v2f dx = gr_v2(t.x) - jx, dy = gr_v2(t.y) - jy, dz = gr_v2(t.z) - jz;
    // brick reduction with k = rint(d / L) per axis (an ulp outside the brick is harmless: the image table covers it)
    gr_v2f q = dz * gr_v2(b.icz);
    gr_v2f k = { -rintf(q.x), -rintf(q.y) };
    dx = gr_v2_fma(k, gr_v2(b.cx), dx); dy = gr_v2_fma(k, gr_v2(b.cy), dy); dz = gr_v2_fma(k, gr_v2(b.cz), dz);
    q = dy * gr_v2(b.iby); k.x = -rintf(q.x); k.y = -rintf(q.y);
    dx = gr_v2_fma(k, gr_v2(b.bx), dx); dy = gr_v2_fma(k, gr_v2(b.by), dy);
    q = dx * gr_v2(b.iax); k.x = -rintf(q.x); k.y = -rintf(q.y);
    dx = gr_v2_fma(k, gr_v2(b.ax), dx);
    const gr_v2f r2 = dx * dx + dy * dy + dz * dz;
    gr_v2f g[NC];
    gr_v2f best = { 0.0f, 0.0f };
    auto gains = [&](auto PAIRS) {            // PAIRS: entry m + 1 = entry m + a (or a pad): d.(t + a) = d.t + ax dx   (gr_box_setup)
        const gr_v2f pa = gr_v2(b.ax) * dx;
#pragma unroll
        for (int m = 0; m < NC; m += 2) {
            const gr_v2f d0 = gr_v2_fma(gr_v2(b.cand[m][0]), dx, gr_v2_fma(gr_v2(b.cand[m][1]), dy, gr_v2(b.cand[m][2]) * dz));
            gr_v2f d1;
            if (decltype(PAIRS)::value) d1 = d0 + pa;
            else d1 = gr_v2_fma(gr_v2(b.cand[m + 1][0]), dx, gr_v2_fma(gr_v2(b.cand[m + 1][1]), dy, gr_v2(b.cand[m + 1][2]) * dz));
            g[m].x = fmaf(-2.0f, __builtin_fabsf(d0.x), b.cand_t2[m]); g[m].y = fmaf(-2.0f, __builtin_fabsf(d0.y), b.cand_t2[m]);
            g[m + 1].x = fmaf(-2.0f, __builtin_fabsf(d1.x), b.cand_t2[m + 1]); g[m + 1].y = fmaf(-2.0f, __builtin_fabsf(d1.y), b.cand_t2[m + 1]);
            best.x = gr_min3f(best.x, g[m].x, g[m + 1].x); best.y = gr_min3f(best.y, g[m].y, g[m + 1].y);
        }
    };
    if (b.cand_pairs) gains(std::true_type()); else gains(std::false_type());
    // (a vector shorter than r_ws is its own minimum image whatever the table says: gr_tric_refine returns before the search)
    const float rws2 = b.r_ws * b.r_ws;
    int ix = NC, iy = NC;
#pragma unroll
    for (int m = NC - 1; m >= 0; --m) { ix = g[m].x == best.x ? m : ix; iy = g[m].y == best.y ? m : iy; }
    ix = (best.x < 0.0f && !(r2.x < rws2)) ? ix : NC; iy = (best.y < 0.0f && !(r2.y < rws2)) ? iy : NC;
    const float4 tx = tab[ix], ty = tab[iy];
    // add t when d.t < 0, subtract it otherwise (gr_tric_refine)
    const float sx = __builtin_copysignf(1.0f, -fmaf(tx.x, dx.x, fmaf(tx.y, dy.x, tx.z * dz.x))), sy = __builtin_copysignf(1.0f, -fmaf(ty.x, dx.y, fmaf(ty.y, dy.y, ty.z * dz.y)));
    const float ax_ = dx.x + sx * tx.x, ay_ = dy.x + sx * tx.y, az_ = dz.x + sx * tx.z;
    const float bx_ = dx.y + sy * ty.x, by_ = dy.y + sy * ty.y, bz_ = dz.y + sy * ty.z;
    gr_v2f r;
    if (DIM == 1) { r.x = ax_; r.y = bx_; }
    else if (DIM == 2) { r.x = ay_; r.y = by_; }
    else if (DIM == 3) { r.x = az_; r.y = bz_; }
    else if (DIM == 4) { r.x = gr_mag3(ax_, ay_, 0.0f); r.y = gr_mag3(bx_, by_, 0.0f); }
    else if (DIM == 5) { r.x = gr_mag3(ax_, 0.0f, az_); r.y = gr_mag3(bx_, 0.0f, bz_); }
    else { r.x = gr_mag3(0.0f, ay_, az_); r.y = gr_mag3(0.0f, by_, bz_); }
    return r;
}

// FUSED REDUCERS (round 5; SURVEY 8 A8's extension, analysis.rs:401-427 with its consumers :1420-1451): the same tiles, the same distances, bit
// for bit -- but instead of 4 n1 n2 bytes of matrix only what the caller wants of it reaches memory: the smallest / largest distance, the number
// of pairs closer than a cut-off (each per row of the matrix or over all of it), or a histogram.  min / max / counts do not depend on the order
// of the pairs, so the results EQUAL the same reduction of the full matrix.  Floats travel as order-preserving unsigned keys (atomicMin / Max).
enum { GR_PDR_MIN = 1, GR_PDR_MAX = 2, GR_PDR_COUNT_BELOW = 3, GR_PDR_HIST = 4 };
#define GR_PDR_MAX_BINS 4096
#define GR_PDR_ROW_TILES 8u        /* row tiles (of GR_PD_TI rows) a reducing workgroup walks */
#define GR_PDR_SHARDS 256u         /* slots a frame's whole-matrix results are spread over (a power of two) */
struct GrPdRed { int op, per_row; float param /* cut-off | bins per nm */; uint32_t nbins; uint32_t *out; size_t out_stride; };
__host__ __device__ __forceinline__ uint32_t gr_f32_key(float f) { uint32_t b; memcpy(&b, &f, 4); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__host__ __device__ __forceinline__ float gr_key_f32(uint32_t k) { const uint32_t b = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu); float f; memcpy(&f, &b, 4); return f; }

template <int NC, bool RED = false>
__global__ __launch_bounds__(GR_WG) void k_pairdist(
    const float *__restrict__ xyz, size_t frame_stride, GrSel s1, GrSel s2, const GrBox *__restrict__ boxes, int dim,
    float *__restrict__ out, size_t out_stride, uint32_t *__restrict__ bad_out, GrPdRed red) {
    const GrBox &box = boxes[blockIdx.z];
    xyz += (size_t)blockIdx.z * frame_stride; out += (size_t)blockIdx.z * out_stride; bad_out += 4 * blockIdx.z;
    // RED: a workgroup walks GR_PDR_ROW_TILES row tiles with its 1024 columns (their atoms loaded once) and hands its results over once
    // per tile (per-row values) or once at the end (whole-matrix values into one of GR_PDR_SHARDS slots the host combines, the
    // histogram from LDS): one atomic per workgroup and row / bin instead of one per pair -- a single word takes ~90 atomics per us.
    constexpr uint32_t RT = RED ? GR_PDR_ROW_TILES : 1u;
    __shared__ uint32_t red_row[RED ? GR_PD_TI : 1], red_hist[RED ? GR_PDR_MAX_BINS : 1], red_blk[RED ? GR_WG / 64 : 1];
    uint32_t red_key = RED && red.op == GR_PDR_MIN ? 0xFFFFFFFFu : 0u, red_cnt = 0u;     // this lane's share of a whole-matrix reduction
    // per_row == 2, "transposed": the host swapped the groups, so the lane's four COLUMN atoms are the rows the caller asked about -- their
    // values are the lane's own (no exchange between lanes at all); the entry is distance(lane's atom, tile's atom), the mirror image of what
    // the loops below compute: the same bits in the fast paths, negated in the signed 1-D dimensions (as k_pairdist_sym), computed the
    // other way round in the generic one
    uint32_t red_col[4] = { red_key, red_key, red_key, red_key };
    const bool red_t = RED && red.per_row == 2, red_neg = red_t && dim >= 1 && dim <= 3;
    if (RED) {
        red.out += (size_t)blockIdx.z * red.out_stride;
        if (threadIdx.x < GR_PD_TI) red_row[threadIdx.x] = red.op == GR_PDR_MIN ? 0xFFFFFFFFu : 0u;
        if (red.op == GR_PDR_HIST) for (uint32_t b = threadIdx.x; b < red.nbins; b += GR_WG) red_hist[b] = 0u;
    }
    __shared__ float ti[GR_PD_TI][4];
    __shared__ uint32_t ldsu[GR_WG / 64];
    __shared__ float4 ttab[NC + 1];            // the image table for per-lane look-ups (1-D / 2-D dimensions of a triclinic cell)
    if (threadIdx.x <= NC) ttab[threadIdx.x] = threadIdx.x < NC ? make_float4(box.cand[threadIdx.x][0], box.cand[threadIdx.x][1], box.cand[threadIdx.x][2], 0.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t j0 = blockIdx.x * (GR_WG * 4) + threadIdx.x * 4;
    uint32_t bad = GR_NOIDX, badj = GR_NOIDX;   // first atom without position among the rows / the columns
    // every coordinate of the tile within a quarter box of the cell (NaN fails the test): licence for gr_mi_near
    const float lx = -0.25f * box.ax, ux = 1.25f * box.ax, ly = -0.25f * box.by, uy = 1.25f * box.by, lz = -0.25f * box.cz, uz = 1.25f * box.cz;
    int far_j = 0;
    float jx[4], jy[4], jz[4];
    if (s2.contiguous && j0 + 3 < s2.n && ((s2.start + j0) & 3u) == 0u) {
        // the lane's 4 atoms are one group of the slot: its three (coalesced) row loads instead of twelve 4-byte ones
        float4 r0, r1, r2;
        gr_rows_load(reinterpret_cast<const float4 *>(xyz), (size_t)((s2.start + j0) >> 2), r0, r1, r2);
        gr_rows_unpack(r0, r1, r2, jx, jy, jz);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (jx[k] != jx[k]) badj = min(badj, s2.start + j0 + k);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t j = j0 + k;
            jx[k] = jy[k] = jz[k] = 0.f;
            if (j < s2.n) {
                const uint32_t a = s2.contiguous ? s2.start + j : s2.idx[j];
                gr_pos_load(xyz, a, jx[k], jy[k], jz[k]);
                if (jx[k] != jx[k]) badj = min(badj, a);
            }
        }
    }
    if (box.ortho) {
#pragma unroll
        for (int k = 0; k < 4; ++k) far_j |= !(jx[k] >= lx && jx[k] <= ux && jy[k] >= ly && jy[k] <= uy && jz[k] >= lz && jz[k] <= uz);
    }
    for (uint32_t rt = 0; rt < RT; ++rt) {
    const uint32_t i0 = (blockIdx.y * RT + rt) * GR_PD_TI;
    if (RED && i0 >= s1.n) break;                          // (uniform)
    if (RED && rt) __syncthreads();                         // the tile before has been read (ti) and handed over (red_row)
    int far = far_j;
    if (threadIdx.x < GR_PD_TI) {
        const uint32_t i = i0 + threadIdx.x;
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < s1.n) {
            const uint32_t a = s1.contiguous ? s1.start + i : s1.idx[i];
            gr_pos_load(xyz, a, x, y, z);
            if (x != x) bad = min(bad, a);
            far |= !(x >= lx && x <= ux && y >= ly && y <= uy && z >= lz && z <= uz);
        }
        ti[threadIdx.x][0] = x; ti[threadIdx.x][1] = y; ti[threadIdx.x][2] = z; ti[threadIdx.x][3] = 0.f;
    }
    far = __syncthreads_or(far);   // (also publishes ti)
    const uint32_t ni = min((uint32_t)GR_PD_TI, s1.n > i0 ? s1.n - i0 : 0u);
    const bool vec_ok = ((s2.n & 3u) == 0u);   // rows stay 16-byte aligned
    auto put = [&](uint32_t r, const float (&d)[4], bool mirrored = false) {
        if (RED) {
            if (red_t) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = (red_neg && !mirrored) ? 0.0f - d[k] : d[k];
                    if (red.op == GR_PDR_MIN) red_col[k] = min(red_col[k], gr_f32_key(v));
                    else if (red.op == GR_PDR_MAX) red_col[k] = max(red_col[k], gr_f32_key(v));
                    else red_col[k] += v < red.param ? 1u : 0u;
                }
                return;
            }
            // the lane's four distances of row r -> the reduction (columns behind the group's end do not exist)
            uint32_t key = red.op == GR_PDR_MIN ? 0xFFFFFFFFu : 0u, cnt = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (j0 + k >= s2.n) continue;
                if (red.op == GR_PDR_MIN) key = min(key, gr_f32_key(d[k]));
                else if (red.op == GR_PDR_MAX) key = max(key, gr_f32_key(d[k]));
                else if (red.op == GR_PDR_COUNT_BELOW) cnt += d[k] < red.param ? 1u : 0u;
                else if (d[k] >= 0.0f) { const float fb = d[k] * red.param; if (fb < (float)red.nbins) atomicAdd(&red_hist[(uint32_t)fb], 1u); }
            }
            if (red.op == GR_PDR_HIST) return;
            if (!red.per_row) { red_key = red.op == GR_PDR_MIN ? min(red_key, key) : max(red_key, key); red_cnt += cnt; return; }
            // the wave's value of row r: four DPP steps and the two lane swaps (no LDS crossbar), one LDS atomic per wave
            uint32_t v = red.op == GR_PDR_COUNT_BELOW ? cnt : key;
            auto comb = [&](uint32_t a, uint32_t b) { return red.op == GR_PDR_MIN ? min(a, b) : red.op == GR_PDR_MAX ? max(a, b) : a + b; };
            v = comb(v, __float_as_uint(gr_xor_lane<1>(__uint_as_float(v)))); v = comb(v, __float_as_uint(gr_xor_lane<2>(__uint_as_float(v))));
            v = comb(v, __float_as_uint(gr_xor_lane<4>(__uint_as_float(v)))); v = comb(v, __float_as_uint(gr_xor_lane<8>(__uint_as_float(v))));
            const auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false); v = comb(r16[0], r16[1]);
            const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = comb(r32[0], r32[1]);
            if ((threadIdx.x & 63u) == 0u) {
                if (red.op == GR_PDR_MIN) atomicMin(&red_row[r], v); else if (red.op == GR_PDR_MAX) atomicMax(&red_row[r], v); else atomicAdd(&red_row[r], v);
            }
            return;
        }
        float *row = out + (size_t)(i0 + r) * s2.n;
        if (vec_ok && j0 + 3 < s2.n) {
#if GR_PD_NT
            gr_stream_store(reinterpret_cast<float4 *>(row + j0), make_float4(d[0], d[1], d[2], d[3]));   // written once, read by nobody here
#else
            *reinterpret_cast<float4 *>(row + j0) = make_float4(d[0], d[1], d[2], d[3]);
#endif
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (j0 + k < s2.n) row[j0 + k] = d[k];
        }
    };
    const gr_v2f jx01 = { jx[0], jx[1] }, jy01 = { jy[0], jy[1] }, jz01 = { jz[0], jz[1] };
    const gr_v2f jx23 = { jx[2], jx[3] }, jy23 = { jy[2], jy[3] }, jz23 = { jz[2], jz[3] };
    if (dim != 0 && box.ortho && !far) {
        const float hx = box.ax / 2.0f, hy = box.by / 2.0f, hz = box.cz / 2.0f;
        auto rows = [&](auto D) {
            for (uint32_t r = 0; r < ni; ++r) {
                const float4 t = *reinterpret_cast<const float4 *>(&ti[r][0]);
                const gr_v2f a = gr_pd_ortho_near2<decltype(D)::value>(t, jx01, jy01, jz01, box, hx, hy, hz), b = gr_pd_ortho_near2<decltype(D)::value>(t, jx23, jy23, jz23, box, hx, hy, hz);
                const float d[4] = { a.x, a.y, b.x, b.y };
                put(r, d);
            }
        };
        switch (dim) {
        case 1: rows(std::integral_constant<int, 1>()); break;
        case 2: rows(std::integral_constant<int, 2>()); break;
        case 3: rows(std::integral_constant<int, 3>()); break;
        case 4: rows(std::integral_constant<int, 4>()); break;
        case 5: rows(std::integral_constant<int, 5>()); break;
        case 6: rows(std::integral_constant<int, 6>()); break;
        default: rows(std::integral_constant<int, 7>()); break;
        }
    } else if (dim == 7 && !box.ortho) {
        for (uint32_t r = 0; r < ni; ++r) {
            const float4 t = *reinterpret_cast<const float4 *>(&ti[r][0]);
            const gr_v2f a = gr_pd_tric2<NC>(t, jx01, jy01, jz01, box), b = gr_pd_tric2<NC>(t, jx23, jy23, jz23, box);
            const float d[4] = { a.x, a.y, b.x, b.y };
            put(r, d);
        }
    } else if (dim != 0 && !box.ortho) {
        // 1-D / 2-D dimensions of a triclinic cell: components of the 3-D minimum-image vector
        auto rows = [&](auto D) {
            for (uint32_t r = 0; r < ni; ++r) {
                const float4 t = *reinterpret_cast<const float4 *>(&ti[r][0]);
                const gr_v2f a = gr_pd_tric_vec2<NC, decltype(D)::value>(t, jx01, jy01, jz01, box, ttab), b = gr_pd_tric_vec2<NC, decltype(D)::value>(t, jx23, jy23, jz23, box, ttab);
                const float d[4] = { a.x, a.y, b.x, b.y };
                put(r, d);
            }
        };
        switch (dim) {
        case 1: rows(std::integral_constant<int, 1>()); break;
        case 2: rows(std::integral_constant<int, 2>()); break;
        case 3: rows(std::integral_constant<int, 3>()); break;
        case 4: rows(std::integral_constant<int, 4>()); break;
        case 5: rows(std::integral_constant<int, 5>()); break;
        default: rows(std::integral_constant<int, 6>()); break;
        }
    } else {
        for (uint32_t r = 0; r < ni; ++r) {
            const float4 t = *reinterpret_cast<const float4 *>(&ti[r][0]);
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = red_t ? gr_distance<NC>(jx[k], jy[k], jz[k], t.x, t.y, t.z, dim, box) : gr_distance<NC>(t.x, t.y, t.z, jx[k], jy[k], jz[k], dim, box);
            put(r, d, true);
        }
    }
    if (RED && red.per_row == 1 && red.op != GR_PDR_HIST) {
        // the tile's rows -> the frame's: one atomic per row and workgroup; the slots start the next tile empty
        __syncthreads();
        if (threadIdx.x < ni) {
            uint32_t *o = red.out + i0 + threadIdx.x;
            if (red.op == GR_PDR_MIN) atomicMin(o, red_row[threadIdx.x]); else if (red.op == GR_PDR_MAX) atomicMax(o, red_row[threadIdx.x]); else atomicAdd(o, red_row[threadIdx.x]);
            red_row[threadIdx.x] = red.op == GR_PDR_MIN ? 0xFFFFFFFFu : 0u;
        }
    }
    }   // row tiles
    if (RED) {
        __syncthreads();
        if (red.op == GR_PDR_HIST) {
            for (uint32_t b = threadIdx.x; b < red.nbins; b += GR_WG) if (red_hist[b]) atomicAdd(red.out + b, red_hist[b]);
        } else if (red_t) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (j0 + k >= s2.n) continue;
                uint32_t *o = red.out + j0 + k;
                if (red.op == GR_PDR_MIN) atomicMin(o, red_col[k]); else if (red.op == GR_PDR_MAX) atomicMax(o, red_col[k]); else atomicAdd(o, red_col[k]);
            }
        } else if (!red.per_row) {
            auto comb = [&](uint32_t a, uint32_t b) { return red.op == GR_PDR_MIN ? min(a, b) : red.op == GR_PDR_MAX ? max(a, b) : a + b; };
            uint32_t v = red.op == GR_PDR_COUNT_BELOW ? red_cnt : red_key;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = comb(v, (uint32_t)__shfl_xor((int)v, off, 64));
            if ((threadIdx.x & 63u) == 0u) red_blk[threadIdx.x >> 6] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                v = red_blk[0];
                for (int w = 1; w < GR_WG / 64; ++w) v = comb(v, red_blk[w]);
                uint32_t *o = red.out + ((blockIdx.y * gridDim.x + blockIdx.x) & (GR_PDR_SHARDS - 1u));     // one of the frame's slots: the host combines them
                if (red.op == GR_PDR_MIN) atomicMin(o, v); else if (red.op == GR_PDR_MAX) atomicMax(o, v); else atomicAdd(o, v);
            }
        }
    }
    bad = gr_block_min_u32(bad, ldsu);
    badj = gr_block_min_u32(badj, ldsu);
    if (threadIdx.x == 0 && bad != GR_NOIDX) atomicMin(bad_out, bad);
    if (threadIdx.x == 0 && badj != GR_NOIDX) atomicMin(bad_out + 1, badj);
}

// The matrix of a selection WITH ITSELF (BASELINE configs[2]: "pair distances for a 10k-atom selection") is symmetric, and on
// this chip the XYZ matrix of a triclinic cell is bound by the VALU, not by its 400 MB of stores: only the tiles on and above the
// diagonal are computed, and every tile above it is written twice -- as it stands, and transposed through LDS so that the mirror
// image is written in rows as well (256-byte runs).  distance(x_i, x_j) and distance(x_j, x_i) are the same bits in the fast paths
// used here (d -> -d commutes with rint, fma and the |d.t| gains; the 1-D dimensions are signed: the mirror is the negation); the
// generic path (Dimension::None) makes no such promise and computes its mirror explicitly.  Only non-orthogonal cells come here:
// the orthorhombic loops are bound by the stores, and there the mirror image's shorter runs cost more than the arithmetic saves.
// Tile: 64 x 64 (128 x 128 measured 7 % slower on single calls: 3 160 long workgroups leave a tail), 256 lanes = 16 row groups x
// 16 column groups; a lane holds 4 consecutive column atoms and walks 4 rows.
#ifndef GR_PDS_T
#define GR_PDS_T 64
#endif
template <int NC, int MODE, int DIM>   // MODE 1: triclinic XYZ; 2: triclinic 1-D / 2-D; 3: generic (Dimension::None)
__device__ __forceinline__ void gr_pds_quad(const float4 t, const float (&jx)[4], const float (&jy)[4], const float (&jz)[4], const GrBox &box,
                                            const float4 *ttab, int dim, float (&d)[4], float (&m)[4]) {
    const gr_v2f jx01 = { jx[0], jx[1] }, jy01 = { jy[0], jy[1] }, jz01 = { jz[0], jz[1] };
    const gr_v2f jx23 = { jx[2], jx[3] }, jy23 = { jy[2], jy[3] }, jz23 = { jz[2], jz[3] };
    if (MODE == 3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { d[k] = gr_distance<NC>(t.x, t.y, t.z, jx[k], jy[k], jz[k], dim, box); m[k] = gr_distance<NC>(jx[k], jy[k], jz[k], t.x, t.y, t.z, dim, box); }
        return;
    }
    gr_v2f a, b;
    if (MODE == 1) { a = gr_pd_tric2<NC>(t, jx01, jy01, jz01, box); b = gr_pd_tric2<NC>(t, jx23, jy23, jz23, box); }
    else { a = gr_pd_tric_vec2<NC, DIM>(t, jx01, jy01, jz01, box, ttab); b = gr_pd_tric_vec2<NC, DIM>(t, jx23, jy23, jz23, box, ttab); }
    d[0] = a.x; d[1] = a.y; d[2] = b.x; d[3] = b.y;
    constexpr bool SIGNED = DIM >= 1 && DIM <= 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = SIGNED ? 0.0f - d[k] : d[k];   // (0 - d: coincident coordinates give +0 in both directions, never -0)
}

template <int NC>
__global__ __launch_bounds__(GR_WG) void k_pairdist_sym(
    const float *__restrict__ xyz, size_t frame_stride, GrSel s, const GrBox *__restrict__ boxes, int dim,
    float *__restrict__ out, size_t out_stride, uint32_t *__restrict__ bad_out) {
    // T x T tile: T / 4 column groups (a lane holds 4 column atoms) x 1024 / T row groups; the mirror image is staged H = 64 rows
    // at a time; a lane walks T / H phases x G groups of 4 rows
    // The mirror image in LDS is XOR-SWIZZLED by quads: element (column j, row quad q) lives at tt[j][4 (q ^ (j >> 2 & 15))].  A wave
    // writes one quad of 64 different columns (16 column groups x 4 row groups): with a plain or padded row stride (round 3: H + 4)
    // the 16-byte writes of lanes that share a row group land on two alternating sets of four banks -- 87.7 M bank-conflict cycles
    // per config-3 launch (profiles/r03_pmc_pairdist_sym.txt) -- with the swizzle the eight lanes of a pass cover all 32 banks,
    // and the row-wise reads (16 lanes x 16 bytes of one row) stay a permutation of one contiguous 256-byte row.
    constexpr uint32_t T = GR_PDS_T, CG = T / 4, RG = GR_WG / CG, H = 64, PH = T / H, G = H / RG / 4, LD = H;
    static_assert(GR_WG == 256 && (T == 128 || T == 64), "tile shape");
    const uint32_t bi = blockIdx.y, bj = blockIdx.x;
    if (bi > bj) return;                                   // below the diagonal: written by the tile above it
    const GrBox &box = boxes[blockIdx.z];
    xyz += (size_t)blockIdx.z * frame_stride; out += (size_t)blockIdx.z * out_stride; bad_out += 4 * blockIdx.z;
    __shared__ float ti[T][4];
    __shared__ float tt[T][LD];                            // the mirror image of half a tile: [column][row]
    __shared__ uint32_t ldsu[GR_WG / 64];
    __shared__ float4 ttab[NC + 1];
    const uint32_t tid = threadIdx.x, cg = tid % CG, rg = tid / CG, n = s.n;
    if (tid <= NC) ttab[tid] = tid < NC ? make_float4(box.cand[tid][0], box.cand[tid][1], box.cand[tid][2], 0.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t i0 = bi * T, j0 = bj * T + cg * 4;
    uint32_t bad = GR_NOIDX, badj = GR_NOIDX;
    if (tid < T) {
        const uint32_t i = i0 + tid;
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < n) {
            const uint32_t a = s.contiguous ? s.start + i : s.idx[i];
            gr_pos_load(xyz, a, x, y, z);
            if (x != x) bad = min(bad, a);
        }
        ti[tid][0] = x; ti[tid][1] = y; ti[tid][2] = z; ti[tid][3] = 0.f;
    }
    float jx[4], jy[4], jz[4];
    if (s.contiguous && j0 + 3 < n && ((s.start + j0) & 3u) == 0u) {
        float4 r0, r1, r2;
        gr_rows_load(reinterpret_cast<const float4 *>(xyz), (size_t)((s.start + j0) >> 2), r0, r1, r2);
        gr_rows_unpack(r0, r1, r2, jx, jy, jz);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (jx[k] != jx[k]) badj = min(badj, s.start + j0 + k);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t j = j0 + k;
            jx[k] = jy[k] = jz[k] = 0.f;
            if (j < n) {
                const uint32_t a = s.contiguous ? s.start + j : s.idx[j];
                gr_pos_load(xyz, a, jx[k], jy[k], jz[k]);
                if (jx[k] != jx[k]) badj = min(badj, a);
            }
        }
    }
    __syncthreads();               // (publishes ti and ttab)
    const bool vec_ok = (n & 3u) == 0u, diag = bi == bj;
    auto run = [&](auto M, auto D) {
        constexpr int MODE = decltype(M)::value, DIM = decltype(D)::value;
        for (uint32_t h = 0; h < PH; ++h) {
#pragma unroll 1
            for (uint32_t g = 0; g < G; ++g) {
                const uint32_t rl = 4u * G * rg + 4u * g;      // row inside this phase: 0 .. 63
                float m[4][4];
#pragma unroll
                for (uint32_t r = 0; r < 4; ++r) {
                    const uint32_t row = H * h + rl + r;
                    const float4 t = *reinterpret_cast<const float4 *>(&ti[row][0]);
                    float d[4];
                    gr_pds_quad<NC, MODE, DIM>(t, jx, jy, jz, box, ttab, dim, d, m[r]);
                    if (i0 + row < n) {
                        float *dst = out + (size_t)(i0 + row) * n + j0;
                        if (vec_ok && j0 + 3 < n) gr_stream_store(reinterpret_cast<float4 *>(dst), make_float4(d[0], d[1], d[2], d[3]));
                        else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) if (j0 + k < n) dst[k] = d[k];
                        }
                    }
                }
                if (!diag) {
#pragma unroll
                    for (uint32_t c = 0; c < 4; ++c) *reinterpret_cast<float4 *>(&tt[4u * cg + c][4u * ((rl >> 2) ^ (cg & 15u))]) = make_float4(m[0][c], m[1][c], m[2][c], m[3][c]);
                }
            }
            __syncthreads();
            if (!diag) {
#pragma unroll 2
                for (uint32_t p = 0; p < T / 16u; ++p) {
                    const uint32_t jr = p * 16u + (tid >> 4), seg = tid & 15u;
                    const uint32_t jg = bj * T + jr, ig = i0 + H * h + 4u * seg;
                    if (jg < n) {
                        const float4 v = *reinterpret_cast<const float4 *>(&tt[jr][4u * (seg ^ ((jr >> 2) & 15u))]);
                        float *dst = out + (size_t)jg * n + ig;
                        if (vec_ok && ig + 3 < n) gr_stream_store(reinterpret_cast<float4 *>(dst), v);
                        else { if (ig < n) dst[0] = v.x; if (ig + 1 < n) dst[1] = v.y; if (ig + 2 < n) dst[2] = v.z; if (ig + 3 < n) dst[3] = v.w; }
                    }
                }
            }
            __syncthreads();
        }
    };
    // (the host sends only non-orthogonal cells here: the orthorhombic loops are bound by the stores, not by the arithmetic)
    if (dim == 7) run(std::integral_constant<int, 1>(), std::integral_constant<int, 7>());
    else if (dim != 0) {
        switch (dim) {
        case 1: run(std::integral_constant<int, 2>(), std::integral_constant<int, 1>()); break;
        case 2: run(std::integral_constant<int, 2>(), std::integral_constant<int, 2>()); break;
        case 3: run(std::integral_constant<int, 2>(), std::integral_constant<int, 3>()); break;
        case 4: run(std::integral_constant<int, 2>(), std::integral_constant<int, 4>()); break;
        case 5: run(std::integral_constant<int, 2>(), std::integral_constant<int, 5>()); break;
        default: run(std::integral_constant<int, 2>(), std::integral_constant<int, 6>()); break;
        }
    } else run(std::integral_constant<int, 3>(), std::integral_constant<int, 0>());
    bad = gr_block_min_u32(bad, ldsu);
    badj = gr_block_min_u32(badj, ldsu);
    if (tid == 0 && bad != GR_NOIDX) atomicMin(bad_out, bad);
    if (tid == 0 && badj != GR_NOIDX) atomicMin(bad_out + 1, badj);
}

// ------------------------------------------------------------------------------------------ synthetic frames
__device__ __forceinline__ uint64_t gr_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float gr_u01(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) {
    const uint64_t h = gr_mix64(gr_mix64(gr_mix64(seed ^ (a * 0xD1342543DE82EF95ull)) ^ (b * 0xA0761D6478BD642Full)) ^ (c * 0xE7037ED1A0B428DBull));
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}

// points uniform in a ball of `radius` about the box centre (rejection from the cube, 16 tries)
__global__ void k_synth_reference(float *xyz, uint32_t n, const GrBox *boxp, float radius, uint64_t seed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GrBox &b = *boxp;
    float x = 0.f, y = 0.f, z = 0.f;
    for (uint32_t t = 0; t < 16; ++t) {
        x = 2.0f * gr_u01(seed, i, 3 * t, 1) - 1.0f; y = 2.0f * gr_u01(seed, i, 3 * t + 1, 1) - 1.0f; z = 2.0f * gr_u01(seed, i, 3 * t + 2, 1) - 1.0f;
        if (x * x + y * y + z * z <= 1.0f) break;
        if (t == 15) { x *= 0.5f; y *= 0.5f; z *= 0.5f; }
    }
    gr_pos_store(xyz, i, b.bcx + radius * x, b.bcy + radius * y, b.bcz + radius * z);
}

// frame f = R_f (x0 - c) + c + t_f + noise, wrapped into the cell
__global__ void k_synth_frames(const float *ref, float *frames, size_t frame_stride, uint32_t first_slot, uint32_t n,
                               const GrBox *boxp, uint64_t first_frame_index, uint64_t frame_index_stride, float sigma, uint64_t seed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t f = blockIdx.y;
    if (i >= n) return;
    const GrBox &b = *boxp;
    const uint64_t fi = first_frame_index + (uint64_t)f * frame_index_stride;
    // random unit quaternion -> rotation
    float q0 = 2.f * gr_u01(seed, fi, 0, 2) - 1.f, q1 = 2.f * gr_u01(seed, fi, 1, 2) - 1.f, q2 = 2.f * gr_u01(seed, fi, 2, 2) - 1.f, q3 = 2.f * gr_u01(seed, fi, 3, 2) - 1.f;
    float qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    if (qn < 1e-3f) { q0 = 1.f; q1 = q2 = q3 = 0.f; qn = 1.f; }
    q0 /= qn; q1 /= qn; q2 /= qn; q3 /= qn;
    const float R[3][3] = {
        { 1 - 2 * (q2 * q2 + q3 * q3), 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2) },
        { 2 * (q1 * q2 + q0 * q3), 1 - 2 * (q1 * q1 + q3 * q3), 2 * (q2 * q3 - q0 * q1) },
        { 2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 1 - 2 * (q1 * q1 + q2 * q2) } };
    // translation anywhere in the cell (fractional)
    const float fa = gr_u01(seed, fi, 4, 2), fb = gr_u01(seed, fi, 5, 2), fc = gr_u01(seed, fi, 6, 2);
    const float tx = fa * b.ax + fb * b.bx + fc * b.cx, ty = fb * b.by + fc * b.cy, tz = fc * b.cz;
    float x0, y0, z0;
    gr_pos_load(ref, i, x0, y0, z0);
    x0 -= b.bcx; y0 -= b.bcy; z0 -= b.bcz;
    // noise: sum of 4 uniforms (variance 1/3) scaled to sigma
    float nz[3];
    for (int a = 0; a < 3; ++a) {
        float s = gr_u01(seed, fi, 16 + 4 * a, 3 + i) + gr_u01(seed, fi, 17 + 4 * a, 3 + i) + gr_u01(seed, fi, 18 + 4 * a, 3 + i) + gr_u01(seed, fi, 19 + 4 * a, 3 + i) - 2.0f;
        nz[a] = s * 1.7320508f * sigma;
    }
    float x = R[0][0] * x0 + R[0][1] * y0 + R[0][2] * z0 + b.bcx + tx + nz[0];
    float y = R[1][0] * x0 + R[1][1] * y0 + R[1][2] * z0 + b.bcy + ty + nz[1];
    float z = R[2][0] * x0 + R[2][1] * y0 + R[2][2] * z0 + b.bcz + tz + nz[2];
    gr_wrap(x, y, z, b);
    gr_pos_store(frames + (size_t)(first_slot + f) * frame_stride, i, x, y, z);

}

__global__ void k_synth_uniform(float *xyz, uint32_t n, const GrBox *boxp, uint64_t seed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GrBox &b = *boxp;
    const float fa = gr_u01(seed, i, 0, 7), fb = gr_u01(seed, i, 1, 7), fc = gr_u01(seed, i, 2, 7);
    gr_pos_store(xyz, i, fa * b.ax + fb * b.bx + fc * b.cx, fb * b.by + fc * b.cy, fc * b.cz);
}
