// gr_hot.h -- the two kernels of the RMSD-fit hot path (contiguous selection):
//
//   k_sums_pk<NOREF>   pass 1 over the group: sum m, sum m v, A = sum p v^T (rmsd.rs:567-570), fractional moments + extents for
//                      the image proof; the workgroup that completes a frame closes it (rotation, shift)
//   k_fit_pk<RMSD>     pass 2 over all atoms: z = R (wrap(x + shift) - box centre) + reference COM and, for the group,
//                      sum w |R q - p|^2 (rmsd.rs:508-528,592-599)
//
// Round 1 measured both passes co-limited by VALU issue (65 and 94 VALU instructions per atom, rocprofv3 SQ_INSTS_VALU) and
// by the access pattern of the packed xyz records.  Here
//   * slots and plans are pair-tiled (gr_layout.h): a lane's 4 atoms arrive as three fully coalesced 16-byte loads -- one
//     element per lane and instruction, the shape that streams fastest on this chip -- and every register pair a lane receives
//     is (coordinate of atom a, same coordinate of atom a + 1): no LDS, no transposes, no register shuffles;
//   * every add / multiply / fma of the per-atom arithmetic is one v_pk_*_f32 for two atoms (gfx950 issues packed f32 at
//     full rate); only rint / floor and the min / max chains stay scalar, and those take two atoms per v_min3 / v_max3;
//   * the box is read through a uniform pointer: scalar loads, every box constant is an SGPR operand (no LDS staging, no barrier);
//   * the fit's wrap is the one-turn closed form k = floor(t / L) per axis (c, then b, then a) with ONE wave-wide check that
//     every result lies in (0, L]; a wave with an atom on a face, more than one cell away, or without position redoes its
//     atoms with the general wrap (gr_wrap: bit-identical to the reference's loops) -- ~11 instead of ~66 instructions per atom;
//   * interior trips carry no per-atom validity tests (a NaN poisons the sums, the closing step sends the frame to the
//     multi-pass path, which names the atom); the two ragged ends of the selection are masked instead of branched on.
// Reference arithmetic kept: rmsd.rs:562-599 (unweighted covariance, mass-weighted final sum), vector3d.rs:380-417 (wrap).
#pragma once
#include "gr_kernels.h"

__device__ __forceinline__ gr_v2f gr_v2p(float lo, float hi) { gr_v2f r = { lo, hi }; return r; }
__device__ __forceinline__ gr_v2f gr_v2_rint(gr_v2f a) { gr_v2f r = { rintf(a.x), rintf(a.y) }; return r; }
__device__ __forceinline__ gr_v2f gr_v2_floor(gr_v2f a) { gr_v2f r = { floorf(a.x), floorf(a.y) }; return r; }
__device__ __forceinline__ float gr_max3f(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// box constants of one frame as wave-uniform values (SGPRs)
struct GrBoxU {
    float ax, by, cz, bx, cx, cy, bcx, bcy, bcz, iax, iby, icz, rws2;
    bool tric;
};
__device__ __forceinline__ GrBoxU gr_box_uniform(const GrBox *__restrict__ b) {
    GrBoxU u;
    u.ax = b->ax; u.by = b->by; u.cz = b->cz; u.bx = b->bx; u.cx = b->cx; u.cy = b->cy;
    u.bcx = b->bcx; u.bcy = b->bcy; u.bcz = b->bcz; u.iax = b->iax; u.iby = b->iby; u.icz = b->icz;
    u.rws2 = b->r_ws * b->r_ws; u.tric = !b->ortho;
    return u;
}

// the three rows of a 4-atom group (gr_layout.h) are pairs of atoms per coordinate already
struct GrP4 { gr_v2f x01, y01, z01, x23, y23, z23; };
__device__ __forceinline__ GrP4 gr_pairs_rows(const float4 &r0, const float4 &r1, const float4 &r2) {
    GrP4 p;
    p.x01 = gr_v2p(r0.x, r0.y); p.y01 = gr_v2p(r0.z, r0.w); p.z01 = gr_v2p(r1.x, r1.y);
    p.x23 = gr_v2p(r1.z, r1.w); p.y23 = gr_v2p(r2.x, r2.y); p.z23 = gr_v2p(r2.z, r2.w);
    return p;
}
__device__ __forceinline__ void gr_rows_pairs(const GrP4 &p, float4 &r0, float4 &r1, float4 &r2) {
    r0 = make_float4(p.x01.x, p.x01.y, p.y01.x, p.y01.y); r1 = make_float4(p.z01.x, p.z01.y, p.x23.x, p.x23.y);
    r2 = make_float4(p.y23.x, p.y23.y, p.z23.x, p.z23.y);
}

// ------------------------------------------------------------------------------------------ pass 1: sums
// per-lane accumulators: every sum as a pair (even atoms, odd atoms), folded once after the loop
struct GrSumsPk {
    gr_v2f m, mx, my, mz;                 // sum m, sum m v
    gr_v2f a[9];                          // A = sum p v^T (unweighted, rmsd.rs:567-570)
    gr_v2f f1a, f1b, f1c, f2a, f2b, f2c;  // first / second moments of the fractional coordinates of v
    gr_v2f f3a, f3b, f3c;                 // third moments (NOREF, get_center / get_com: they pin the Bai-Breen estimate to within ~1e-3 of the mean)
    float mn[3], mx3[3], fmn[3], fmx[3];  // Cartesian and fractional extents of v
};

// the sums the closed-form RMSD needs on top of the rotation's (k_sums_pk<false, true>): B = sum (w p) v^T, sum w |v|^2 -- f32 products
// in SHORT chains (GR_RMSD_FLUSH trips = 8 atoms per half) that are widened to fp64 per lane before they grow: see k_sums_pk
struct GrRmsdPk { gr_v2f b[9]; gr_v2f wvv; gr_v2f wvb[2]; gr_v2f trb; };     // wvb: sum m |v|^2 once more, as two chains (atoms 01 / atoms 23 of every trip); trb: tr B = sum m (p . v) with the products associated the other way: the rounding probes
#ifndef GR_RMSD_FLUSH
#define GR_RMSD_FLUSH 4            // trips (4 atoms per lane each) between two flushes of the f32 chains into the lane's fp64 sums
#endif

// two difference vectors -> their minimum images: closed-form brick reduction along c, b, a (gr_image_about), which in a
// triclinic cell is already THE minimum image whenever |v| < r_ws; otherwise the image table is searched (rare: wave-uniform branch)
__device__ __forceinline__ void gr_image_pair(gr_v2f &vx, gr_v2f &vy, gr_v2f &vz, const GrBoxU &B, const GrBox *__restrict__ boxp) {
    gr_v2f k = gr_v2_rint(vz * gr_v2(B.icz));
    vx = gr_v2_fma(-k, gr_v2(B.cx), vx); vy = gr_v2_fma(-k, gr_v2(B.cy), vy); vz = gr_v2_fma(-k, gr_v2(B.cz), vz);
    k = gr_v2_rint(vy * gr_v2(B.iby));
    vx = gr_v2_fma(-k, gr_v2(B.bx), vx); vy = gr_v2_fma(-k, gr_v2(B.by), vy);
    k = gr_v2_rint(vx * gr_v2(B.iax));
    vx = gr_v2_fma(-k, gr_v2(B.ax), vx);
    if (B.tric) {
        const gr_v2f r2 = gr_v2_fma(vx, vx, gr_v2_fma(vy, vy, vz * vz));
        if (__builtin_amdgcn_ballot_w64(!(gr_fmaxf(r2.x, r2.y) < B.rws2)) != 0ull) {
            if (!(r2.x < B.rws2)) { float a = vx.x, b = vy.x, c = vz.x; gr_tric_refine(a, b, c, *boxp); vx.x = a; vy.x = b; vz.x = c; }
            if (!(r2.y < B.rws2)) { float a = vx.y, b = vy.y, c = vz.y; gr_tric_refine(a, b, c, *boxp); vx.y = a; vy.y = b; vz.y = c; }
        }
    }
}

// two atoms: v = image of (x - g) nearest to g, then every sum.  `p*` = reference coordinates (NOREF: unused), m = masses.
template <bool NOREF, bool RMSD = false, bool PRESUB = false>
__device__ __forceinline__ void gr_sums_pair(GrSumsPk &S, gr_v2f x, gr_v2f y, gr_v2f z, gr_v2f px, gr_v2f py, gr_v2f pz, gr_v2f m,
                                             const GrBoxU &B, const GrBox *__restrict__ boxp, float gx, float gy, float gz, GrRmsdPk *Rm = nullptr, int half = 0) {
    // (PRESUB: x, y, z are x - g already: the caller took the difference to free the row registers for the next request)
    gr_v2f vx = PRESUB ? x : x - gr_v2(gx), vy = PRESUB ? y : y - gr_v2(gy), vz = PRESUB ? z : z - gr_v2(gz);
    gr_image_pair(vx, vy, vz, B, boxp);
    // fractional coordinates of v: moments + extents feed the image proof (gr_finalize_math)
    const gr_v2f fc = vz * gr_v2(B.icz);
    const gr_v2f fb = gr_v2_fma(-fc, gr_v2(B.cy), vy) * gr_v2(B.iby);
    const gr_v2f fa = gr_v2_fma(-fc, gr_v2(B.cx), gr_v2_fma(-fb, gr_v2(B.bx), vx)) * gr_v2(B.iax);
    S.f1a += fa; S.f1b += fb; S.f1c += fc;
    S.f2a = gr_v2_fma(fa, fa, S.f2a); S.f2b = gr_v2_fma(fb, fb, S.f2b); S.f2c = gr_v2_fma(fc, fc, S.f2c);
    if (NOREF) { S.f3a = gr_v2_fma(fa * fa, fa, S.f3a); S.f3b = gr_v2_fma(fb * fb, fb, S.f3b); S.f3c = gr_v2_fma(fc * fc, fc, S.f3c); }
    S.fmn[0] = gr_min3f(S.fmn[0], fa.x, fa.y); S.fmn[1] = gr_min3f(S.fmn[1], fb.x, fb.y); S.fmn[2] = gr_min3f(S.fmn[2], fc.x, fc.y);
    S.fmx[0] = gr_max3f(S.fmx[0], fa.x, fa.y); S.fmx[1] = gr_max3f(S.fmx[1], fb.x, fb.y); S.fmx[2] = gr_max3f(S.fmx[2], fc.x, fc.y);
    S.mn[0] = gr_min3f(S.mn[0], vx.x, vx.y); S.mn[1] = gr_min3f(S.mn[1], vy.x, vy.y); S.mn[2] = gr_min3f(S.mn[2], vz.x, vz.y);
    S.mx3[0] = gr_max3f(S.mx3[0], vx.x, vx.y); S.mx3[1] = gr_max3f(S.mx3[1], vy.x, vy.y); S.mx3[2] = gr_max3f(S.mx3[2], vz.x, vz.y);
    S.m += m;
    S.mx = gr_v2_fma(m, vx, S.mx); S.my = gr_v2_fma(m, vy, S.my); S.mz = gr_v2_fma(m, vz, S.mz);
    if (!NOREF) {
        S.a[0] = gr_v2_fma(px, vx, S.a[0]); S.a[1] = gr_v2_fma(px, vy, S.a[1]); S.a[2] = gr_v2_fma(px, vz, S.a[2]);
        S.a[3] = gr_v2_fma(py, vx, S.a[3]); S.a[4] = gr_v2_fma(py, vy, S.a[4]); S.a[5] = gr_v2_fma(py, vz, S.a[5]);
        S.a[6] = gr_v2_fma(pz, vx, S.a[6]); S.a[7] = gr_v2_fma(pz, vy, S.a[7]); S.a[8] = gr_v2_fma(pz, vz, S.a[8]);
    }
    if (RMSD) {   // weights = masses (the caller checked): B = sum (m p) v^T, sum m |v|^2; sum m v is S.mx / my / mz
        const gr_v2f wx = m * px, wy = m * py, wz = m * pz;
        Rm->b[0] = gr_v2_fma(wx, vx, Rm->b[0]); Rm->b[1] = gr_v2_fma(wx, vy, Rm->b[1]); Rm->b[2] = gr_v2_fma(wx, vz, Rm->b[2]);
        Rm->b[3] = gr_v2_fma(wy, vx, Rm->b[3]); Rm->b[4] = gr_v2_fma(wy, vy, Rm->b[4]); Rm->b[5] = gr_v2_fma(wy, vz, Rm->b[5]);
        Rm->b[6] = gr_v2_fma(wz, vx, Rm->b[6]); Rm->b[7] = gr_v2_fma(wz, vy, Rm->b[7]); Rm->b[8] = gr_v2_fma(wz, vz, Rm->b[8]);
        const gr_v2f vv = gr_v2_fma(vx, vx, gr_v2_fma(vy, vy, vz * vz));
        Rm->wvv = gr_v2_fma(m, vv, Rm->wvv);
        Rm->wvb[half] = gr_v2_fma(m, vv, Rm->wvb[half]);
        Rm->trb = gr_v2_fma(m, gr_v2_fma(px, vx, gr_v2_fma(py, vy, pz * vz)), Rm->trb);      // = b[0] + b[4] + b[8], whose terms are (m p) v: other roundings of the same numbers
    }
}

// Same contract as k_rmsd_accum<0, true, NOREF> for a CONTIGUOUS selection (the caller checks): partial record per workgroup
// in `partials`, optional fused closing of the frame through `fuse` / `state_out` (see k_rmsd_accum).
//
// RMSD = true: the RMSD WITHOUT fit of a contiguous, mass-weighted selection in this one read-only pass (calc_rmsd, rmsd.rs:75-129,
// 141-166; round 3 ran it through k_rmsd_accum<0>: exact fp64 products, 74 instructions per atom, 3.8 us per 1e6-atom frame against
// a 1.7 us read stream).  On top of the rotation's sums the lane keeps B = sum (m p) v^T, sum m |v|^2 and sum m v as packed-f32
// chains of GR_RMSD_FLUSH trips (8 atoms per half) and widens them to fp64 before they grow -- 13 folds + conversions + fp64 adds
// per 16 atoms; the closing step evaluates rmsd^2 = (sum w|p|^2 + sum w|q|^2 - 2 tr(R^T Hw)) / W (rmsd.rs:592-599 expanded, as
// k_rmsd_finalize<0> does).  That expression is a small difference of large sums, and what the short chains leave of the f32
// rounding is a random walk over ~n / 8 independent partials: the closing step ESTIMATES it from the magnitudes of the sums and
// hands the frame back (GR_ST_REDO_EXACT -> the exact-product pass) when the estimate is not far below the 1e-5 nm bar --
// rigid copies of the reference (rmsd ~ 0), tiny groups; see gr_finalize_math<.., FAST>.
// MASK = true: a dense non-contiguous selection (GrSel::masked): the kernel walks the selection's whole span and takes every atom's
// membership from one bit (a 4-bit nibble per lane and trip, prefetched with the rows); plan.p is then laid out by atom over the span.
// Measured at 1e6 atoms (tools/gather_bench.py, us per frame, gather list -> masked span): every third atom 4.3 -> see DESIGN.
template <bool NOREF = false, bool RMSD = false, bool MASK = false>
// (three waves per SIMD forced through __launch_bounds__(GR_WG, 3): 168 VGPRs + 74 spilled, 9.0-9.6 us per 1e6-atom frame instead of 2.35 -- round 5)
__global__ __launch_bounds__(GR_WG) void k_sums_pk(
    const float *__restrict__ frames, size_t frame_stride, uint32_t first_slot,
    const float *__restrict__ masses, GrSel sel, const GrBox *__restrict__ boxes,
    GrPlanDev plan, GrAccPartial *partials, uint32_t *fuse, GrFrameState *state_out) {
    static_assert(!(NOREF && RMSD), "the RMSD needs the reference");
    __shared__ double lds[(GR_WG / 64) * GR_ACC_K];
    __shared__ double lds_pd[RMSD ? (GR_WG / 64) * 16 : 1];
    __shared__ double lds_acc[RMSD ? 15 * GR_WG : 1];        // RMSD: the lanes' fp64 sums [13][GR_WG] + the two rounding probes (30 KiB: no registers, one ds_add_f64 per value and flush)
    const uint32_t frame = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const float *xyz = frames + (size_t)(first_slot + frame) * frame_stride;
    const GrBox *boxp = boxes + first_slot + frame;
    const GrBoxU B = gr_box_uniform(boxp);
    const bool wm = plan.w_is_mass != 0;     // NOREF: "weighted"
    float gx, gy, gz;                         // provisional centre: the first atom of the selection
    gr_pos_load(xyz, sel.start, gx, gy, gz);
    GrSumsPk S;
    S.m = S.mx = S.my = S.mz = gr_v2(0.0f);
#pragma unroll
    for (int k = 0; k < 9; ++k) S.a[k] = gr_v2(0.0f);
    S.f1a = S.f1b = S.f1c = S.f2a = S.f2b = S.f2c = S.f3a = S.f3b = S.f3c = gr_v2(0.0f);
#pragma unroll
    for (int a = 0; a < 3; ++a) { S.mn[a] = S.fmn[a] = 3.0e38f; S.mx3[a] = S.fmx[a] = -3.0e38f; }
    GrRmsdPk Rm;
    // RMSD: the lane's fp64 sums -- B (9), sum m |v|^2, sum m v (3) -- live in LDS, one column per lane (as registers they cost the
    // kernel a wave per SIMD: 178 VGPRs, 2.8 us per 1e6-atom frame)
    uint32_t trips = 0;
    if (RMSD) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rm.b[k] = gr_v2(0.0f);
        Rm.wvv = Rm.wvb[0] = Rm.wvb[1] = Rm.trb = gr_v2(0.0f);
#pragma unroll
        for (int k = 0; k < 15; ++k) lds_acc[k * GR_WG + threadIdx.x] = 0.0;
    }
    auto widen = [&](int k, gr_v2f v) { (void)__hip_atomic_fetch_add(&lds_acc[k * GR_WG + threadIdx.x], (double)(v.x + v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
    auto flush = [&]() {
        // the probe: the same eight terms summed as two chains of four -- what the two associations differ by is a sample of the chains' rounding
        (void)__hip_atomic_fetch_add(&lds_acc[13 * GR_WG + threadIdx.x], (double)(Rm.wvv.x + Rm.wvv.y) - ((double)(Rm.wvb[0].x + Rm.wvb[0].y) + (double)(Rm.wvb[1].x + Rm.wvb[1].y)),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        Rm.wvb[0] = Rm.wvb[1] = gr_v2(0.0f);
        // ... and the trace of B against sum m (p . v): the same numbers with the products taken in the other order -- a sample of what forming
        // the TERMS in f32 costs (m p is rounded before it meets v), which repeats from atom to atom when the coordinates do
        (void)__hip_atomic_fetch_add(&lds_acc[14 * GR_WG + threadIdx.x], ((double)(Rm.b[0].x + Rm.b[0].y) + (double)(Rm.b[4].x + Rm.b[4].y) + (double)(Rm.b[8].x + Rm.b[8].y)) - (double)(Rm.trb.x + Rm.trb.y),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        Rm.trb = gr_v2(0.0f);
#pragma unroll
        for (int k = 0; k < 9; ++k) { widen(k, Rm.b[k]); Rm.b[k] = gr_v2(0.0f); }
        widen(9, Rm.wvv); Rm.wvv = gr_v2(0.0f);
        widen(10, S.mx); widen(11, S.my); widen(12, S.mz);
        S.mx = S.my = S.mz = gr_v2(0.0f);
    };

    const uint32_t first = sel.start, last = sel.start + (MASK ? sel.span : sel.n);
    const uint32_t g1 = (last + 3u) >> 2;                        // float4 groups [sel.g0 << 6, g1)
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    // ADDRESSES AND MASKS ARE THE WAVE'S, NOT THE LANE'S (round 5).  A wave walks whole 256-atom tiles -- tile T0 + k TS, both wave-uniform --
    // and a lane keeps its place inside the tile, so the frame, the reference, the masses and the mask bits are buffer resources read at
    // (lane's constant offset) + (an SGPR that a scalar add moves on), and "does this trip exist / is every atom of it inside the selection"
    // is a scalar comparison.  Until then a trip spent ~30 of its ~220 vector instructions on 64-bit address arithmetic, index clamps and the
    // per-atom bounds of a selection's ragged ends that 99.9 % of all trips do not touch.
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    const uint32_t tile_first = sel.g0, tile_end = (g1 + 63u) >> 6;                            // tiles [tile_first, tile_end)
    const uint32_t tstep = nchunks * (GR_WG / 64u);
    const __amdgpu_buffer_rsrc_t rs_f = gr_buf_rsrc(xyz, tile_end * 3072u);
    const __amdgpu_buffer_rsrc_t rs_p = gr_buf_rsrc(NOREF ? (const void *)xyz : (const void *)plan.p, NOREF ? 0u : (tile_end - tile_first) * 3072u);
    const __amdgpu_buffer_rsrc_t rs_m = gr_buf_rsrc(masses, tile_end * 1024u);
    const __amdgpu_buffer_rsrc_t rs_b = gr_buf_rsrc(MASK ? (const void *)sel.mask : (const void *)xyz, MASK ? tile_end * 32u : 0u);
    const uint32_t lane16 = lane * 16u, lane_bits = (lane >> 3) * 4u;
    // interior tiles: every atom inside [first, last) -- no lane of such a trip needs a mask (MASK: the bits decide, always)
    const uint32_t int_lo = (first + 255u) >> 8, int_hi = last >> 8;                          // tiles [int_lo, int_hi) are interior
#ifndef GR_SUMS_DEEP
#define GR_SUMS_DEEP 1             /* the frame's rows are requested TWO trips ahead (into the registers the current trip has just emptied), reference + masses one */
#endif
    struct Trip { float4 r0, r1, r2, q0, q1, q2, mm; uint32_t bits; };
    // (a trip behind the selection reads the selection's first tile: every request is unconditional, the trip count wave-uniform)
    auto request_rows = [&](uint32_t tile, Trip &t) {
#ifdef GR_EXP_SUMS_NOLOAD     /* experiment: the arithmetic alone (every trip works on the first trip's rows) */
        if (tile >= tile_first + (chunk + 2u * nchunks) * (GR_WG / 64u)) { asm volatile("" : "+v"(t.r0.x), "+v"(t.r0.y), "+v"(t.r0.z), "+v"(t.r0.w), "+v"(t.r1.x), "+v"(t.r1.y), "+v"(t.r1.z), "+v"(t.r1.w), "+v"(t.r2.x), "+v"(t.r2.y), "+v"(t.r2.z), "+v"(t.r2.w)); return; }
#endif
        const uint32_t tc = tile < tile_end ? tile : tile_first, so = tc * 3072u;
        t.r0 = gr_buf_load_f4<2>(rs_f, lane16, so); t.r1 = gr_buf_load_f4<2>(rs_f, lane16 + 1024u, so); t.r2 = gr_buf_load_f4<2>(rs_f, lane16 + 2048u, so);
    };
    auto request_pm = [&](uint32_t tile, Trip &t) {
#if defined(GR_EXP_SUMS_NOLOAD) || defined(GR_EXP_SUMS_NOPM)     /* NOPM: reference + masses loaded for the first two trips only */
        if (tile >= tile_first + (chunk + 2u * nchunks) * (GR_WG / 64u)) return;
#endif
        const uint32_t tc = tile < tile_end ? tile : tile_first;
        t.bits = MASK ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_b, (int)lane_bits, (int)(tc * 32u), 0) : 0u;
        if (!NOREF) {
            const uint32_t so = (tc - tile_first) * 3072u;
            t.q0 = gr_buf_load_f4<0>(rs_p, lane16, so); t.q1 = gr_buf_load_f4<0>(rs_p, lane16 + 1024u, so); t.q2 = gr_buf_load_f4<0>(rs_p, lane16 + 2048u, so);
        }
        if (!(NOREF && !wm)) t.mm = gr_buf_load_f4<0>(rs_m, lane16, tc * 1024u); else t.mm = one4;
    };
    auto process = [&](Trip &t, uint32_t tile) {
        const float4 q0 = t.q0, q1 = t.q1, q2 = t.q2;
        float4 mm = t.mm;
        GrP4 q = gr_pairs_rows(t.r0, t.r1, t.r2), p;
        if (!NOREF) p = gr_pairs_rows(q0, q1, q2); else p.x01 = p.y01 = p.z01 = p.x23 = p.y23 = p.z23 = gr_v2(0.0f);
        // which of the lane's four atoms count: inside [first, last), in a trip that exists, bit set (MASK).  The others -- a ragged end
        // of the selection, a trip behind it, an atom whose bit is clear -- become copies of the first atom with zero mass and zero
        // reference coordinates: v = 0 adds nothing to any sum and lies inside every extent (the first atom's own v is 0), and
        // whatever such an atom holds, a NaN included, never reaches the arithmetic.  One wave-uniform branch around plain selects:
        // written as per-atom `if`s under a divergent `if`, the masked kernel came out of the compiler with atom 0's coordinates
        // NOT replaced in one of the two unrolled copies (its mass was) -- its fractional coordinates leaked into the image proof.
        const bool interior = tile >= int_lo && tile < int_hi;                                 // (scalar)
        if (MASK || !interior) {
            const uint32_t gg = (tile << 6) + lane, i = gg << 2;
            const uint32_t nib = MASK ? (t.bits >> ((gg & 7u) * 4u)) & 15u : 15u;             // MASK: which of the lane's four atoms are selected
            uint32_t keep = tile < tile_end ? nib : 0u;
            if (!interior) {
                keep &= (i >= first ? 1u : 0u) | (i + 1 >= first ? 2u : 0u) | (i + 2 >= first ? 4u : 0u) | (i + 3 >= first ? 8u : 0u);
                keep &= (i < last ? 1u : 0u) | (i + 1 < last ? 2u : 0u) | (i + 2 < last ? 4u : 0u) | (i + 3 < last ? 8u : 0u);
            }
            if (__builtin_amdgcn_ballot_w64(keep != 15u) != 0ull) {
                const bool k0 = (keep & 1u) != 0, k1 = (keep & 2u) != 0, k2 = (keep & 4u) != 0, k3 = (keep & 8u) != 0;
                q.x01.x = k0 ? q.x01.x : gx; q.y01.x = k0 ? q.y01.x : gy; q.z01.x = k0 ? q.z01.x : gz; mm.x = k0 ? mm.x : 0.f;
                q.x01.y = k1 ? q.x01.y : gx; q.y01.y = k1 ? q.y01.y : gy; q.z01.y = k1 ? q.z01.y : gz; mm.y = k1 ? mm.y : 0.f;
                q.x23.x = k2 ? q.x23.x : gx; q.y23.x = k2 ? q.y23.x : gy; q.z23.x = k2 ? q.z23.x : gz; mm.z = k2 ? mm.z : 0.f;
                q.x23.y = k3 ? q.x23.y : gx; q.y23.y = k3 ? q.y23.y : gy; q.z23.y = k3 ? q.z23.y : gz; mm.w = k3 ? mm.w : 0.f;
                if (!NOREF) {
                    p.x01.x = k0 ? p.x01.x : 0.f; p.y01.x = k0 ? p.y01.x : 0.f; p.z01.x = k0 ? p.z01.x : 0.f;
                    p.x01.y = k1 ? p.x01.y : 0.f; p.y01.y = k1 ? p.y01.y : 0.f; p.z01.y = k1 ? p.z01.y : 0.f;
                    p.x23.x = k2 ? p.x23.x : 0.f; p.y23.x = k2 ? p.y23.x : 0.f; p.z23.x = k2 ? p.z23.x : 0.f;
                    p.x23.y = k3 ? p.x23.y : 0.f; p.y23.y = k3 ? p.y23.y : 0.f; p.z23.y = k3 ? p.z23.y : 0.f;
                }
            }
        }
#ifdef GR_EXP_SUMS_NOMATH      /* experiment: the loads alone (every loaded value is consumed by one add) */
        {
            S.m += (q.x01 + q.y01) + (q.z01 + q.x23) + (q.y23 + q.z23) + (p.x01 + p.y01) + (p.z01 + p.x23) + (p.y23 + p.z23) + gr_v2p(mm.x + mm.z, mm.y + mm.w);
            if (GR_SUMS_DEEP) request_rows(tile + 2u * tstep, t);
            return;
        }
#endif
        if (GR_SUMS_DEEP) {
            // the differences first: the row registers are then free, and the rows of the trip after next are requested into them now
            q.x01 -= gr_v2(gx); q.y01 -= gr_v2(gy); q.z01 -= gr_v2(gz); q.x23 -= gr_v2(gx); q.y23 -= gr_v2(gy); q.z23 -= gr_v2(gz);
            asm volatile("" :: "v"(q.x01), "v"(q.y01), "v"(q.z01), "v"(q.x23), "v"(q.y23), "v"(q.z23));
            request_rows(tile + 2u * tstep, t);
            gr_sums_pair<NOREF, RMSD, true>(S, q.x01, q.y01, q.z01, p.x01, p.y01, p.z01, gr_v2p(mm.x, mm.y), B, boxp, gx, gy, gz, &Rm, 0);
            gr_sums_pair<NOREF, RMSD, true>(S, q.x23, q.y23, q.z23, p.x23, p.y23, p.z23, gr_v2p(mm.z, mm.w), B, boxp, gx, gy, gz, &Rm, 1);
        } else {
            gr_sums_pair<NOREF, RMSD>(S, q.x01, q.y01, q.z01, p.x01, p.y01, p.z01, gr_v2p(mm.x, mm.y), B, boxp, gx, gy, gz, &Rm, 0);
            gr_sums_pair<NOREF, RMSD>(S, q.x23, q.y23, q.z23, p.x23, p.y23, p.z23, gr_v2p(mm.z, mm.w), B, boxp, gx, gy, gz, &Rm, 1);
        }
        if (RMSD && (++trips % GR_RMSD_FLUSH) == 0) flush();
    };
    // Software pipeline with TWO NAMED register sets and the loop unrolled by two: reference + masses of trip k + 1 are requested before
    // the arithmetic of trip k, the frame's rows of trip k + 2 inside it (GR_SUMS_DEEP), and the only wait before the arithmetic is "all but
    // the loads issued since".  (Rounds 2-3 wrote this as `if (next < end) request(next, n..); ...; r = n` -- and the compiler, which has
    // to merge the loaded and the not-loaded value at the join and to copy the landing registers at the end of the body, waited for the
    // new loads right after issuing them and drained the queue at the end of every trip: nothing was prefetched.)
    uint32_t tile = tile_first + chunk * (GR_WG / 64u) + wave_u;
    if (tile < tile_end) {
        Trip TA, TB;
        request_rows(tile, TA); request_pm(tile, TA);
        if (GR_SUMS_DEEP) request_rows(tile + tstep, TB);
        for (;;) {
            if (!GR_SUMS_DEEP) request_rows(tile + tstep, TB);
            request_pm(tile + tstep, TB);
            process(TA, tile);
            tile += tstep;
            if (tile >= tile_end) break;
            if (!GR_SUMS_DEEP) request_rows(tile + tstep, TA);
            request_pm(tile + tstep, TA);
            process(TB, tile);
            tile += tstep;
            if (tile >= tile_end) break;
        }
    }
    // epilogue (as k_rmsd_accum's LITE epilogue): every wave reduce-scatters its 19 sums and 12 extents, the four waves
    // meet in LDS once, and 33 lanes of wave 0 write the record
    float s32[32], e32[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) { s32[k] = 0.0f; e32[k] = -3.0e38f; }
    if (RMSD) {
        // the lane's fp64 sums -> the wave (reduce-scatter in fp64) -> LDS; sum m v of the record is the fp64 one, rounded
        flush();
        double d32[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) d32[k] = k < 15 ? lds_acc[k * GR_WG + threadIdx.x] : 0.0;
        S.mx = gr_v2p((float)d32[10], 0.0f); S.my = gr_v2p((float)d32[11], 0.0f); S.mz = gr_v2p((float)d32[12], 0.0f);
        const double dt = gr_wave_sum_scatter16_f64(d32, lane);           // lane l: the wave total of value l >> 2
        if ((lane & 3u) == 0) lds_pd[wave * 16 + (lane >> 2)] = dt;
    }
    s32[0] = S.m.x + S.m.y; s32[1] = S.mx.x + S.mx.y; s32[2] = S.my.x + S.my.y; s32[3] = S.mz.x + S.mz.y;
#pragma unroll
    for (int k = 0; k < 9; ++k) s32[4 + k] = S.a[k].x + S.a[k].y;
    s32[13] = S.f1a.x + S.f1a.y; s32[14] = S.f1b.x + S.f1b.y; s32[15] = S.f1c.x + S.f1c.y;
    s32[16] = S.f2a.x + S.f2a.y; s32[17] = S.f2b.x + S.f2b.y; s32[18] = S.f2c.x + S.f2c.y;
    if (NOREF) { s32[19] = S.f3a.x + S.f3a.y; s32[20] = S.f3b.x + S.f3b.y; s32[21] = S.f3c.x + S.f3c.y; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { e32[k] = -S.mn[k]; e32[3 + k] = S.mx3[k]; e32[6 + k] = -S.fmn[k]; e32[9 + k] = S.fmx[k]; }
    const float tot = gr_wave_sum_scatter32(s32, lane);
    const float emax = gr_wave_max_scatter16(e32, lane);
    float *wsum = reinterpret_cast<float *>(lds);            // [4 waves][32 sums | 16 maxima]
    if ((lane & 1u) == 0) wsum[wave * 48 + (lane >> 1)] = tot;
    if ((lane & 3u) == 0) wsum[wave * 48 + 32 + (lane >> 2)] = emax;
    __syncthreads();
    if (wave != 0) return;
    GrAccPartial &o = partials[(size_t)frame * nchunks + chunk];
    if (lane < 19) {
        const double v = (double)wsum[lane] + (double)wsum[48 + lane] + (double)wsum[96 + lane] + (double)wsum[144 + lane];
        gr_st_agent(&o.s[lane < 13 ? lane : 13 + lane], v);           // sums 13..18 are the moments: record slots 26..31
    } else if (lane < 32) {
        // slots 13..25: B (Hw before the centring), sum w |v|^2, sum w v -- the closed-form RMSD's sums; unused (zero) by the two-pass fit
        double v = 0.0;
        if (RMSD) { const uint32_t q = lane - 19; v = ((lds_pd[q] + lds_pd[16 + q]) + lds_pd[32 + q]) + lds_pd[48 + q]; }
        else if (NOREF && lane < 22) v = (double)wsum[lane] + (double)wsum[48 + lane] + (double)wsum[96 + lane] + (double)wsum[144 + lane];   // third moments: slots 13..15
        gr_st_agent(&o.s[lane - 6], v);
    } else if (lane < 44) {
        const uint32_t q = lane - 32;
        const float m = gr_fmaxf(gr_fmaxf(wsum[32 + q], wsum[48 + 32 + q]), gr_fmaxf(wsum[96 + 32 + q], wsum[144 + 32 + q]));
        if (q < 3) gr_st_agent(&o.vmin[q], -m); else if (q < 6) gr_st_agent(&o.vmax[q - 3], m); else if (q < 9) gr_st_agent(&o.fmin[q - 6], -m); else gr_st_agent(&o.fmax[q - 9], m);
    } else if (lane == 44) {
        gr_st_agent(&o.bad_pos, GR_NOIDX);      // no per-atom tests here: a missing position / mass shows up as a NaN sum
        gr_st_agent(&o.bad_mass, GR_NOIDX);
        if (RMSD) {
            // the chunk's measured roundings -- sum m |v|^2 as one chain of 8 against two of 4; tr B with the products taken in the other order --
            // against the model's figure for a chunk of this many atoms, 6e-8 (magnitude of the terms) sqrt(20 / n): see gr_finalize_math<.., FAST>
            const double d1 = ((lds_pd[13] + lds_pd[16 + 13]) + lds_pd[32 + 13]) + lds_pd[48 + 13], d2 = ((lds_pd[14] + lds_pd[16 + 14]) + lds_pd[32 + 14]) + lds_pd[48 + 14];
            const double ssum = ((lds_pd[9] + lds_pd[16 + 9]) + lds_pd[32 + 9]) + lds_pd[48 + 9];
            const double n_chunk = fmax(8.0, (double)(MASK ? sel.span : sel.n) / (double)nchunks);
            const double expect = 6.0e-8 * 0.5 * (ssum + plan.swpp / (double)nchunks) * sqrt(20.0 / n_chunk);
            const double dsum = fmax(fabs(d1), fabs(d2));
            gr_st_agent(&o.rd, expect > 0.0 ? (float)fmin(fabs(dsum) / expect, 1.0e30) : 0.0f);
        }
    }
    if (fuse) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every lane's record stores have left
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(fuse + frame, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(old) == nchunks - 1) {     // this frame's records are complete: close it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            double *totd = lds;                                       // wsum (same LDS) has been consumed by this wave
            float *ext = reinterpret_cast<float *>(lds + 32);
            gr_finalize_frame_lite<NOREF, RMSD>(partials, nchunks, frame, frames, frame_stride, first_slot, sel, boxes, plan, state_out, totd, ext, lane);
            if (lane == 0) fuse[frame] = 0u;
        }
    }
}

// ------------------------------------------------------------------------------------------ pass 2: fit
// one-turn wrap of two atoms along c, b, a: k = floor(t / L) per stage; `lo` / `hi` collect the smallest and largest
// result per axis so that ONE test per tile decides whether every result landed in (0, L]
__device__ __forceinline__ void gr_wrap_pair_fast(gr_v2f &x, gr_v2f &y, gr_v2f &z, const GrBoxU &B) {
    gr_v2f k = gr_v2_floor(z * gr_v2(B.icz));
    x = gr_v2_fma(-k, gr_v2(B.cx), x); y = gr_v2_fma(-k, gr_v2(B.cy), y); z = gr_v2_fma(-k, gr_v2(B.cz), z);
    k = gr_v2_floor(y * gr_v2(B.iby));
    x = gr_v2_fma(-k, gr_v2(B.bx), x); y = gr_v2_fma(-k, gr_v2(B.by), y);
    k = gr_v2_floor(x * gr_v2(B.iax));
    x = gr_v2_fma(-k, gr_v2(B.ax), x);
}

// A wave owns ONE group row set per trip and, at the default grid (one 256-atom tile per wave), one trip: whatever a wave
// waits for in sequence is exposed, so the tile and -- for the rmsd -- the reference coordinates and weights of the tile are
// requested first (they depend on the kernel arguments only), then the frame state and the box (dependent scalar loads).
template <bool RMSD>
__global__ __launch_bounds__(GR_WG) void k_fit_pk(
    float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, uint32_t n_atoms,
    const GrBox *__restrict__ boxes, GrPlanDev plan, const GrFrameState *__restrict__ state,
    const float *__restrict__ masses, GrSel sel, double *fit_partials) {
    const uint32_t frame = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *xyz = frames + (size_t)(first_slot + frame) * frame_stride;
    float4 *f4 = reinterpret_cast<float4 *>(xyz);
    const float4 *p4 = reinterpret_cast<const float4 *>(plan.p);
    const float4 *w4 = plan.w_is_mass ? reinterpret_cast<const float4 *>(masses) : reinterpret_cast<const float4 *>(plan.w);
    const uint32_t first = sel.start, last = sel.start + sel.n, g0 = sel.g0 << 6;
    const uint32_t wofs = plan.w_is_mass ? 0u : g0;        // masses are indexed by atom group, plan.w by selection group
    const uint32_t ngroups = ((n_atoms + 255u) >> 8) << 6;   // the slot is padded to whole tiles; pad atoms are transformed too (harmless)
    const uint32_t gstep = gridDim.x * GR_WG;
    uint32_t g = blockIdx.x * GR_WG + threadIdx.x;
    if (g >= ngroups && !RMSD) return;
    float4 r0, r1, r2, pa, pb, pc, ww;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    r0 = r1 = r2 = pa = pb = pc = ww = zero4;
    bool in_sel = false;
    auto request = [&](uint32_t gg) {
        gr_rows_load<true>(f4, gg, r0, r1, r2);   // non-temporal loads AND stores measured best (4.28 us/frame; plain loads 4.6, plain stores 4.5)
        if (RMSD) {
            const uint32_t i = gg << 2;
            in_sel = (i + 3 >= first) && (i < last);
            if (in_sel) { gr_rows_load(p4, (size_t)(gg - g0), pa, pb, pc); ww = w4[gg - wofs]; }
        }
    };
    if (g < ngroups) request(g);
    const GrFrameState &st = state[frame];
    const GrBox *boxp = boxes + first_slot + frame;
    const GrBoxU B = gr_box_uniform(boxp);
    const float sx = st.shift[0], sy = st.shift[1], sz = st.shift[2];
    const float r00 = st.R[0], r10 = st.R[1], r20 = st.R[2], r01 = st.R[3], r11 = st.R[4], r21 = st.R[5], r02 = st.R[6], r12 = st.R[7], r22 = st.R[8];
    const float cx = plan.ref_com[0], cy = plan.ref_com[1], cz = plan.ref_com[2];
    if (st.status != 0) return;   // analysis failed -> frame left unmodified (rmsd.rs:91)
    double rs = 0.0;
    while (g < ngroups) {
        GrP4 q = gr_pairs_rows(r0, r1, r2);
        q.x01 += gr_v2(sx); q.y01 += gr_v2(sy); q.z01 += gr_v2(sz); q.x23 += gr_v2(sx); q.y23 += gr_v2(sy); q.z23 += gr_v2(sz);
        gr_wrap_pair_fast(q.x01, q.y01, q.z01, B);
        gr_wrap_pair_fast(q.x23, q.y23, q.z23, B);
        {
            // every wrapped coordinate in (0, L]?  (NaN fails; an atom exactly on a face, or whose k came out one off by the
            // rounding of t * (1/L), fails too) -- otherwise the wave redoes its atoms with the general wrap
            const float xl = gr_fminf(gr_min3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y), xh = gr_fmaxf(gr_max3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y);
            const float yl = gr_fminf(gr_min3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y), yh = gr_fmaxf(gr_max3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y);
            const float zl = gr_fminf(gr_min3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y), zh = gr_fmaxf(gr_max3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y);
            const bool ok = (xl > 0.0f) & (xh <= B.ax) & (yl > 0.0f) & (yh <= B.by) & (zl > 0.0f) & (zh <= B.cz);
            if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {
                float x[4], y[4], z[4];
                gr_rows_unpack(r0, r1, r2, x, y, z);
                float x0 = x[0] + sx, y0 = y[0] + sy, z0 = z[0] + sz, x1 = x[1] + sx, y1 = y[1] + sy, z1 = z[1] + sz;
                float x2 = x[2] + sx, y2 = y[2] + sy, z2 = z[2] + sz, x3 = x[3] + sx, y3 = y[3] + sy, z3 = z[3] + sz;
                gr_wrap(x0, y0, z0, *boxp); gr_wrap(x1, y1, z1, *boxp); gr_wrap(x2, y2, z2, *boxp); gr_wrap(x3, y3, z3, *boxp);
                q.x01 = gr_v2p(x0, x1); q.x23 = gr_v2p(x2, x3); q.y01 = gr_v2p(y0, y1); q.y23 = gr_v2p(y2, y3); q.z01 = gr_v2p(z0, z1); q.z23 = gr_v2p(z2, z3);
            }
        }
        // q = wrap(x + shift) - box centre; R q
        q.x01 -= gr_v2(B.bcx); q.y01 -= gr_v2(B.bcy); q.z01 -= gr_v2(B.bcz); q.x23 -= gr_v2(B.bcx); q.y23 -= gr_v2(B.bcy); q.z23 -= gr_v2(B.bcz);
        GrP4 n;
        n.x01 = gr_v2_fma(gr_v2(r02), q.z01, gr_v2_fma(gr_v2(r01), q.y01, gr_v2(r00) * q.x01));
        n.y01 = gr_v2_fma(gr_v2(r12), q.z01, gr_v2_fma(gr_v2(r11), q.y01, gr_v2(r10) * q.x01));
        n.z01 = gr_v2_fma(gr_v2(r22), q.z01, gr_v2_fma(gr_v2(r21), q.y01, gr_v2(r20) * q.x01));
        n.x23 = gr_v2_fma(gr_v2(r02), q.z23, gr_v2_fma(gr_v2(r01), q.y23, gr_v2(r00) * q.x23));
        n.y23 = gr_v2_fma(gr_v2(r12), q.z23, gr_v2_fma(gr_v2(r11), q.y23, gr_v2(r10) * q.x23));
        n.z23 = gr_v2_fma(gr_v2(r22), q.z23, gr_v2_fma(gr_v2(r21), q.y23, gr_v2(r20) * q.x23));
        if (RMSD && in_sel) {
            const uint32_t i = g << 2;
            const GrP4 p = gr_pairs_rows(pa, pb, pc);
            // sum w |R q - p|^2 (rmsd.rs:592-599): two atoms per packed operation, 4-atom f32 partial -> fp64 per lane.  At a ragged end
            // of the selection the atoms outside it contribute an exact zero -- their TERM, not just their weight: an atom without a
            // position next to the selection's edge would otherwise turn the sum into 0 * NaN
            gr_v2f dx = n.x01 - p.x01, dy = n.y01 - p.y01, dz = n.z01 - p.z01;
            gr_v2f s01 = gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz));
            dx = n.x23 - p.x23; dy = n.y23 - p.y23; dz = n.z23 - p.z23;
            gr_v2f s23 = gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz));
            const uint32_t keep = ((i >= first && i < last) ? 1u : 0u) | ((i + 1 >= first && i + 1 < last) ? 2u : 0u) | ((i + 2 >= first && i + 2 < last) ? 4u : 0u) | ((i + 3 >= first && i + 3 < last) ? 8u : 0u);
            if (__builtin_amdgcn_ballot_w64(keep != 15u) != 0ull) {
                s01.x = (keep & 1u) ? s01.x : 0.f; s01.y = (keep & 2u) ? s01.y : 0.f; s23.x = (keep & 4u) ? s23.x : 0.f; s23.y = (keep & 8u) ? s23.y : 0.f;
                ww.x = (keep & 1u) ? ww.x : 0.f; ww.y = (keep & 2u) ? ww.y : 0.f; ww.z = (keep & 4u) ? ww.z : 0.f; ww.w = (keep & 8u) ? ww.w : 0.f;
            }
            const gr_v2f part = gr_v2_fma(gr_v2p(ww.z, ww.w), s23, gr_v2p(ww.x, ww.y) * s01);
            rs += (double)(part.x + part.y);
        }
        n.x01 += gr_v2(cx); n.y01 += gr_v2(cy); n.z01 += gr_v2(cz); n.x23 += gr_v2(cx); n.y23 += gr_v2(cy); n.z23 += gr_v2(cz);
        float4 o0, o1, o2;
        gr_rows_pairs(n, o0, o1, o2);
        gr_rows_store<true>(f4, g, o0, o1, o2);
        g += gstep;
        if (g < ngroups) request(g);
    }
    if (RMSD) {
        // the wave's share goes straight to memory, one fp64 word per wave (k_rmsd_close adds a frame's words in a fixed order): no LDS
        // crossbar, no barrier, no second hop -- at the default grid a workgroup is ONE trip, and this epilogue used to cost it twelve
        // ds_bpermute and a barrier (round 5)
        const double v = gr_xor_lane_f64<1>(rs) + rs;
        const double w = gr_xor_lane_f64<2>(v) + v;
        const double x = gr_xor_lane_f64<4>(w) + w;
        const double y = gr_xor_lane_f64<8>(x) + x;
        const unsigned long long yb = (unsigned long long)__double_as_longlong(y);
        const auto l16 = __builtin_amdgcn_permlane16_swap((uint32_t)yb, (uint32_t)yb, false, false), h16 = __builtin_amdgcn_permlane16_swap((uint32_t)(yb >> 32), (uint32_t)(yb >> 32), false, false);
        const double z = gr_f64_from_halves(l16[0], h16[0]) + gr_f64_from_halves(l16[1], h16[1]);
        const unsigned long long zb = (unsigned long long)__double_as_longlong(z);
        const auto l32 = __builtin_amdgcn_permlane32_swap((uint32_t)zb, (uint32_t)zb, false, false), h32 = __builtin_amdgcn_permlane32_swap((uint32_t)(zb >> 32), (uint32_t)(zb >> 32), false, false);
        const double tot = gr_f64_from_halves(l32[0], h32[0]) + gr_f64_from_halves(l32[1], h32[1]);
        if (lane == 0) fit_partials[((size_t)frame * gridDim.x + blockIdx.x) * (GR_WG / 64) + wave] = tot;
    }
}

