// gr_resident.h -- the RMSD fit as ONE pass over HBM when a frame fits on the chip.
//
// The two-pass path (gr_hot.h) reads every frame twice: once for the sums that give the rotation, once to apply it -- the
// rotation of a frame depends on all of its atoms, so no atom can be written before every atom has been read.  36 bytes per
// atom and frame cross HBM (12 + 12 read, 12 written), plus 32 bytes per atom and frame of reference coordinates and
// weights from the L2 / Infinity Cache.  But a 1e6-atom frame is 12 MB, and the chip has 256 CUs x 512 KiB of vector
// registers + 160 KiB of LDS: the frame can WAIT ON CHIP for its rotation.
//
//   k_fit_resident   one cooperative launch for a whole batch: `n_stream` workgroups of 1024 lanes, one 4-atom group per lane
//                    for the whole launch (its reference coordinates, masses and weights are loaded once and stay in
//                    registers); a lane walks the frames of the batch with its own group:
//                       sums stage, frame i      rows arrive (requested one frame earlier), image about the first atom, the
//                                                19 sums + 12 extents of its 4 atoms -> wave reduce-scatter -> LDS; the last
//                                                wave of the workgroup adds the 16 wave records in wave order and writes the
//                                                workgroup's record: 31 tagged words (value | epoch << 32), no flag, no fence;
//                                                the rows are parked in LDS (48 KiB per frame and CU)
//                       fit stage, frame i - 3   the frame's record (status, shift, R) has come back from a finalizer ->
//                                                rows out of LDS, wrap + rotate + translate, sum w |R q - p|^2, one
//                                                non-temporal store per row
//                    + `n_fin` workgroups that stream nothing: finalizer j owns the frames j, j + n_fin, ...; its 16 waves
//                    read 16 workgroup records each (one load round trip for the whole frame), re-reading until every
//                    word carries the launch's tag; a fixed tree adds them in fp64, one lane closes the frame exactly as
//                    the two-pass path does (gr_finalize_math: image proof, Kabsch rotation in fp64) and publishes
//                    status | shift | R as 13 tagged 64-bit words.  The frame's box, first atom and host-side status are
//                    requested BEFORE the wait: the finalize is the latency the parked frames have to cover.
// HBM traffic: 12 bytes per atom read + 12 written per frame = 24 (was 36), and nothing from the caches.
// Measured floor of that traffic at the same launch shape (tools/ceiling_bench.hip "resident copy"): 4.2 us per 1e6-atom frame.
//
// Synchronisation.  All waiting is on data that a DIFFERENT workgroup produces, so every workgroup must be resident: the
// kernel is launched with hipLaunchCooperativeKernel (refused by the runtime unless the whole grid fits at once) and is only
// chosen when the grid fits with one workgroup per CU.  Every wait is bounded (GR_RES_PATIENCE polls with s_sleep, a few
// seconds): a wave that runs out of patience raises `abort` and leaves, every other wait then ends too, the grid drains and
// the host reports the batch as failed.  Every word that crosses between workgroups carries the launch's epoch in its upper
// half and is written / read as ONE 64-bit access at agent scope (bypassing the non-coherent caches): a reader can never
// take a word of an older launch for a new one, nothing has to be cleared between launches, and no store has to be ordered
// against another -- which keeps "s_waitcnt vmcnt(0)" (a drain of the wave's prefetched rows) out of the streaming loop.
// Order within a workgroup is free (waves combine through LDS counters, no barrier); results do not depend on it: the wave
// records are added in wave order by whichever wave arrives last, the workgroup records by a fixed tree in the finalizer.
// Precision: a workgroup record is the f32 sum of 16 f32 wave sums (4096 atoms; the two-pass path sums ~7800 atoms in f32
// before it widens), the finalizer adds the workgroup records in fp64.
#pragma once
#include "gr_hot.h"

#define GR_RES_LANES 1024
#define GR_RES_WAVES (GR_RES_LANES / 64)
#define GR_RES_K 3                 // frames between the sums stage and the fit stage (= frames parked in LDS)
#define GR_RES_R 4                 // ring of wave-record slots ( > GR_RES_K: no wave is more than K frames ahead of another)
#define GR_RES_MAX_FIN 8
#define GR_RES_PATIENCE 3000000u   // polls (each ~1 us of s_sleep + a load) before a wait gives up

#define GR_RES_REC_WORDS 32         // tagged words per workgroup record: 0..18 sums, 19..30 extents (as maxima), 31 unused
struct GrResCtl {
    unsigned long long *wgrec;     // [frames][n_stream_pad][32] value | epoch << 32 (n_stream_pad = n_stream rounded up to 16)
    unsigned long long *rec;       // [frames][16] value | epoch << 32: 0 status, 1..3 shift, 4..12 R (column-major)
    uint32_t *abort;               // one word, 0 = fine
    uint32_t epoch, n_stream, n_fin;
};

// LDS of a streaming workgroup (dynamic: more than the 64 KiB static limit)
#define GR_RES_PARK_F4 (GR_RES_K * 3 * GR_RES_LANES)                        // float4
#define GR_RES_WSUM_F (GR_RES_R * GR_RES_WAVES * 48)                        // float
#define GR_RES_LDS_BYTES (GR_RES_PARK_F4 * 16 + GR_RES_WSUM_F * 4 + GR_RES_R * GR_RES_WAVES * 8 + 2 * GR_RES_R * 4)

template <typename T> __device__ __forceinline__ T gr_ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float gr_first_f(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
__device__ __forceinline__ float gr_lane_f(unsigned long long v, int l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l)); }

template <bool WMASS>
__global__ __launch_bounds__(GR_RES_LANES) void k_fit_resident(
    float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, uint32_t nframes, uint32_t n_atoms,
    const float *__restrict__ masses, GrSel sel, const GrBox *__restrict__ boxes, GrPlanDev plan,
    GrFrameState *state, double *__restrict__ fit_partials, GrResCtl ctl) {
    extern __shared__ float4 smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    // ------------------------------------------------------------------------------------------ finalizers
    if (blockIdx.x >= ctl.n_stream) {
        double *wtot = reinterpret_cast<double *>(smem);              // [16 waves][32]
        const uint32_t n_pad = (ctl.n_stream + 15u) & ~15u;
        const uint32_t r = wave * 16u + (lane >> 2), part = lane & 3u;   // this lane's record and its 8 words
        const unsigned long long tagv = (unsigned long long)ctl.epoch << 32;
        for (uint32_t f = blockIdx.x - ctl.n_stream; f < nframes; f += ctl.n_fin) {
            // what the closing step needs besides the sums: requested now, in flight while the records are awaited
            const GrBox *bp = boxes + first_slot + f;
            GrBox lb;
            lb.ax = bp->ax; lb.by = bp->by; lb.cz = bp->cz; lb.bx = bp->bx; lb.cx = bp->cx; lb.cy = bp->cy;
            lb.bcx = bp->bcx; lb.bcy = bp->bcy; lb.bcz = bp->bcz; lb.r_ws = bp->r_ws; lb.ortho = bp->ortho;
            float g0x, g0y, g0z;
            gr_pos_load(frames + (size_t)(first_slot + f) * frame_stride, sel.start, g0x, g0y, g0z);
            const int pre_status = state[f].status;
            const unsigned long long *src = ctl.wgrec + ((size_t)f * n_pad + r) * GR_RES_REC_WORDS + part * 8u;
            unsigned long long w[8];
            uint32_t polls = 0;
#ifdef GR_DBG_RES
            const unsigned long long tf0 = wall_clock64();
#endif
            for (;;) {
                bool ok = true;
                if (r < ctl.n_stream) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) w[k] = gr_ld_agent(src + k);
#pragma unroll
                    for (int k = 0; k < 8; ++k) ok = ok && ((w[k] >> 32) == (unsigned long long)ctl.epoch || (part == 3u && k == 7));
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (++polls > GR_RES_PATIENCE || ((polls & 255u) == 0 && gr_ld_agent(ctl.abort) != 0u)) { if (lane == 0) gr_st_agent(ctl.abort, 1u); polls = 0xFFFFFFFFu; break; }
                __builtin_amdgcn_s_sleep(2);
            }
#ifdef GR_DBG_RES
            const unsigned long long tf1 = wall_clock64();
#endif
            // (a wave that gave up still meets the others at the barriers; the abort word ends the launch)
            // sums in fp64, extents as maxima: word index = part * 8 + k; 0..18 sums, 19..30 maxima
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool mxw = part == 3u || (part == 2u && k >= 3);          // words 19..31 are maxima
                const float x = (r < ctl.n_stream && polls != 0xFFFFFFFFu) ? __uint_as_float((uint32_t)w[k]) : (mxw ? -3.0e38f : 0.0f);
                v[k] = (double)x;
            }
            const bool is_max_lane_hi = part == 3u;        // words 24..31: all maxima (31 unused)
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const double o = __shfl_xor(v[k], off, 64);
                    // part 2 holds words 16..23: 16, 17, 18 are sums, 19..23 maxima
                    const bool mx = is_max_lane_hi || (part == 2u && k >= 3);
                    v[k] = mx ? fmax(v[k], o) : v[k] + o;
                }
            }
            if (lane < 4u) {
#pragma unroll
                for (int k = 0; k < 8; ++k) wtot[wave * 32u + lane * 8u + k] = v[k];
            }
            __syncthreads();
            if (wave == 0 && lane < 31u) {   // totals over the 16 waves, in wave order: lane k owns word k
                double a = wtot[lane];
                const bool mx = lane >= 19u;
#pragma unroll 1
                for (uint32_t wv = 1; wv < GR_RES_WAVES; ++wv) { const double o = wtot[wv * 32u + lane]; a = mx ? fmax(a, o) : a + o; }
                wtot[GR_RES_WAVES * 32u + lane] = a;
            }
            gr_wave_sync();
            if (wave == 0 && lane == 0) {
                GrFrameState &st = state[f];
                if (pre_status == 0 && polls != 0xFFFFFFFFu) {
                    const double *t = wtot + GR_RES_WAVES * 32u;
                    double acc[GR_ACC_K];
#pragma unroll
                    for (int k = 0; k < GR_ACC_K; ++k) acc[k] = 0.0;
#pragma unroll
                    for (int k = 0; k < 13; ++k) acc[k] = t[k];
#pragma unroll
                    for (int k = 0; k < 6; ++k) acc[26 + k] = t[13 + k];
                    const float mn[3] = { -(float)t[19], -(float)t[20], -(float)t[21] }, mx3[3] = { (float)t[22], (float)t[23], (float)t[24] };
                    const float fmn[3] = { -(float)t[25], -(float)t[26], -(float)t[27] }, fmx[3] = { (float)t[28], (float)t[29], (float)t[30] };
                    const double g[3] = { g0x, g0y, g0z };
#ifdef GR_DBG_NOMATH
                    st.R[0] = st.R[4] = st.R[8] = 1.0f; st.shift[0] = (float)(acc[1] * 1e-30 + g[0] * 1e-30 + mn[0] * 1e-30f + fmx[2] * 1e-30f);   // experiment: no closing algebra
#else
                    gr_finalize_math<0, true, false>(acc, mn, mx3, fmn, fmx, GR_NOIDX, GR_NOIDX, lb, plan, g, sel.n, st);
#endif
                }
                unsigned long long *o = ctl.rec + (size_t)f * 16;
                // (a finalizer that gave up publishes a failed frame: the streaming waves leave it unmodified and move on)
                gr_st_agent(o + 0, tagv | (uint32_t)(polls != 0xFFFFFFFFu ? st.status : 1));
#pragma unroll
                for (int k = 0; k < 3; ++k) gr_st_agent(o + 1 + k, tagv | __float_as_uint(st.shift[k]));
#pragma unroll
                for (int k = 0; k < 9; ++k) gr_st_agent(o + 4 + k, tagv | __float_as_uint(st.R[k]));
            }
            __syncthreads();                                          // wtot is free again
#ifdef GR_DBG_RES
            if (blockIdx.x == ctl.n_stream && tid == 0) {
                const unsigned long long tf2 = wall_clock64();
                if (f >= 512 && f < 560) printf("[fin0] frame %u: waited for records %.2f us (%u polls), reduce + close + publish %.2f us; records complete at %.2f\n", f, (tf1 - tf0) * 0.01, polls, (tf2 - tf1) * 0.01, (tf1 % 10000000ull) * 0.01);
            }
#endif
        }
        return;
    }

    // ------------------------------------------------------------------------------------------ streaming workgroups
    const uint32_t ngroups = ((n_atoms + 255u) >> 8) << 6;            // the slot is padded to whole tiles
    const uint32_t wg = blockIdx.x, g = wg * GR_RES_LANES + tid;
    if (wg * GR_RES_LANES + wave * 64u >= ngroups) return;            // a whole wave behind the last tile (ngroups is a multiple of 64)
    const uint32_t n_waves = min((uint32_t)GR_RES_WAVES, (ngroups - wg * GR_RES_LANES) >> 6);
    float4 *park = smem;
    float *wsum = reinterpret_cast<float *>(smem + GR_RES_PARK_F4);
    double *fsum = reinterpret_cast<double *>(wsum + GR_RES_WSUM_F);
    uint32_t *cnt_s = reinterpret_cast<uint32_t *>(fsum + GR_RES_R * GR_RES_WAVES), *cnt_f = cnt_s + GR_RES_R;
    if (tid < 2 * GR_RES_R) cnt_s[tid] = 0u;
    __syncthreads();                                                  // the only barrier: before the first frame

    const uint32_t first = sel.start, last = sel.start + sel.n, g0 = sel.g0 << 6;
    const uint32_t i0 = g << 2;
    const bool in_sel = (i0 + 3u >= first) && (i0 < last);
    const bool full = (i0 >= first) && (i0 + 3u < last);
    const bool in0 = (i0 >= first) && (i0 < last), in1 = (i0 + 1u >= first) && (i0 + 1u < last), in2 = (i0 + 2u >= first) && (i0 + 2u < last),
               in3 = (i0 + 3u >= first) && (i0 + 3u < last);
    const size_t b = gr_row_index(g, 0);
    // the lane's reference coordinates, masses and weights: loaded once, registers for the whole launch
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pa = zero4, pb = zero4, pc = zero4, mm = zero4, ww = zero4;
    if (in_sel) {
        gr_rows_load(reinterpret_cast<const float4 *>(plan.p), (size_t)(g - g0), pa, pb, pc);
        mm = reinterpret_cast<const float4 *>(masses)[g];
        if (!WMASS) ww = reinterpret_cast<const float4 *>(plan.w)[g - g0];
        if (!full) {   // ragged end of the selection: atoms outside it weigh nothing and have no reference
            if (!in0) { mm.x = 0.f; ww.x = 0.f; pa.x = 0.f; pa.z = 0.f; pb.x = 0.f; }
            if (!in1) { mm.y = 0.f; ww.y = 0.f; pa.y = 0.f; pa.w = 0.f; pb.y = 0.f; }
            if (!in2) { mm.z = 0.f; ww.z = 0.f; pb.z = 0.f; pc.x = 0.f; pc.z = 0.f; }
            if (!in3) { mm.w = 0.f; ww.w = 0.f; pb.w = 0.f; pc.y = 0.f; pc.w = 0.f; }
        }
    }
    const GrP4 P = gr_pairs_rows(pa, pb, pc);
    const float cx = plan.ref_com[0], cy = plan.ref_com[1], cz = plan.ref_com[2];

    auto request = [&](uint32_t f, float4 &r0, float4 &r1, float4 &r2, float &gx, float &gy, float &gz) {
        const float *xyz = frames + (size_t)(first_slot + f) * frame_stride;
        const float4 *f4 = reinterpret_cast<const float4 *>(xyz);
        r0 = gr_stream_load(f4 + b); r1 = gr_stream_load(f4 + b + 64); r2 = gr_stream_load(f4 + b + 128);
        gr_pos_load(xyz, first, gx, gy, gz);                          // provisional centre: the first atom of the selection
    };
    auto request_rec = [&](uint32_t f) -> unsigned long long { return lane < 13u ? gr_ld_agent(ctl.rec + (size_t)f * 16 + lane) : 0ull; };
    bool bail = false;
#ifdef GR_DBG_RES
    unsigned long long dbg_wait = 0, dbg_polls = 0, dbg_rows = 0; const unsigned long long dbg_t0 = wall_clock64();
#endif

    // ---- the sums stage of frame i: rows r*, provisional centre (gx, gy, gz)
    auto sums = [&](uint32_t i, const float4 &r0, const float4 &r1, const float4 &r2, float gxv, float gyv, float gzv) {
        const GrBox *boxp = boxes + first_slot + i;
        const GrBoxU B = gr_box_uniform(boxp);
        const float gx = gr_first_f(gxv), gy = gr_first_f(gyv), gz = gr_first_f(gzv);    // wave-uniform: SGPR operands
        const uint32_t ps = i % GR_RES_K;
#ifdef GR_DBG_RES
        { const unsigned long long tr0 = wall_clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_rows += wall_clock64() - tr0; }
#endif
        park[(ps * 3 + 0) * GR_RES_LANES + tid] = r0; park[(ps * 3 + 1) * GR_RES_LANES + tid] = r1; park[(ps * 3 + 2) * GR_RES_LANES + tid] = r2;
        GrP4 q = gr_pairs_rows(r0, r1, r2);
        if (!full) {   // atoms outside the selection become copies of the first atom: v = 0 adds nothing and lies inside every extent
            if (!in0) { q.x01.x = gx; q.y01.x = gy; q.z01.x = gz; }
            if (!in1) { q.x01.y = gx; q.y01.y = gy; q.z01.y = gz; }
            if (!in2) { q.x23.x = gx; q.y23.x = gy; q.z23.x = gz; }
            if (!in3) { q.x23.y = gx; q.y23.y = gy; q.z23.y = gz; }
        }
        // the 19 sums and 12 extents of the lane's four atoms (gr_sums_pair's arithmetic, folded at once: nothing accumulates)
        gr_v2f vxa = q.x01 - gr_v2(gx), vya = q.y01 - gr_v2(gy), vza = q.z01 - gr_v2(gz);
        gr_v2f vxb = q.x23 - gr_v2(gx), vyb = q.y23 - gr_v2(gy), vzb = q.z23 - gr_v2(gz);
        gr_image_pair(vxa, vya, vza, B, boxp);
        gr_image_pair(vxb, vyb, vzb, B, boxp);
        auto fold = [](gr_v2f v) { return v.x + v.y; };
        float s32[32], e32[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) { s32[k] = 0.0f; e32[k] = -3.0e38f; }
        {
            const gr_v2f ma = gr_v2p(mm.x, mm.y), mb = gr_v2p(mm.z, mm.w);
            s32[0] = fold(ma + mb);
            s32[1] = fold(gr_v2_fma(ma, vxa, mb * vxb)); s32[2] = fold(gr_v2_fma(ma, vya, mb * vyb)); s32[3] = fold(gr_v2_fma(ma, vza, mb * vzb));
            s32[4] = fold(gr_v2_fma(P.x01, vxa, P.x23 * vxb)); s32[5] = fold(gr_v2_fma(P.x01, vya, P.x23 * vyb)); s32[6] = fold(gr_v2_fma(P.x01, vza, P.x23 * vzb));
            s32[7] = fold(gr_v2_fma(P.y01, vxa, P.y23 * vxb)); s32[8] = fold(gr_v2_fma(P.y01, vya, P.y23 * vyb)); s32[9] = fold(gr_v2_fma(P.y01, vza, P.y23 * vzb));
            s32[10] = fold(gr_v2_fma(P.z01, vxa, P.z23 * vxb)); s32[11] = fold(gr_v2_fma(P.z01, vya, P.z23 * vyb)); s32[12] = fold(gr_v2_fma(P.z01, vza, P.z23 * vzb));
            e32[0] = -gr_fminf(gr_min3f(vxa.x, vxa.y, vxb.x), vxb.y); e32[1] = -gr_fminf(gr_min3f(vya.x, vya.y, vyb.x), vyb.y); e32[2] = -gr_fminf(gr_min3f(vza.x, vza.y, vzb.x), vzb.y);
            e32[3] = gr_fmaxf(gr_max3f(vxa.x, vxa.y, vxb.x), vxb.y); e32[4] = gr_fmaxf(gr_max3f(vya.x, vya.y, vyb.x), vyb.y); e32[5] = gr_fmaxf(gr_max3f(vza.x, vza.y, vzb.x), vzb.y);
            // fractional coordinates of v: moments + extents feed the image proof (gr_finalize_math)
            const gr_v2f fca = vza * gr_v2(B.icz), fcb = vzb * gr_v2(B.icz);
            const gr_v2f fba = gr_v2_fma(-fca, gr_v2(B.cy), vya) * gr_v2(B.iby), fbb = gr_v2_fma(-fcb, gr_v2(B.cy), vyb) * gr_v2(B.iby);
            const gr_v2f faa = gr_v2_fma(-fca, gr_v2(B.cx), gr_v2_fma(-fba, gr_v2(B.bx), vxa)) * gr_v2(B.iax), fab = gr_v2_fma(-fcb, gr_v2(B.cx), gr_v2_fma(-fbb, gr_v2(B.bx), vxb)) * gr_v2(B.iax);
            s32[13] = fold(faa + fab); s32[14] = fold(fba + fbb); s32[15] = fold(fca + fcb);
            s32[16] = fold(gr_v2_fma(faa, faa, fab * fab)); s32[17] = fold(gr_v2_fma(fba, fba, fbb * fbb)); s32[18] = fold(gr_v2_fma(fca, fca, fcb * fcb));
            e32[6] = -gr_fminf(gr_min3f(faa.x, faa.y, fab.x), fab.y); e32[7] = -gr_fminf(gr_min3f(fba.x, fba.y, fbb.x), fbb.y); e32[8] = -gr_fminf(gr_min3f(fca.x, fca.y, fcb.x), fcb.y);
            e32[9] = gr_fmaxf(gr_max3f(faa.x, faa.y, fab.x), fab.y); e32[10] = gr_fmaxf(gr_max3f(fba.x, fba.y, fbb.x), fbb.y); e32[11] = gr_fmaxf(gr_max3f(fca.x, fca.y, fcb.x), fcb.y);
        }
        const float tot = gr_wave_sum_scatter32(s32, lane);
        const float emax = gr_wave_max_scatter16(e32, lane);
        const uint32_t rs = i % GR_RES_R;
#ifdef GR_DBG_RES
        if (lane == 0 && (i == 512 || i == 520) && (wave == 0 || wave == 15) && (wg % 61 == 0)) printf("[deliver] frame %u wg %u wave %u at %.2f\n", i, wg, wave, (wall_clock64() % 10000000ull) * 0.01);
#endif
        float *mine = wsum + (rs * GR_RES_WAVES + wave) * 48;
        if ((lane & 1u) == 0) mine[lane >> 1] = tot;
        if ((lane & 3u) == 0) mine[32 + (lane >> 2)] = emax;
        gr_wave_sync();
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt_s + rs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)old) != n_waves - 1u) return;
        // this wave completed the workgroup's record of frame i: add the wave records in wave order, publish 31 tagged words
        gr_wave_sync();
        const float *all = wsum + rs * GR_RES_WAVES * 48;
        if (lane < 31u) {
            const uint32_t src = lane < 19u ? lane : 32u + (lane - 19u);
            float v = all[src];
            if (lane < 19u) { for (uint32_t w = 1; w < n_waves; ++w) v += all[w * 48 + src]; }
            else { for (uint32_t w = 1; w < n_waves; ++w) v = gr_fmaxf(v, all[w * 48 + src]); }
            const uint32_t n_pad = (ctl.n_stream + 15u) & ~15u;
            gr_st_agent(ctl.wgrec + ((size_t)i * n_pad + wg) * GR_RES_REC_WORDS + lane, ((unsigned long long)ctl.epoch << 32) | __float_as_uint(v));
        }
        if (lane == 0) __hip_atomic_store(cnt_s + rs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // ---- the fit stage of frame j: `rv` = the frame's record as requested earlier (lanes 0..12)
    auto fit = [&](uint32_t j, unsigned long long rv) {
        uint32_t polls = 0;
#ifdef GR_DBG_RES
        const unsigned long long tp0 = wall_clock64();
#endif
#ifdef GR_DBG_NOSYNC
        rv = (lane == 4u || lane == 8u || lane == 12u) ? (unsigned long long)__float_as_uint(1.0f) : 0ull;     // experiment: no wait, R = 1, shift = 0
#else
        while (__builtin_amdgcn_ballot_w64(lane < 13u && (uint32_t)(rv >> 32) != ctl.epoch) != 0ull) {
            if (++polls > GR_RES_PATIENCE || ((polls & 255u) == 0 && gr_ld_agent(ctl.abort) != 0u)) { if (lane == 0) gr_st_agent(ctl.abort, 1u); bail = true; return; }
            __builtin_amdgcn_s_sleep(16);
            rv = request_rec(j);
        }
#endif
#ifdef GR_DBG_RES
        dbg_wait += wall_clock64() - tp0; dbg_polls += polls;
#endif
        const int status = __builtin_amdgcn_readlane((int)(uint32_t)rv, 0);
        double rs = 0.0;
        if (status == 0) {
            const float sx = gr_lane_f(rv, 1), sy = gr_lane_f(rv, 2), sz = gr_lane_f(rv, 3);
            const float r00 = gr_lane_f(rv, 4), r10 = gr_lane_f(rv, 5), r20 = gr_lane_f(rv, 6), r01 = gr_lane_f(rv, 7), r11 = gr_lane_f(rv, 8), r21 = gr_lane_f(rv, 9),
                        r02 = gr_lane_f(rv, 10), r12 = gr_lane_f(rv, 11), r22 = gr_lane_f(rv, 12);
            const GrBox *boxp = boxes + first_slot + j;
            const GrBoxU B = gr_box_uniform(boxp);
            const uint32_t ps = j % GR_RES_K;
            const float4 r0 = park[(ps * 3 + 0) * GR_RES_LANES + tid], r1 = park[(ps * 3 + 1) * GR_RES_LANES + tid], r2 = park[(ps * 3 + 2) * GR_RES_LANES + tid];
            GrP4 q = gr_pairs_rows(r0, r1, r2);
            q.x01 += gr_v2(sx); q.y01 += gr_v2(sy); q.z01 += gr_v2(sz); q.x23 += gr_v2(sx); q.y23 += gr_v2(sy); q.z23 += gr_v2(sz);
            gr_wrap_pair_fast(q.x01, q.y01, q.z01, B);
            gr_wrap_pair_fast(q.x23, q.y23, q.z23, B);
            {
                const float xl = gr_fminf(gr_min3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y), xh = gr_fmaxf(gr_max3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y);
                const float yl = gr_fminf(gr_min3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y), yh = gr_fmaxf(gr_max3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y);
                const float zl = gr_fminf(gr_min3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y), zh = gr_fmaxf(gr_max3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y);
                const bool ok = (xl > 0.0f) & (xh <= B.ax) & (yl > 0.0f) & (yh <= B.by) & (zl > 0.0f) & (zh <= B.cz);
                if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {   // an atom on a face / farther than one cell / without position: the general wrap
                    float x[4], y[4], z[4];
                    gr_rows_unpack(r0, r1, r2, x, y, z);
                    float x0 = x[0] + sx, y0 = y[0] + sy, z0 = z[0] + sz, x1 = x[1] + sx, y1 = y[1] + sy, z1 = z[1] + sz;
                    float x2 = x[2] + sx, y2 = y[2] + sy, z2 = z[2] + sz, x3 = x[3] + sx, y3 = y[3] + sy, z3 = z[3] + sz;
                    gr_wrap(x0, y0, z0, *boxp); gr_wrap(x1, y1, z1, *boxp); gr_wrap(x2, y2, z2, *boxp); gr_wrap(x3, y3, z3, *boxp);
                    q.x01 = gr_v2p(x0, x1); q.x23 = gr_v2p(x2, x3); q.y01 = gr_v2p(y0, y1); q.y23 = gr_v2p(y2, y3); q.z01 = gr_v2p(z0, z1); q.z23 = gr_v2p(z2, z3);
                }
            }
            q.x01 -= gr_v2(B.bcx); q.y01 -= gr_v2(B.bcy); q.z01 -= gr_v2(B.bcz); q.x23 -= gr_v2(B.bcx); q.y23 -= gr_v2(B.bcy); q.z23 -= gr_v2(B.bcz);
            GrP4 n;
            n.x01 = gr_v2_fma(gr_v2(r02), q.z01, gr_v2_fma(gr_v2(r01), q.y01, gr_v2(r00) * q.x01));
            n.y01 = gr_v2_fma(gr_v2(r12), q.z01, gr_v2_fma(gr_v2(r11), q.y01, gr_v2(r10) * q.x01));
            n.z01 = gr_v2_fma(gr_v2(r22), q.z01, gr_v2_fma(gr_v2(r21), q.y01, gr_v2(r20) * q.x01));
            n.x23 = gr_v2_fma(gr_v2(r02), q.z23, gr_v2_fma(gr_v2(r01), q.y23, gr_v2(r00) * q.x23));
            n.y23 = gr_v2_fma(gr_v2(r12), q.z23, gr_v2_fma(gr_v2(r11), q.y23, gr_v2(r10) * q.x23));
            n.z23 = gr_v2_fma(gr_v2(r22), q.z23, gr_v2_fma(gr_v2(r21), q.y23, gr_v2(r20) * q.x23));
            if (in_sel) {   // sum w |R q - p|^2 (rmsd.rs:592-599); the weights of atoms outside the selection are zero
                const gr_v2f w01 = WMASS ? gr_v2p(mm.x, mm.y) : gr_v2p(ww.x, ww.y), w23 = WMASS ? gr_v2p(mm.z, mm.w) : gr_v2p(ww.z, ww.w);
                gr_v2f dx = n.x01 - P.x01, dy = n.y01 - P.y01, dz = n.z01 - P.z01;
                gr_v2f part = w01 * gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz));
                dx = n.x23 - P.x23; dy = n.y23 - P.y23; dz = n.z23 - P.z23;
                part = gr_v2_fma(w23, gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz)), part);
                rs = (double)(part.x + part.y);
            }
            n.x01 += gr_v2(cx); n.y01 += gr_v2(cy); n.z01 += gr_v2(cz); n.x23 += gr_v2(cx); n.y23 += gr_v2(cy); n.z23 += gr_v2(cz);
            float4 o0, o1, o2;
            gr_rows_pairs(n, o0, o1, o2);
            float4 *f4 = reinterpret_cast<float4 *>(frames + (size_t)(first_slot + j) * frame_stride);
            gr_stream_store(f4 + b, o0); gr_stream_store(f4 + b + 64, o1); gr_stream_store(f4 + b + 128, o2);
        }
        // the workgroup's share of sum w |R q - p|^2: wave sums in wave order by the last wave to arrive
        rs = gr_wave_sum(rs);
        const uint32_t fs = j % GR_RES_R;
        if (lane == 0) fsum[fs * GR_RES_WAVES + wave] = rs;
        gr_wave_sync();
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt_f + fs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)old) != n_waves - 1u) return;
        gr_wave_sync();
        if (lane == 0) {
            double t = 0.0;
            for (uint32_t w = 0; w < n_waves; ++w) t += fsum[fs * GR_RES_WAVES + w];
            fit_partials[(size_t)j * ctl.n_stream + wg] = t;
            __hip_atomic_store(cnt_f + fs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    // ---- the walk: iteration i = fit of frame i - K, then sums of frame i; rows one frame ahead in two named register sets
    float4 a0 = zero4, a1 = zero4, a2 = zero4, b0 = zero4, b1 = zero4, b2 = zero4;
    float agx = 0.f, agy = 0.f, agz = 0.f, bgx = 0.f, bgy = 0.f, bgz = 0.f;
    unsigned long long rv = 0ull;
    request(0, a0, a1, a2, agx, agy, agz);
    const uint32_t n_iter = nframes + GR_RES_K;
    for (uint32_t i = 0; i < n_iter; i += 2) {
        // even iteration: frame i is in set a, frame i + 1 goes to set b
        if (i + 1 < nframes) request(i + 1, b0, b1, b2, bgx, bgy, bgz);
        if (i >= GR_RES_K) { fit(i - GR_RES_K, rv); if (bail) return; }
        if (i + 1 >= GR_RES_K && i + 1 < n_iter) rv = request_rec(i + 1 - GR_RES_K);
        if (i < nframes) sums(i, a0, a1, a2, agx, agy, agz);
        if (i + 1 >= n_iter) break;
        // odd iteration: frame i + 1 is in set b, frame i + 2 goes to set a
        if (i + 2 < nframes) request(i + 2, a0, a1, a2, agx, agy, agz);
        if (i + 1 >= GR_RES_K) { fit(i + 1 - GR_RES_K, rv); if (bail) return; }
        if (i + 2 >= GR_RES_K && i + 2 < n_iter) rv = request_rec(i + 2 - GR_RES_K);
        if (i + 1 < nframes) sums(i + 1, b0, b1, b2, bgx, bgy, bgz);
    }
#ifdef GR_DBG_RES
    if (lane == 0 && ((wg == 7 && wave == 3) || (wg == 100 && wave == 9) || (wg == 200 && wave == 15)))
        printf("[wg %u w %u] %u frames in %.1f us: %.2f us/frame; waiting for records %.2f us/frame (%.2f polls/frame), waiting for rows %.2f us/frame\n", wg, wave, nframes,
               (wall_clock64() - dbg_t0) * 0.01, (wall_clock64() - dbg_t0) * 0.01 / nframes, dbg_wait * 0.01 / nframes, (double)dbg_polls / nframes, dbg_rows * 0.01 / nframes);
#endif
}
