// gr_resident.h -- the RMSD fit as ONE pass over HBM when a frame fits on the chip.
//
// The two-pass path (gr_hot.h) reads every frame twice: once for the sums that give the rotation, once to apply it -- the
// rotation of a frame depends on all of its atoms, so no atom can be written before every atom has been read.  36 bytes per
// atom and frame cross HBM (12 + 12 read, 12 written), plus 32 bytes per atom and frame of reference coordinates and
// weights from the L2 / Infinity Cache.  But a 1e6-atom frame is 12 MB, and the chip has 256 CUs x 512 KiB of vector
// registers + 160 KiB of LDS: the frame can WAIT ON CHIP for its rotation.
//
//   k_fit_resident   one launch for a whole batch: `n_stream` workgroups of 512 lanes, TWO 4-atom groups per lane
//                    for the whole launch (its reference coordinates, masses and weights are loaded once and stay in
//                    registers); a lane walks the frames of the batch with its own groups:
//                       sums stage, frame i      rows arrive (requested one frame earlier), image about the first atom, the
//                                                18 sums + 12 extents of its 8 atoms -> wave reduce-scatter -> LDS; the last
//                                                wave of the workgroup adds the 8 wave records in wave order and writes the
//                                                workgroup's record: 31 tagged words (value | epoch << 32), no flag, no fence;
//                                                the rows are parked: group A in LDS (24 KiB per frame and CU), group B in
//                                                registers (12 per frame)
//                       fit stage, frame i - 6   the frame's record (status, shift, R) has come back from a finalizer ->
//                                                rows out of LDS / the register queue, wrap + rotate + translate, sum w |R q - p|^2, one
//                                                non-temporal store per row
//                    + `n_fin` workgroups that stream nothing: finalizer j owns the frames j, j + n_fin, ...; its 8 waves
//                    read 32 workgroup records each (one load round trip for the whole frame), re-reading until every
//                    word carries the launch's tag; a fixed tree adds them in fp64, one lane closes the frame exactly as
//                    the two-pass path does (gr_finalize_math: image proof, Kabsch rotation in fp64) and publishes
//                    status | shift | R as 13 tagged 64-bit words.  The frame's box, first atom and host-side status are
//                    requested BEFORE the wait: the finalize is the latency the parked frames have to cover.
// HBM traffic: 12 bytes per atom read + 12 written per frame = 24 (was 36), and nothing from the caches.
// Measured floor of that traffic at the same launch shape (tools/ceiling_bench.hip "resident copy"): 4.2 us per 1e6-atom frame.
//
// STATUS (round 2, MI355X, 1e6 atoms, 768-1024 frames per launch): 6.26 us per frame = 156-157 k frames/s with two groups per lane,
// against 6.6-6.8 us = 148-151 k for the two-pass path in the same jobs: the pass is the DEFAULT for frames that fill at least
// 15/16 of the chip (GR_TUNE_RESIDENT = 1; every CU runs its 4096 atoms' worth of a frame or idles, so a smaller frame is
// better off with the two passes, whose time shrinks with it).  What bounds it is not memory (a lane waits 0.2-0.3 us per frame
// for its rows; 24 MB per frame cross HBM where the floor of that traffic is 4.2 us) and not the
// finalizers (no closing algebra at all: same time; 8 or 11 of them: same time) but the instruction streams themselves: every CU
// runs ALL of a frame's arithmetic for its 4096 atoms within one frame period -- ~900 VALU + ~220 scalar instructions per wave
// and frame (sums + the per-frame wave reduction + fit + the queue of parked register sets), two waves per SIMD that overlap
// poorly: the VALU is busy 45-50 % of the time; the last wave of every workgroup (second on its SIMD, and the one that adds up
// the workgroup's record) never waits for a record and sets the pace, the others wait for it a third of their time.  The two-pass
// kernels run 20 % fewer VALU instructions per frame in total and keep 12-16 waves per CU in flight.
// History of the measurement (us per frame): 1024 lanes x 1 group, three frames parked, records collected through seven
// dependent round trips 9.1; tagged records read in one round trip, lane-swap reductions 9.1 (by then the spread between the
// four waves of a SIMD was the longer pole); 512 lanes x 2 groups, four parked 8.4; six parked, register sets named by unrolling
// x6 7.6 (the loop body, with the rare paths inlined in every copy, was ~140 KB of code against a 64 KB instruction cache); a
// queue of register sets + the rare paths out of line 7.3; lane facts as bits of one register, waiting waves at low priority
// 7.2; wave records of 32 floats and the parking moved out of the reduction lambdas 6.6; reductions on DPP moves instead of
// ds_bpermute 6.5; an ordinary launch instead of hipLaunchCooperativeKernel 6.35; the per-frame sums as a 16-wide scatter + two
// plain wave sums (the mass sum does not depend on the frame) 6.26.  Frames parked: 4 -> 7.0, 5 -> 6.6, 6 -> 6.5
// (before the last step; LDS holds no more).  One group per lane (1024 lanes, five frames
// parked, 128 registers): 9.3 -- the register budget spills into scratch memory inside the loop.
//
// Synchronisation.  All waiting is on data that a DIFFERENT workgroup produces, so every workgroup must become resident: the
// pass is only chosen when the grid fits the device with one workgroup per CU (occupancy query at context creation), the host
// lets one such launch run per device and process at a time (a second one would share the CUs with the first and both could
// starve; the loser takes the two-pass path), kernels of other streams that hold CUs when it starts end on their own, and the
// kernel opens with a START HANDSHAKE: every workgroup checks in and the last one opens the launch; when that does not happen
// within ~0.2 s the launch closes itself and every workgroup leaves before any frame is touched (the host then runs the
// segment on the two-pass path).  Past the handshake every workgroup is on the chip and no wait can last.  The
// launch is an ordinary one: hipLaunchCooperativeKernel -- the runtime's own co-residency check -- makes rocprofv3 --pmc fault
// and crashed a process that issued it from two host threads at exit (ROCm 7.2).  Every wait is bounded (GR_RES_PATIENCE polls with s_sleep, a few
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

// Shape: G = 4-atom groups per lane.  A workgroup always owns 1024 groups (4096 atoms), so it has 1024 / G lanes:
//   G = 2   512 lanes, 8 waves (two per SIMD, up to 256 registers); group A of a frame waits in LDS, group B in a queue of
//           register sets; 6 frames parked.  Halves the number of per-frame wave reductions.
//   G = 1   1024 lanes, 16 waves (four per SIMD, 128 registers): more waves to cover one another's latencies; a frame waits
//           first in a short queue of register sets, then in one of three LDS slots.
// (The first version -- G = 1 with three LDS slots only, records collected through seven dependent round trips -- ran at 9.1 us.)
template <int G> struct GrResShape {
    static constexpr int LANES = 1024 / G, WAVES = LANES / 64;
    static constexpr int KL = G == 2 ? 6 : 3;        // LDS slots (a slot = three rows of one group for every lane: 48 KiB / G ... x KL = 144 KiB)
    static constexpr int KV = G == 2 ? 6 : 2;        // register sets in the queue (G = 2: group B for its whole wait; G = 1: the first KV frames of the wait)
    static constexpr int K = G == 2 ? 6 : KL + KV;   // frames between the sums stage and the fit stage
    static constexpr int R = G == 2 ? 8 : 6;         // ring of wave-record slots ( > K: no wave is more than K frames ahead of another)
    static constexpr int PARK_F4 = KL * 3 * LANES;   // float4
    static constexpr int WSUM_F = R * WAVES * 32;    // float: a wave record = 19 sums + 12 extents
    static constexpr int LDS_BYTES = PARK_F4 * 16 + WSUM_F * 4 + R * WAVES * 8 + 2 * R * 4;
    static constexpr int REC_PER_WAVE = 256 / WAVES, LANES_PER_REC = 64 / REC_PER_WAVE, WORDS_PER_LANE = 32 / LANES_PER_REC;   // finalizer
};
#ifndef GR_RES_SLEEP
#define GR_RES_SLEEP 8              // s_sleep argument between two looks at a record that is not there yet (x 64 clocks)
#endif
#ifndef GR_RES_PRIO
#define GR_RES_PRIO 1
#endif
#define GR_RES_GROUPS 1024         // 4-atom groups per workgroup
#define GR_RES_MAX_FIN 8
#define GR_RES_PATIENCE 3000000u   // polls (each ~1 us) before a wait gives up
#define GR_RES_START_PATIENCE 200000u   // polls of the start handshake (~0.2 s: other kernels may hold CUs when the launch begins)

#define GR_RES_REC_WORDS 32         // tagged words per workgroup record: 0..18 sums, 19..30 extents (as maxima), 31 unused
#define GR_RES_REC_PAD 32           // workgroup records per frame are padded to a multiple of this
struct GrResCtl {
    unsigned long long *wgrec;     // [frames][n_stream padded][32] value | epoch << 32
    unsigned long long *rec;       // [frames][16] value | epoch << 32: 0 status, 1..3 shift, 4..12 R (column-major)
    uint32_t *abort;               // [3]: 0 abort (0 = fine), 1 workgroups that have started, 2 start verdict (0 open, 1 go, 2 never started);
                                   // words 1 and 2 are zeroed by the host before every launch
    uint32_t epoch, n_stream, n_fin;
};

template <typename T> __device__ __forceinline__ T gr_ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float gr_first_f(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
__device__ __forceinline__ float gr_lane_f(unsigned long long v, int l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l)); }

// one 4-atom group of a lane: which of its atoms belong to the selection, its reference rows, masses and weights (registers)
// (the per-lane facts are bits of ONE register: seven `bool`s would be seven 64-bit lane masks -- 14 SGPRs per group held
// across the whole loop, and the kernel was spilling SGPRs into vector lanes)
enum { GR_RG_IN0 = 1u, GR_RG_IN1 = 2u, GR_RG_IN2 = 4u, GR_RG_IN3 = 8u, GR_RG_ANY = 16u /* some atom in the selection */, GR_RG_FULL = 32u /* all four */ };
struct GrResGroup {
    bool valid;        // wave-uniform: the group lies inside the slot
    uint32_t flags;
    size_t b;          // float4 index of the group's first row inside a slot
    GrP4 P;
    float4 mm, ww;
};

// frame state as the fit stage needs it (wave-uniform: SGPRs)
struct GrResRot { float sx, sy, sz, r00, r10, r20, r01, r11, r21, r02, r12, r22; };

// The rare paths are real function calls: inlined, every copy of the streaming loop's body carried four general wraps and four
// image-table searches per group, and the loop outgrew the instruction cache.
// (arguments and results by value: a reference parameter would pin the caller's registers to scratch memory on the hot path)
__device__ __attribute__((noinline)) GrP4 gr_res_wrap_slow(float4 r0, float4 r1, float4 r2, float sx, float sy, float sz, const GrBox *__restrict__ boxp) {
    GrP4 q;
    float x[4], y[4], z[4];
    gr_rows_unpack(r0, r1, r2, x, y, z);
#pragma unroll 1
    for (int k = 0; k < 4; ++k) { x[k] += sx; y[k] += sy; z[k] += sz; gr_wrap(x[k], y[k], z[k], *boxp); }
    q.x01 = gr_v2p(x[0], x[1]); q.x23 = gr_v2p(x[2], x[3]); q.y01 = gr_v2p(y[0], y[1]); q.y23 = gr_v2p(y[2], y[3]); q.z01 = gr_v2p(z[0], z[1]); q.z23 = gr_v2p(z[2], z[3]);
    return q;
}
struct GrV6 { gr_v2f x, y, z; };
__device__ __attribute__((noinline)) GrV6 gr_res_refine_slow(gr_v2f vx, gr_v2f vy, gr_v2f vz, float rws2, const GrBox *__restrict__ boxp) {
    const gr_v2f r2 = gr_v2_fma(vx, vx, gr_v2_fma(vy, vy, vz * vz));
    if (!(r2.x < rws2)) { float a = vx.x, b = vy.x, c = vz.x; gr_tric_refine(a, b, c, *boxp); vx.x = a; vy.x = b; vz.x = c; }
    if (!(r2.y < rws2)) { float a = vx.y, b = vy.y, c = vz.y; gr_tric_refine(a, b, c, *boxp); vx.y = a; vy.y = b; vz.y = c; }
    GrV6 o; o.x = vx; o.y = vy; o.z = vz;
    return o;
}
// gr_image_pair with the image-table search out of line
__device__ __forceinline__ void gr_res_image_pair(gr_v2f &vx, gr_v2f &vy, gr_v2f &vz, const GrBoxU &B, const GrBox *__restrict__ boxp) {
    gr_v2f k = gr_v2_rint(vz * gr_v2(B.icz));
    vx = gr_v2_fma(-k, gr_v2(B.cx), vx); vy = gr_v2_fma(-k, gr_v2(B.cy), vy); vz = gr_v2_fma(-k, gr_v2(B.cz), vz);
    k = gr_v2_rint(vy * gr_v2(B.iby));
    vx = gr_v2_fma(-k, gr_v2(B.bx), vx); vy = gr_v2_fma(-k, gr_v2(B.by), vy);
    k = gr_v2_rint(vx * gr_v2(B.iax));
    vx = gr_v2_fma(-k, gr_v2(B.ax), vx);
    if (B.tric) {
        const gr_v2f r2 = gr_v2_fma(vx, vx, gr_v2_fma(vy, vy, vz * vz));
        if (__builtin_amdgcn_ballot_w64(!(gr_fmaxf(r2.x, r2.y) < B.rws2)) != 0ull) { const GrV6 o = gr_res_refine_slow(vx, vy, vz, B.rws2, boxp); vx = o.x; vy = o.y; vz = o.z; }
    }
}

// wrap(x + shift) - box centre, rotate, (sum w |R q - p|^2), + reference COM: the arithmetic of k_fit_pk for one group
template <bool WMASS>
__device__ __forceinline__ void gr_res_fit_group(const GrResGroup &G, const float4 &r0, const float4 &r1, const float4 &r2, const GrResRot &T, const GrBoxU &B,
                                                 const GrBox *__restrict__ boxp, float cx, float cy, float cz, float4 *__restrict__ f4, float &rs) {
    GrP4 q = gr_pairs_rows(r0, r1, r2);
    q.x01 += gr_v2(T.sx); q.y01 += gr_v2(T.sy); q.z01 += gr_v2(T.sz); q.x23 += gr_v2(T.sx); q.y23 += gr_v2(T.sy); q.z23 += gr_v2(T.sz);
    gr_wrap_pair_fast(q.x01, q.y01, q.z01, B);
    gr_wrap_pair_fast(q.x23, q.y23, q.z23, B);
    {
        const float xl = gr_fminf(gr_min3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y), xh = gr_fmaxf(gr_max3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y);
        const float yl = gr_fminf(gr_min3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y), yh = gr_fmaxf(gr_max3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y);
        const float zl = gr_fminf(gr_min3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y), zh = gr_fmaxf(gr_max3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y);
        const bool ok = (xl > 0.0f) & (xh <= B.ax) & (yl > 0.0f) & (yh <= B.by) & (zl > 0.0f) & (zh <= B.cz);
        if (__builtin_amdgcn_ballot_w64(!ok) != 0ull)   // an atom on a face / farther than one cell / without position: the general wrap
            q = gr_res_wrap_slow(r0, r1, r2, T.sx, T.sy, T.sz, boxp);
    }
    q.x01 -= gr_v2(B.bcx); q.y01 -= gr_v2(B.bcy); q.z01 -= gr_v2(B.bcz); q.x23 -= gr_v2(B.bcx); q.y23 -= gr_v2(B.bcy); q.z23 -= gr_v2(B.bcz);
    GrP4 n;
    n.x01 = gr_v2_fma(gr_v2(T.r02), q.z01, gr_v2_fma(gr_v2(T.r01), q.y01, gr_v2(T.r00) * q.x01));
    n.y01 = gr_v2_fma(gr_v2(T.r12), q.z01, gr_v2_fma(gr_v2(T.r11), q.y01, gr_v2(T.r10) * q.x01));
    n.z01 = gr_v2_fma(gr_v2(T.r22), q.z01, gr_v2_fma(gr_v2(T.r21), q.y01, gr_v2(T.r20) * q.x01));
    n.x23 = gr_v2_fma(gr_v2(T.r02), q.z23, gr_v2_fma(gr_v2(T.r01), q.y23, gr_v2(T.r00) * q.x23));
    n.y23 = gr_v2_fma(gr_v2(T.r12), q.z23, gr_v2_fma(gr_v2(T.r11), q.y23, gr_v2(T.r10) * q.x23));
    n.z23 = gr_v2_fma(gr_v2(T.r22), q.z23, gr_v2_fma(gr_v2(T.r21), q.y23, gr_v2(T.r20) * q.x23));
    if (G.flags & GR_RG_ANY) {   // sum w |R q - p|^2 (rmsd.rs:592-599); the weights of atoms outside the selection are zero
        const gr_v2f w01 = WMASS ? gr_v2p(G.mm.x, G.mm.y) : gr_v2p(G.ww.x, G.ww.y), w23 = WMASS ? gr_v2p(G.mm.z, G.mm.w) : gr_v2p(G.ww.z, G.ww.w);
        gr_v2f dx = n.x01 - G.P.x01, dy = n.y01 - G.P.y01, dz = n.z01 - G.P.z01;
        gr_v2f part = w01 * gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz));
        dx = n.x23 - G.P.x23; dy = n.y23 - G.P.y23; dz = n.z23 - G.P.z23;
        part = gr_v2_fma(w23, gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz)), part);
        rs += part.x + part.y;
    }
    n.x01 += gr_v2(cx); n.y01 += gr_v2(cy); n.z01 += gr_v2(cz); n.x23 += gr_v2(cx); n.y23 += gr_v2(cy); n.z23 += gr_v2(cz);
    float4 o0, o1, o2;
    gr_rows_pairs(n, o0, o1, o2);
    gr_stream_store(f4 + G.b, o0); gr_stream_store(f4 + G.b + 64, o1); gr_stream_store(f4 + G.b + 128, o2);
}

// UBOX: every frame of the launch has the same box (the host compared them): its constants are loaded once, not per frame
template <bool WMASS, bool UBOX, int G>
__global__ __launch_bounds__(GrResShape<G>::LANES) void k_fit_resident(
    float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, uint32_t nframes, uint32_t n_atoms,
    const float *__restrict__ masses, GrSel sel, const GrBox *__restrict__ boxes, GrPlanDev plan,
    GrFrameState *state, double *__restrict__ fit_partials, GrResCtl ctl) {
    typedef GrResShape<G> S;
    constexpr uint32_t LANES = S::LANES, WAVES = S::WAVES, K = S::K, KL = S::KL, KV = S::KV, R = S::R;
    extern __shared__ float4 smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_pad = (ctl.n_stream + GR_RES_REC_PAD - 1u) & ~(uint32_t)(GR_RES_REC_PAD - 1u);

    // ------------------------------------------------------------------------------------------ start handshake
    // Every workgroup waits for data other workgroups produce, so nothing may begin before ALL of them are on the chip: each one
    // checks in, the one that completes the count opens the launch (verdict 1).  A workgroup that waits too long -- the device is
    // shared with another process's kernels that will not leave, or two of these launches hold half the chip each -- closes it
    // (verdict 2) and everybody leaves WITHOUT having touched a frame: the host then runs the segment on the two-pass path.
    {
        __shared__ uint32_t verdict;
        if (tid == 0) {
            const uint32_t n = __hip_atomic_fetch_add(ctl.abort + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            uint32_t zero = 0u;
            if (n == gridDim.x) (void)__hip_atomic_compare_exchange_strong(ctl.abort + 2, &zero, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t v, polls = 0;
            while ((v = gr_ld_agent(ctl.abort + 2)) == 0u) {
                if (++polls > GR_RES_START_PATIENCE) { zero = 0u; (void)__hip_atomic_compare_exchange_strong(ctl.abort + 2, &zero, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                __builtin_amdgcn_s_sleep(16);
            }
            verdict = v;
        }
        __syncthreads();
        if (verdict != 1u) return;
    }

    // ------------------------------------------------------------------------------------------ finalizers
    if (blockIdx.x >= ctl.n_stream) {
        constexpr uint32_t RPW = S::REC_PER_WAVE, LPR = S::LANES_PER_REC, W = S::WORDS_PER_LANE;
        double *wtot = reinterpret_cast<double *>(smem);              // [WAVES][32] + [32] totals
        const uint32_t r = wave * RPW + lane / LPR, part = lane % LPR;   // this lane's record and its W words
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
            const unsigned long long *src = ctl.wgrec + ((size_t)f * n_pad + r) * GR_RES_REC_WORDS + part * W;
            unsigned long long w[W];
            uint32_t polls = 0;
            for (;;) {
                bool ok = true;
                if (r < ctl.n_stream) {
#pragma unroll
                    for (uint32_t k = 0; k < W; ++k) w[k] = gr_ld_agent(src + k);
#pragma unroll
                    for (uint32_t k = 0; k < W; ++k) ok = ok && ((uint32_t)(w[k] >> 32) == ctl.epoch || part * W + k == 31u);
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (++polls > GR_RES_PATIENCE || ((polls & 255u) == 0 && gr_ld_agent(ctl.abort) != 0u)) { if (lane == 0) gr_st_agent(ctl.abort, 1u); polls = 0xFFFFFFFFu; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            // (a wave that gave up still meets the others at the barriers; the abort word ends the launch)
            // sums in fp64, extents as maxima: word index = part * W + k; 0..18 sums, 19..30 maxima
            double v[W];
#pragma unroll
            for (uint32_t k = 0; k < W; ++k) {
                const bool mxw = part * W + k >= 19u;
                const float x = (r < ctl.n_stream && polls != 0xFFFFFFFFu) ? __uint_as_float((uint32_t)w[k]) : (mxw ? -3.0e38f : 0.0f);
                v[k] = (double)x;
            }
#pragma unroll
            for (uint32_t off = LPR; off < 64u; off <<= 1) {
#pragma unroll
                for (uint32_t k = 0; k < W; ++k) {
                    const double o = __shfl_xor(v[k], (int)off, 64);
                    const bool mxw = part * W + k >= 19u;
                    v[k] = mxw ? fmax(v[k], o) : v[k] + o;
                }
            }
            if (lane < LPR) {
#pragma unroll
                for (uint32_t k = 0; k < W; ++k) wtot[wave * 32u + lane * W + k] = v[k];
            }
            __syncthreads();
            if (wave == 0 && lane < 31u) {   // totals over the waves, in wave order: lane k owns word k
                double a = wtot[lane];
                const bool mx = lane >= 19u;
#pragma unroll
                for (uint32_t wv = 1; wv < WAVES; ++wv) { const double o = wtot[wv * 32u + lane]; a = mx ? fmax(a, o) : a + o; }
                wtot[WAVES * 32u + lane] = a;
            }
            gr_wave_sync();
            if (wave == 0 && lane == 0) {
                GrFrameState &st = state[f];
                if (pre_status == 0 && polls != 0xFFFFFFFFu) {
                    const double *t = wtot + WAVES * 32u;
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
                    gr_finalize_math<0, true, false>(acc, mn, mx3, fmn, fmx, GR_NOIDX, GR_NOIDX, lb, plan, g, sel.n, st);
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
        }
        return;
    }

    // ------------------------------------------------------------------------------------------ streaming workgroups
    const uint32_t ngroups = ((n_atoms + 255u) >> 8) << 6;            // the slot is padded to whole tiles (a multiple of 64 groups)
    const uint32_t wg = blockIdx.x, base = wg * GR_RES_GROUPS;
    if (base + wave * 64u >= ngroups) return;                         // every chunk of this wave lies behind the last tile
    const uint32_t n_waves = min(WAVES, (ngroups - base) >> 6);
    float4 *park = smem;
    float *wsum = reinterpret_cast<float *>(smem + S::PARK_F4);
    double *fsum = reinterpret_cast<double *>(wsum + S::WSUM_F);
    uint32_t *cnt_s = reinterpret_cast<uint32_t *>(fsum + R * WAVES), *cnt_f = cnt_s + R;
    if (tid < 2 * R) cnt_s[tid] = 0u;
    __syncthreads();                                                  // the only barrier: before the first frame

    const uint32_t first = sel.start, last = sel.start + sel.n, g0 = sel.g0 << 6;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // the lane's groups: A = chunk `wave` of the workgroup's first LANES groups, B (G = 2) = the same chunk of its second LANES
    auto setup = [&](uint32_t g, GrResGroup &Gr) {
        const uint32_t i0 = g << 2;
        Gr.valid = g < ngroups;                                       // wave-uniform
        const bool in0 = Gr.valid && (i0 >= first) && (i0 < last), in1 = Gr.valid && (i0 + 1u >= first) && (i0 + 1u < last);
        const bool in2 = Gr.valid && (i0 + 2u >= first) && (i0 + 2u < last), in3 = Gr.valid && (i0 + 3u >= first) && (i0 + 3u < last);
        const bool in_sel = in0 || in1 || in2 || in3, full = in0 && in1 && in2 && in3;
        Gr.flags = (in0 ? GR_RG_IN0 : 0u) | (in1 ? GR_RG_IN1 : 0u) | (in2 ? GR_RG_IN2 : 0u) | (in3 ? GR_RG_IN3 : 0u) | (in_sel ? GR_RG_ANY : 0u) | (full ? GR_RG_FULL : 0u);
        Gr.b = gr_row_index(Gr.valid ? g : 0u, 0);
        float4 pa = zero4, pb = zero4, pc = zero4;
        Gr.mm = zero4; Gr.ww = zero4;
        if (in_sel) {
            gr_rows_load(reinterpret_cast<const float4 *>(plan.p), (size_t)(g - g0), pa, pb, pc);
            Gr.mm = reinterpret_cast<const float4 *>(masses)[g];
            if (!WMASS) Gr.ww = reinterpret_cast<const float4 *>(plan.w)[g - g0];
            if (!full) {   // ragged end of the selection: atoms outside it weigh nothing and have no reference
                if (!in0) { Gr.mm.x = 0.f; Gr.ww.x = 0.f; pa.x = 0.f; pa.z = 0.f; pb.x = 0.f; }
                if (!in1) { Gr.mm.y = 0.f; Gr.ww.y = 0.f; pa.y = 0.f; pa.w = 0.f; pb.y = 0.f; }
                if (!in2) { Gr.mm.z = 0.f; Gr.ww.z = 0.f; pb.z = 0.f; pc.x = 0.f; pc.z = 0.f; }
                if (!in3) { Gr.mm.w = 0.f; Gr.ww.w = 0.f; pb.w = 0.f; pc.y = 0.f; pc.w = 0.f; }
            }
        }
        Gr.P = gr_pairs_rows(pa, pb, pc);
    };
    GrResGroup GA, GB;
    setup(base + tid, GA);
    if (G == 2) setup(base + LANES + tid, GB); else { GB.valid = false; GB.flags = 0u; GB.b = 0; GB.mm = GB.ww = zero4; GB.P = gr_pairs_rows(zero4, zero4, zero4); }
    const float cx = plan.ref_com[0], cy = plan.ref_com[1], cz = plan.ref_com[2];
    // sum of the masses of the wave's atoms inside the selection: the same for every frame of the launch
    const float m_wave = gr_wave_allsum_f32(((GA.mm.x + GA.mm.y) + (GA.mm.z + GA.mm.w)) + ((GB.mm.x + GB.mm.y) + (GB.mm.z + GB.mm.w)));

    struct Rows { float4 r0, r1, r2; };
    struct Landing { Rows a, b; float gx, gy, gz; };
    auto request = [&](uint32_t f, Landing &L) {
        const float *xyz = frames + (size_t)(first_slot + f) * frame_stride;
        const float4 *f4 = reinterpret_cast<const float4 *>(xyz);
        L.a.r0 = gr_stream_load(f4 + GA.b); L.a.r1 = gr_stream_load(f4 + GA.b + 64); L.a.r2 = gr_stream_load(f4 + GA.b + 128);
        if (G == 2 && GB.valid) { L.b.r0 = gr_stream_load(f4 + GB.b); L.b.r1 = gr_stream_load(f4 + GB.b + 64); L.b.r2 = gr_stream_load(f4 + GB.b + 128); }
        gr_pos_load(xyz, first, L.gx, L.gy, L.gz);                    // provisional centre: the first atom of the selection
    };
    auto request_rec = [&](uint32_t f) -> unsigned long long { return lane < 13u ? gr_ld_agent(ctl.rec + (size_t)f * 16 + lane) : 0ull; };
    bool bail = false;

    // the sums of one group added to the lane's 19 + 12 values
    auto group_sums = [&](const GrResGroup &Gr, const Rows &rw, const GrBoxU &B, const GrBox *boxp,
                          float gx, float gy, float gz, float (&s32)[32], float (&e32)[32], bool init) {
        GrP4 q = gr_pairs_rows(rw.r0, rw.r1, rw.r2);
        // atoms outside the selection become copies of the first atom: v = 0 adds nothing and lies inside every extent (a whole
        // wave of complete groups -- every wave but the two at the ends of the selection -- skips this on one scalar branch)
        if (__builtin_amdgcn_ballot_w64((Gr.flags & GR_RG_FULL) == 0u) != 0ull) {
            if (!(Gr.flags & GR_RG_IN0)) { q.x01.x = gx; q.y01.x = gy; q.z01.x = gz; }
            if (!(Gr.flags & GR_RG_IN1)) { q.x01.y = gx; q.y01.y = gy; q.z01.y = gz; }
            if (!(Gr.flags & GR_RG_IN2)) { q.x23.x = gx; q.y23.x = gy; q.z23.x = gz; }
            if (!(Gr.flags & GR_RG_IN3)) { q.x23.y = gx; q.y23.y = gy; q.z23.y = gz; }
        }
        gr_v2f vxa = q.x01 - gr_v2(gx), vya = q.y01 - gr_v2(gy), vza = q.z01 - gr_v2(gz);
        gr_v2f vxb = q.x23 - gr_v2(gx), vyb = q.y23 - gr_v2(gy), vzb = q.z23 - gr_v2(gz);
        gr_res_image_pair(vxa, vya, vza, B, boxp);
        gr_res_image_pair(vxb, vyb, vzb, B, boxp);
        auto fold = [](gr_v2f v) { return v.x + v.y; };
        auto acc = [&](int k, float x) { s32[k - 1] = init ? x : s32[k - 1] + x; };   // value k (1..18) lives at s32[k - 1]; value 0 = sum m is not per frame
        auto ext = [&](int k, float x) { e32[k] = init ? x : gr_fmaxf(e32[k], x); };
        const gr_v2f ma = gr_v2p(Gr.mm.x, Gr.mm.y), mb = gr_v2p(Gr.mm.z, Gr.mm.w);
        acc(1, fold(gr_v2_fma(ma, vxa, mb * vxb))); acc(2, fold(gr_v2_fma(ma, vya, mb * vyb))); acc(3, fold(gr_v2_fma(ma, vza, mb * vzb)));
        acc(4, fold(gr_v2_fma(Gr.P.x01, vxa, Gr.P.x23 * vxb))); acc(5, fold(gr_v2_fma(Gr.P.x01, vya, Gr.P.x23 * vyb))); acc(6, fold(gr_v2_fma(Gr.P.x01, vza, Gr.P.x23 * vzb)));
        acc(7, fold(gr_v2_fma(Gr.P.y01, vxa, Gr.P.y23 * vxb))); acc(8, fold(gr_v2_fma(Gr.P.y01, vya, Gr.P.y23 * vyb))); acc(9, fold(gr_v2_fma(Gr.P.y01, vza, Gr.P.y23 * vzb)));
        acc(10, fold(gr_v2_fma(Gr.P.z01, vxa, Gr.P.z23 * vxb))); acc(11, fold(gr_v2_fma(Gr.P.z01, vya, Gr.P.z23 * vyb))); acc(12, fold(gr_v2_fma(Gr.P.z01, vza, Gr.P.z23 * vzb)));
        ext(0, -gr_fminf(gr_min3f(vxa.x, vxa.y, vxb.x), vxb.y)); ext(1, -gr_fminf(gr_min3f(vya.x, vya.y, vyb.x), vyb.y)); ext(2, -gr_fminf(gr_min3f(vza.x, vza.y, vzb.x), vzb.y));
        ext(3, gr_fmaxf(gr_max3f(vxa.x, vxa.y, vxb.x), vxb.y)); ext(4, gr_fmaxf(gr_max3f(vya.x, vya.y, vyb.x), vyb.y)); ext(5, gr_fmaxf(gr_max3f(vza.x, vza.y, vzb.x), vzb.y));
        // fractional coordinates of v: moments + extents feed the image proof (gr_finalize_math)
        const gr_v2f fca = vza * gr_v2(B.icz), fcb = vzb * gr_v2(B.icz);
        const gr_v2f fba = gr_v2_fma(-fca, gr_v2(B.cy), vya) * gr_v2(B.iby), fbb = gr_v2_fma(-fcb, gr_v2(B.cy), vyb) * gr_v2(B.iby);
        const gr_v2f faa = gr_v2_fma(-fca, gr_v2(B.cx), gr_v2_fma(-fba, gr_v2(B.bx), vxa)) * gr_v2(B.iax), fab = gr_v2_fma(-fcb, gr_v2(B.cx), gr_v2_fma(-fbb, gr_v2(B.bx), vxb)) * gr_v2(B.iax);
        acc(13, fold(faa + fab)); acc(14, fold(fba + fbb)); acc(15, fold(fca + fcb));
        acc(16, fold(gr_v2_fma(faa, faa, fab * fab))); acc(17, fold(gr_v2_fma(fba, fba, fbb * fbb))); acc(18, fold(gr_v2_fma(fca, fca, fcb * fcb)));
        ext(6, -gr_fminf(gr_min3f(faa.x, faa.y, fab.x), fab.y)); ext(7, -gr_fminf(gr_min3f(fba.x, fba.y, fbb.x), fbb.y)); ext(8, -gr_fminf(gr_min3f(fca.x, fca.y, fcb.x), fcb.y));
        ext(9, gr_fmaxf(gr_max3f(faa.x, faa.y, fab.x), fab.y)); ext(10, gr_fmaxf(gr_max3f(fba.x, fba.y, fbb.x), fbb.y)); ext(11, gr_fmaxf(gr_max3f(fca.x, fca.y, fcb.x), fcb.y));
    };

    // ---- the sums stage of frame i (rows in L): the lane's 19 + 12 values -> wave (reduce-scatter) -> workgroup (LDS, the last
    // wave to arrive adds the wave records in wave order) -> the frame's tagged record
    auto sums = [&](uint32_t i, const Landing &L, const GrBoxU &B) {
        float s32[32], e32[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) { s32[k] = 0.0f; e32[k] = -3.0e38f; }
        {
            const GrBox *boxp = boxes + first_slot + i;
            const float gx = gr_first_f(L.gx), gy = gr_first_f(L.gy), gz = gr_first_f(L.gz);    // wave-uniform: SGPR operands
            group_sums(GA, L.a, B, boxp, gx, gy, gz, s32, e32, true);
            if (G == 2 && GB.valid) group_sums(GB, L.b, B, boxp, gx, gy, gz, s32, e32, false);
        }
        // 18 sums per frame (the 19th, sum m, does not depend on the frame: m_wave below) = a reduce-scatter of 16 + two plain wave
        // sums (a 32-wide scatter would push 14 zeros through its two widest steps); 12 extents = a 16-wide scatter with max
        const float tot = gr_wave_sum_scatter16(s32, lane);                       // values 1..16
        const float t17 = gr_wave_allsum_f32(s32[16]), t18 = gr_wave_allsum_f32(s32[17]);
        const float emax = gr_wave_max_scatter16(e32, lane);
        const uint32_t rs = i % R;
        float *mine = wsum + (rs * WAVES + wave) * 32;
        if ((lane & 3u) == 0) mine[1 + (lane >> 2)] = tot;                         // sums 1..16
        if (lane == 1u) { mine[0] = m_wave; mine[17] = t17; mine[18] = t18; }      // sum m (the same every frame), sums 17, 18
        if ((lane & 3u) == 0 && lane < 48u) mine[19 + (lane >> 2)] = emax;         // extents 0..11
        gr_wave_sync();
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt_s + rs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)old) != n_waves - 1u) return;
        // this wave completed the workgroup's record of frame i: add the wave records in wave order, publish 31 tagged words
        gr_wave_sync();
        const float *all = wsum + rs * WAVES * 32;
        if (lane < 31u) {
            float v = all[lane];
            if (lane < 19u) { for (uint32_t w = 1; w < n_waves; ++w) v += all[w * 32 + lane]; }
            else { for (uint32_t w = 1; w < n_waves; ++w) v = gr_fmaxf(v, all[w * 32 + lane]); }
            gr_st_agent(ctl.wgrec + ((size_t)i * n_pad + wg) * GR_RES_REC_WORDS + lane, ((unsigned long long)ctl.epoch << 32) | __float_as_uint(v));
        }
        if (lane == 0) __hip_atomic_store(cnt_s + rs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // ---- the fit stage of frame j: `rv` = the frame's record as requested earlier (lanes 0..12); rows as they were parked.
    auto fit = [&](uint32_t j, unsigned long long rv, const Rows &ra, const Rows &rb, const GrBoxU &B) {
        uint32_t polls = 0;
        while (__builtin_amdgcn_ballot_w64(lane < 13u && (uint32_t)(rv >> 32) != ctl.epoch) != 0ull) {
#if GR_RES_PRIO
            __builtin_amdgcn_s_setprio(0);                             // a wave that is ahead waits below the waves it shares the SIMD with
#endif
            if (++polls > GR_RES_PATIENCE || ((polls & 255u) == 0 && gr_ld_agent(ctl.abort) != 0u)) { if (lane == 0) gr_st_agent(ctl.abort, 1u); bail = true; return; }
            __builtin_amdgcn_s_sleep(GR_RES_SLEEP);
            rv = request_rec(j);
        }
#if GR_RES_PRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        const int status = __builtin_amdgcn_readlane((int)(uint32_t)rv, 0);
        float rs = 0.0f;
        if (status == 0) {
            GrResRot T;
            T.sx = gr_lane_f(rv, 1); T.sy = gr_lane_f(rv, 2); T.sz = gr_lane_f(rv, 3);
            T.r00 = gr_lane_f(rv, 4); T.r10 = gr_lane_f(rv, 5); T.r20 = gr_lane_f(rv, 6); T.r01 = gr_lane_f(rv, 7); T.r11 = gr_lane_f(rv, 8); T.r21 = gr_lane_f(rv, 9);
            T.r02 = gr_lane_f(rv, 10); T.r12 = gr_lane_f(rv, 11); T.r22 = gr_lane_f(rv, 12);
            const GrBox *boxp = boxes + first_slot + j;
            float4 *f4 = reinterpret_cast<float4 *>(frames + (size_t)(first_slot + j) * frame_stride);
            gr_res_fit_group<WMASS>(GA, ra.r0, ra.r1, ra.r2, T, B, boxp, cx, cy, cz, f4, rs);
            if (G == 2 && GB.valid) gr_res_fit_group<WMASS>(GB, rb.r0, rb.r1, rb.r2, T, B, boxp, cx, cy, cz, f4, rs);
        }
        // the workgroup's share of sum w |R q - p|^2: the lane's eight atoms in f32, the wave in f32 (no LDS crossbar), waves in fp64
        // in wave order by the last wave to arrive
        const float wtot = gr_wave_allsum_f32(rs);
        const uint32_t fs = j % R;
        if (lane == 0) fsum[fs * WAVES + wave] = (double)wtot;
        gr_wave_sync();
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt_f + fs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)old) != n_waves - 1u) return;
        gr_wave_sync();
        if (lane == 0) {
            double t = 0.0;
            for (uint32_t w = 0; w < n_waves; ++w) t += fsum[fs * WAVES + w];
            fit_partials[(size_t)j * ctl.n_stream + wg] = t;
            __hip_atomic_store(cnt_f + fs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    // ---- the walk: iteration i = fit of frame i - K, then sums of frame i.  Where a frame waits:
    //   G = 2   group A in LDS slot i % K for the whole wait, group B in the queue of register sets (Q[0] = the oldest);
    //   G = 1   the first KV iterations in the queue, then KL iterations in LDS slot (i - KV) % KL  (K = KV + KL).
    // The queue moves up by one set per iteration (register moves); naming the sets by i % K instead means unrolling the loop K
    // times, and six copies of this body were far larger than the 64 KiB instruction cache.
    Landing L0, L1;
    L0.a.r0 = L0.a.r1 = L0.a.r2 = L0.b.r0 = L0.b.r1 = L0.b.r2 = zero4; L0.gx = L0.gy = L0.gz = 0.f;
    L1 = L0;
    Rows Q[KV];
#pragma unroll
    for (uint32_t u = 0; u < KV; ++u) Q[u].r0 = Q[u].r1 = Q[u].r2 = zero4;
    unsigned long long rv = 0ull;
    request(0, L0);
    const uint32_t n_iter = nframes + K;
    const GrBoxU B0 = gr_box_uniform(boxes + first_slot);
    auto lds_put = [&](uint32_t slot, const Rows &rw) { park[(slot * 3 + 0) * LANES + tid] = rw.r0; park[(slot * 3 + 1) * LANES + tid] = rw.r1; park[(slot * 3 + 2) * LANES + tid] = rw.r2; };
    auto lds_get = [&](uint32_t slot) { Rows rw; rw.r0 = park[(slot * 3 + 0) * LANES + tid]; rw.r1 = park[(slot * 3 + 1) * LANES + tid]; rw.r2 = park[(slot * 3 + 2) * LANES + tid]; return rw; };
    auto step = [&](uint32_t i, Landing &cur, Landing &nxt) {
        if (i + 1 < nframes) request(i + 1, nxt);
        // both boxes of the iteration are requested here (scalar loads): they arrive while the record is checked
        const GrBoxU Bf = UBOX ? B0 : gr_box_uniform(boxes + first_slot + (i >= K ? i - K : 0u));
        const GrBoxU Bs = UBOX ? B0 : gr_box_uniform(boxes + first_slot + (i < nframes ? i : 0u));
        if (G == 2) {
            const uint32_t ps = i % K;                         // LDS slot: frame i - K leaves it, frame i takes it
            if (i >= K) { fit(i - K, rv, lds_get(ps), Q[0], Bf); if (bail) return; }
            if (i + 1 >= K && i + 1 < n_iter) rv = request_rec(i + 1 - K);
#pragma unroll
            for (uint32_t u = 0; u + 1 < KV; ++u) Q[u] = Q[u + 1];
            if (i < nframes) { lds_put(ps, cur.a); Q[KV - 1] = cur.b; sums(i, cur, Bs); }
        } else {
            const uint32_t pl = (i + KL - (K % KL)) % KL;       // LDS slot of frame i - K = the slot frame i - KV is about to take
            if (i >= K) { fit(i - K, rv, lds_get(pl), cur.b, Bf); if (bail) return; }
            if (i + 1 >= K && i + 1 < n_iter) rv = request_rec(i + 1 - K);
            if (i >= KV && i - KV < nframes) lds_put(pl, Q[0]); // frame i - KV moves from the queue into LDS
#pragma unroll
            for (uint32_t u = 0; u + 1 < KV; ++u) Q[u] = Q[u + 1];
            if (i < nframes) { Q[KV - 1] = cur.a; sums(i, cur, Bs); }
        }
    };
    for (uint32_t i = 0; i < n_iter; i += 2) {
        step(i, L0, L1);
        if (bail) return;
        if (i + 1 < n_iter) step(i + 1, L1, L0);
        if (bail) return;
    }
}
