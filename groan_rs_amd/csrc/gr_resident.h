// gr_resident.h -- the RMSD fit as ONE pass over HBM when a frame (or several frames side by side) fits on the chip.
//
// The two-pass path (gr_hot.h) reads every frame twice: once for the sums that give the rotation, once to apply it -- the
// rotation of a frame depends on all of its atoms, so no atom can be written before every atom has been read.  36 bytes per
// atom and frame cross HBM (12 + 12 read, 12 written), plus 32 bytes per atom and frame of reference coordinates and
// weights from the L2 / Infinity Cache.  But a 1e6-atom frame is 12 MB, and the chip has 256 CUs x 512 KiB of vector
// registers + 160 KiB of LDS: the frame can WAIT ON CHIP for its rotation.
//
//   k_fit_resident   one launch for a whole batch: `n_stream` workgroups of 512 lanes (8 waves, two per SIMD), TWO 4-atom groups
//                    per lane for the whole launch (their reference coordinates, masses and weights are loaded once and stay in
//                    registers); a lane walks the frames of the batch with its own groups:
//                       sums stage, frame i      rows arrive (requested one frame earlier), image v of every atom about the first
//                                                atom of the selection, the 18 sums + 12 extents of the lane's 8 atoms -> wave
//                                                reduce-scatter -> LDS; a wave that is AHEAD (the first to pass its next hand-over
//                                                and find the count complete: see `claim`) adds the 8 wave records in wave order and
//                                                writes the workgroup's record: 31 tagged words (value | epoch << 32), no flag, no
//                                                fence between workgroups; the frame is parked: group A in LDS
//                                                (24 KiB per frame and CU), group B in one of six register sets (12 registers)
//                       fit stage, frame i - 6   the frame's record (status, shift, R, t0) has come back from a finalizer ->
//                                                the parked data out of LDS / the register set, rotate + translate,
//                                                sum w |R q - p|^2 (one fp64 word per wave and frame, added up by k_rmsd_close),
//                                                one non-temporal store per row
//                    + `n_fin` workgroups that stream nothing: finalizer j owns the frames j, j + n_fin, ...; its 8 waves
//                    read 32 workgroup records each (one load round trip for the whole frame), re-reading until every
//                    word carries the launch's tag; a fixed tree adds them in fp64, one lane closes the frame exactly as
//                    the two-pass path does (gr_finalize_math: image proof, Kabsch rotation in fp64) and publishes
//                    status | shift | R | t0 as 16 tagged 64-bit words.  The frame's box, first atom and host-side status are
//                    requested BEFORE the wait: the finalize is the latency the parked frames have to cover.
//                    FRAME STREAMS.  A frame that fills a fraction of the chip does not get the launch to itself: S = 2 .. 32
//                    streams of `wgs_frame` workgroups run side by side, stream s walking the frames s, s + S, ... of the batch
//                    (workgroup b: stream b / wgs_frame, atoms of workgroup b % wgs_frame).  Each stream is the pipeline above on
//                    its own share of the CUs; nothing couples them but the finalizers, which then close several frames at a
//                    time: a frame of <= 32 / 64 / 128 workgroups needs only 1 / 2 / 4 of a finalizer's waves for its records,
//                    so the 8 waves form TEAMS that close 8 / 4 / 2 consecutive frames together (never more frames than there
//                    are streams: the frames of a round must be frames that become ready together).  The closing arithmetic
//                    is one lane's chain of fp64 operations, ~15 us per frame: 8 finalizer workgroups x 1 frame capped the
//                    launch at 0.54 M frames/s whatever the size of the frame; teams lift that to ~4 M.  Measured against the
//                    two-pass path (profiles/r03_size_sweep.txt): + 39 % at 500 000 atoms (2 streams), + 51 % at 250 000 (4),
//                    + 53 % at 125 000 (8), + 57 % at 62 000 (15).
// HBM traffic: 12 bytes per atom read + 12 written per frame = 24 (two passes: 36), and nothing from the caches.
// Measured floor of that traffic at the same launch shape (tools/ceiling_bench.hip "resident copy"): 4.2 us per 1e6-atom frame.
//
// Two kernels, chosen by the host:
//   V = true   the selection is the WHOLE system (the all-atom fit of the benchmark).  What is parked is not the frame's rows x but
//              the image vectors v = image of (x - first atom) that the sums stage computes anyway.  Once the frame's image proof
//              holds (gr_finalize_math: every atom inside the brick about the COM with 1e-3 nm to spare), the path's
//              q = wrap(x + shift) - box centre (rmsd.rs:479-492) IS v - cv (cv = COM - first atom), so the fit stage is
//                  R q = R v + t0,   t0 = -R cv (published by the finalizer),   z = R q + reference COM      (rmsd.rs:508-528)
//              -- 18 packed FMAs per group instead of shift + three-stage wrap + range check + centre + rotation.
//   V = false  any other contiguous selection: atoms outside it have no proven image, so rows are parked and the fit stage is the
//              literal arithmetic of k_fit_pk (wrap with its general fall-back); waves without any atom of the selection skip
//              the sums arithmetic.
//
// What set the pace in round 2 (6.26 us per frame) was not memory and not the finalizers but how the two waves of a SIMD shared
// it: the hardware lets the OLDER wave issue first, so the older wave of each pair ran ahead until it had to wait for a record,
// and the younger one -- whose sums every record needs -- was left the slots the older one did not use; on top of that the wave
// that arrived last combined the workgroup's record through a chain of dependent LDS reads.  Round 3 (us per frame, 1e6 atoms,
// 768 frames per launch): each wave publishes its progress and takes PRIORITY WHEN IT IS BEHIND ITS SIMD PARTNER 5.44; the
// combine as independent loads 5.30; register sets picked by frame % 6 instead of a queue that moves up every iteration 5.27;
// image vectors parked (V) + the sums of both groups in one chain of four FMAs ... (see DESIGN.md for the current figure).
// History of round 2: 1024 lanes x 1 group, three frames parked 9.1; tagged records read in one round trip, lane-swap reductions
// 9.1; 512 lanes x 2 groups, four parked 8.4; six parked, register sets named by unrolling x6 7.6 (the loop body did not fit the
// instruction cache); a queue of register sets + the rare paths out of line 7.3; lane facts as bits of one register, waiting waves
// at low priority 7.2; wave records of 32 floats 6.6; reductions on DPP moves instead of ds_bpermute 6.5; an ordinary launch
// instead of hipLaunchCooperativeKernel 6.35; the per-frame sums as a 16-wide scatter + two plain wave sums 6.26.  The one-group
// shape (1024 lanes, 128 registers, 9.3 us: spills inside the loop) was removed in round 3.
//
// Synchronisation.  All waiting is on data that a DIFFERENT workgroup produces, so every workgroup must become resident: the
// pass is only chosen when the grid (streams x workgroups per frame + finalizers) fits the device with one workgroup per CU (occupancy query at context creation), the host
// lets one such launch run per device and process at a time (a second one would share the CUs with the first and both could
// starve; the loser takes the two-pass path), kernels of other streams that hold CUs when it starts end on their own, and the
// kernel opens with a START HANDSHAKE: every workgroup checks in and the last one opens the launch; when that does not happen
// within ~0.2 s the launch closes itself and every workgroup leaves before any frame is touched (the host then runs the
// segment on the two-pass path).  Past the handshake every workgroup is on the chip and no wait can last.  The
// launch is an ordinary one: hipLaunchCooperativeKernel -- the runtime's own co-residency check -- makes rocprofv3 --pmc fault
// and crashed a process that issued it from two host threads at exit (ROCm 7.2).  Every wait is bounded (ctl.patience_ticks of the device clock, GR_RES_PATIENCE polls
// with s_sleep, a few seconds): a wave that runs out of patience raises `abort` and leaves, every other wait then ends too and
// the grid drains.  Every streaming wave records how many frames it has fitted when it leaves (`progress`): after an abort the
// host knows which frames are complete, which are untouched (those it redoes on the two-pass path) and which -- if a wave gave
// up in the instant its record arrived for the others -- are torn.
// Between workgroups: every word carries the launch's epoch in its upper half and is written / read as ONE 64-bit access at
// agent scope (bypassing the non-coherent caches): a reader can never take a word of an older launch for a new one, nothing has
// to be cleared between launches, and no store has to be ordered against another -- which keeps "s_waitcnt vmcnt(0)" (a drain
// of the wave's prefetched rows) out of the streaming loop.
// Inside a workgroup (no barrier in the loop): a wave writes its record to LDS, RELEASES it with a workgroup-scope fence
// restricted to the LDS address space (one s_waitcnt lgkmcnt(0); the unrestricted fence would drain the prefetched rows too)
// and bumps the slot's LDS counter; the wave that later claims the complete frame ACQUIRES with the same kind of fence before it
// reads the waves' records.  Results do not depend on the order of arrival or on who claims: the wave records are added in wave
// order, the workgroup records by a fixed tree in the finalizer.
// Precision: a workgroup record is the f32 sum of 16 f32 wave sums (4096 atoms; the two-pass path sums ~7800 atoms in f32
// before it widens), the finalizer adds the workgroup records in fp64.
#pragma once
#include <type_traits>
#include "gr_hot.h"

// Shape: a workgroup owns 1024 4-atom groups (4096 atoms) = 512 lanes x 2 groups; 8 waves, two per SIMD, up to 256 registers.
struct GrResShape {
    static constexpr int LANES = 512, WAVES = LANES / 64;
#ifndef GR_RES_K
#define GR_RES_K 6
#endif
    static constexpr int K = GR_RES_K;               // frames between the sums stage and the fit stage = LDS slots = register sets
    static constexpr int R = 8;                      // ring of wave-record slots ( > K: no wave is more than K frames ahead of another)
    static constexpr int PARK_F4 = K * 3 * LANES;    // float4: a slot = three rows of group A for every lane (24 KiB) x K = 144 KiB
    static constexpr int WSUM_F = R * WAVES * 32;    // float: a wave record = 19 sums + 12 extents
#ifdef GR_EXP_TIMELINE
    static constexpr int TL_BYTES = 128 * 4 * 8;     // experiment: stamps of workgroup 0's last 128 turns (published / first look / polls / fit done)
#else
    static constexpr int TL_BYTES = 0;
#endif
    static constexpr int LDS_BYTES = PARK_F4 * 16 + WSUM_F * 4 + R * WAVES * 8 + 3 * R * 4 + 2 * WAVES * 4 + TL_BYTES;   // ... + fit sums, counters (arrivals at the sums / the fit hand-over, combines done), progress and SIMD of every wave
    static constexpr int REC_PER_WAVE = 256 / WAVES, LANES_PER_REC = 64 / REC_PER_WAVE, WORDS_PER_LANE = 32 / LANES_PER_REC;   // finalizer
};
#ifndef GR_RES_SLEEP
#define GR_RES_SLEEP 2              // s_sleep argument between two looks at a record that is not there yet (x 64 clocks)
#endif
#ifndef GR_RES_BAL
#define GR_RES_BAL 1               // priority by progress relative to the wave's SIMD partner (the wave that is behind runs first); 0: A/B only
#endif
#define GR_RES_GROUPS 1024         // 4-atom groups per workgroup at most (two per lane)
#ifndef GR_RES_MAX_FIN
#define GR_RES_MAX_FIN 16          // finalizer workgroups of a launch with more than 8 frame streams, 8 otherwise (fewer when the streaming workgroups leave fewer CUs)
#endif
#ifndef GR_RES_MAX_STREAMS
#define GR_RES_MAX_STREAMS 32      // frame streams side by side in one launch (frames that fill a fraction of the chip)
#endif
// Every wait is bounded in TIME (the device's constant-rate clock, wall_clock64: read every 256 polls) -- ctl.patience_ticks, ~3 s,
// for a record, ctl.start_ticks, ~0.2 s, for the start handshake (other kernels may hold CUs when the launch begins) -- and, as a
// second line, in polls: a poll costs ~1 us on an idle device but many times that under a profiler's counter collection, where a
// bound in polls alone let a stuck launch sit for minutes.
#define GR_RES_PATIENCE 30000000u        // polls before a wait gives up whatever the clock says
#define GR_RES_START_PATIENCE 2000000u   // ... of the start handshake
#define GR_RES_IDLE_WAVE 0xFFFFFFFFu /* ctl.progress word of a streaming wave without atoms (the ragged last workgroup of a frame) */
#define GR_ST_ABORTED 102          /* internal: the frame's finalizer gave up (abort): the frame is untouched and is redone on the two-pass path */

#define GR_RES_REC_WORDS 32         // tagged words per workgroup record: 0..18 sums, 19..30 extents (as maxima), 31 unused
#define GR_RES_REC_PAD 32           // workgroup records per frame are padded to a multiple of this
struct GrResCtl {
    unsigned long long *wgrec;     // [frames][n_stream padded][32] value | epoch << 32
    unsigned long long *rec;       // [frames][16] value | epoch << 32: 0 status, 1..3 shift, 4..12 R (column-major), 13..15 t0 = -R (COM - first atom)
    uint32_t *abort;               // [12]: 0 abort (0 = fine), 1 workgroups that have started, 2 start verdict (0 open, 1 go, 2 never started);
                                   // 4-5 the metronome's origin t0 (device clock, written by the workgroup that opens the launch), 6-7 the clock
                                   // when the last streaming wave left, 8 slots that waves reached late, 9 slots they waited for;
                                   // 10, 11 shader-clock and device-clock ticks (/ 256) of workgroup 0's walk;
                                   // words 1 .. 9 are zeroed by the host before every launch
    uint32_t *progress;            // [n_stream][8]: turns (frames of its stream) each streaming wave had been through when it left
    uint32_t epoch, n_stream, n_fin;   // n_stream = streams x wgs_frame streaming workgroups, then n_fin finalizers
    uint32_t wgs_frame, streams;   // workgroups one frame needs; frame streams the launch runs side by side (stream s: frames s, s + streams, ...)
    uint32_t groups_wg;            // 4-atom groups per streaming workgroup: a multiple of 64, 64 .. GR_RES_GROUPS
    uint32_t team_waves;           // waves of a finalizer workgroup that close one frame together: 1, 2, 4 or 8 with 32 x that >= wgs_frame
    unsigned long long patience_ticks, start_ticks;   // bounds of the waits in ticks of wall_clock64() (the host knows the rate)
    uint32_t test_abort_frame;     // tests: the finalizer of this frame raises `abort` instead of closing it (0xFFFFFFFF: never)
    uint32_t metro_t16;            // METRONOME (see the walk): the turn period in ticks of wall_clock64() x 16; 0 = the waves run free
    uint32_t metro_lead;           // ticks between the opening of the launch and turn 0's first slot
    uint32_t cen_weighted, cen_dim_mask;   // MODE 1 (atoms_center): the reference group's centre is mass-weighted; the Dimension's axes (bit 0 x, 1 y, 2 z)
#ifdef GR_EXP_TIMELINE
    unsigned long long *tl;        // [frames][8] device-clock stamps of a frame's way through the launch (tools/timeline_bench.sh)
#endif
#ifdef GR_EXP_STEPTIME
    unsigned long long *dbg;       // [streaming waves][4]: shader-clock ticks spent waiting for records, fits that polled, XCC | SIMD << 8, total ticks
#endif
};

// before a launch: the start handshake's two words (count, verdict) and every streaming wave's progress word are zeroed
__global__ void k_res_prepare(uint32_t *handshake, uint32_t *progress, uint32_t n_progress) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 11u) handshake[i] = 0u;                                  // (ctl.abort + 1 .. + 11)
    if (i < n_progress) progress[i] = 0u;
}

template <typename T> __device__ __forceinline__ T gr_ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float gr_first_f(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
__device__ __forceinline__ float gr_lane_f(unsigned long long v, int l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l)); }
// release / acquire of LDS data between the waves of a workgroup: fences restricted to the LDS address space -- the unrestricted
// workgroup fence also waits for the wave's outstanding global loads (vmcnt(0)), i.e. for the rows it prefetched a frame ahead
__device__ __forceinline__ void gr_lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void gr_lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }

// (buffer resources and loads: gr_layout.h)
// the fitted rows leave as agent-scope non-temporal stores (sc1 nt): measured on a persistent copy of this launch shape
// (tools/store_variants.hip, three runs): 4.31-4.35 us per 1e6-atom frame against 4.45-4.49 with plain non-temporal stores
#ifndef GR_RES_STORE_AUX
#define GR_RES_STORE_AUX 18        /* 1 sc0, 2 nt, 16 sc1 */
#endif
__device__ __forceinline__ void gr_buf_store_stream(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, const float4 &v) {
    gr_i4 t; t.x = __float_as_int(v.x); t.y = __float_as_int(v.y); t.z = __float_as_int(v.z); t.w = __float_as_int(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)byte_off, 0, GR_RES_STORE_AUX);
}

// one 4-atom group of a lane: which of its atoms belong to the selection, its reference rows, masses and weights (registers)
// (the per-lane facts are bits of ONE register: seven `bool`s would be seven 64-bit lane masks -- 14 SGPRs per group held
// across the whole loop, and the kernel was spilling SGPRs into vector lanes)
enum { GR_RG_IN0 = 1u, GR_RG_IN1 = 2u, GR_RG_IN2 = 4u, GR_RG_IN3 = 8u, GR_RG_ANY = 16u /* some atom in the selection */, GR_RG_FULL = 32u /* all four */,
       GR_RG_EX0 = 64u, GR_RG_EX1 = 128u, GR_RG_EX2 = 256u, GR_RG_EX3 = 512u /* the atom exists (index < n_atoms): MODE 1 moves every atom of the system, not only the group's */, GR_RG_EXALL = 1024u };
struct GrResGroup {
    bool valid;        // wave-uniform: the group lies inside the slot
    uint32_t flags;
    uint32_t b;        // float4 index of the group's first row inside a slot
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
__device__ __forceinline__ void gr_res_fit_group(const GrResGroup &Gr, const float4 &r0, const float4 &r1, const float4 &r2, const GrResRot &T, const GrBoxU &B,
                                                 const GrBox *__restrict__ boxp, float cx, float cy, float cz, float4 &o0, float4 &o1, float4 &o2, float &rs) {
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
    if (Gr.flags & GR_RG_ANY) {   // sum w |R q - p|^2 (rmsd.rs:592-599); the weights of atoms outside the selection are zero
        const gr_v2f w01 = WMASS ? gr_v2p(Gr.mm.x, Gr.mm.y) : gr_v2p(Gr.ww.x, Gr.ww.y), w23 = WMASS ? gr_v2p(Gr.mm.z, Gr.mm.w) : gr_v2p(Gr.ww.z, Gr.ww.w);
        gr_v2f dx = n.x01 - Gr.P.x01, dy = n.y01 - Gr.P.y01, dz = n.z01 - Gr.P.z01;
        gr_v2f s01 = gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz));
        dx = n.x23 - Gr.P.x23; dy = n.y23 - Gr.P.y23; dz = n.z23 - Gr.P.z23;
        gr_v2f s23 = gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz));
        // atoms of the group outside the selection (its ragged ends, clear bits of a masked one) contribute an exact zero -- the TERM,
        // not just the weight: such an atom may have no position, and 0 * NaN would take the frame's rmsd with it
        if (__builtin_amdgcn_ballot_w64((Gr.flags & GR_RG_FULL) == 0u) != 0ull) {
            s01.x = (Gr.flags & GR_RG_IN0) ? s01.x : 0.f; s01.y = (Gr.flags & GR_RG_IN1) ? s01.y : 0.f;
            s23.x = (Gr.flags & GR_RG_IN2) ? s23.x : 0.f; s23.y = (Gr.flags & GR_RG_IN3) ? s23.y : 0.f;
        }
        const gr_v2f part = gr_v2_fma(w23, s23, w01 * s01);
        rs += part.x + part.y;
    }
    n.x01 += gr_v2(cx); n.y01 += gr_v2(cy); n.z01 += gr_v2(cz); n.x23 += gr_v2(cx); n.y23 += gr_v2(cy); n.z23 += gr_v2(cz);
    gr_rows_pairs(n, o0, o1, o2);                     // (stored by the caller: unconditionally, see `fit`)
}

// UBOX: every frame of the launch has the same box (the host compared them): its constants are loaded once, not per frame.
// V: the selection is the whole system -> image vectors are parked (see the header of this file).
// FL: a step runs the sums of frame i before the fit of frame i - K (see `step`)
// MODE 1: ATOMS_CENTER (utility.rs:109-185) instead of the RMSD fit -- the same walk with other arithmetic: the sums stage forms the Bai-Breen
//         sums of the reference group (`sel`; gr_center_atom<1>'s terms, 4-atom f32 partials widened to fp64 as k_center_sums does), a finalizer
//         turns them into the group's centre (gr_center_close) and publishes shift = filter(box centre - centre, Dimension), the fit stage is
//         x <- wrap(x + shift) for EVERY atom of the system (k_translate_wrap's arithmetic).  24 B per atom and frame instead of 36.  V is false
//         (rows are parked), WMASS unused (ctl.cen_weighted), `plan` unused.  A frame with an atom without position, or whose sums are not
//         finite, is published as GR_ST_FALLBACK: nobody touches it and the host runs the two passes on it, which name the atom.
template <bool WMASS, bool UBOX, bool V, bool FL, int MODE = 0>
__global__ __launch_bounds__(GrResShape::LANES) void k_fit_resident(
    float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, uint32_t nframes, uint32_t n_atoms,
    const float *__restrict__ masses, GrSel sel, const GrBox *__restrict__ boxes, GrPlanDev plan,
    GrFrameState *state, double *__restrict__ fit_partials, GrResCtl ctl) {
    typedef GrResShape S;
    constexpr uint32_t LANES = S::LANES, WAVES = S::WAVES, K = S::K, R = S::R;
    extern __shared__ float4 smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_pad = (ctl.wgs_frame + GR_RES_REC_PAD - 1u) & ~(uint32_t)(GR_RES_REC_PAD - 1u);

    // ------------------------------------------------------------------------------------------ start handshake
    // Every workgroup waits for data other workgroups produce, so nothing may begin before ALL of them are on the chip: each one
    // checks in, the one that completes the count opens the launch (verdict 1).  A workgroup that waits too long -- the device is
    // shared with another process's kernels that will not leave, or two of these launches hold half the chip each -- closes it
    // (verdict 2) and everybody leaves WITHOUT having touched a frame: the host then runs the segment on the two-pass path.
    __shared__ unsigned long long metro_t0;
    {
        __shared__ uint32_t verdict;
        if (tid == 0) {
            const uint32_t n = __hip_atomic_fetch_add(ctl.abort + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            uint32_t zero = 0u;
            if (n == gridDim.x) {
                // the workgroup that opens the launch also sets the metronome's origin: visible (release) before the verdict is
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(ctl.abort + 4), wall_clock64() + ctl.metro_lead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                (void)__hip_atomic_compare_exchange_strong(ctl.abort + 2, &zero, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            uint32_t v, polls = 0;
            const unsigned long long t0 = wall_clock64();
            while ((v = gr_ld_agent(ctl.abort + 2)) == 0u) {
                if (++polls > GR_RES_START_PATIENCE || ((polls & 255u) == 0 && wall_clock64() - t0 > ctl.start_ticks)) { zero = 0u; (void)__hip_atomic_compare_exchange_strong(ctl.abort + 2, &zero, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                __builtin_amdgcn_s_sleep(16);
            }
            verdict = v;
            if (v == 1u) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); metro_t0 = gr_ld_agent(reinterpret_cast<const unsigned long long *>(ctl.abort + 4)); }
        }
        __syncthreads();
        if (verdict != 1u) return;
    }

    // ------------------------------------------------------------------------------------------ finalizers
    if (blockIdx.x >= ctl.n_stream) {
        constexpr uint32_t RPW = S::REC_PER_WAVE;
        // A frame of up to 32 / 64 / 128 / 256 workgroups is closed by a TEAM of 1 / 2 / 4 / 8 waves (a wave collects 32 records),
        // and a finalizer workgroup closes 8 / 4 / 2 / 1 consecutive frames at a time, one per team: the closing arithmetic is one
        // lane's chain of fp64 operations (~10 us), so with several frame streams the frames per second the finalizers can close
        // is what bounds the launch -- teams in different waves run their chains side by side.
        const uint32_t TW = ctl.team_waves, NT = WAVES / TW, team = wave / TW, wt = wave % TW;
        double *wtot = reinterpret_cast<double *>(smem);              // [WAVES][32] per wave + [teams <= WAVES][32] totals
        uint32_t *gave_up = reinterpret_cast<uint32_t *>(wtot + 2u * WAVES * 32u);   // some wave of this workgroup ran out of patience
        if (tid == 0) *gave_up = 0u;
        __syncthreads();
        const unsigned long long tagv = (unsigned long long)ctl.epoch << 32;
        for (uint32_t f0 = (blockIdx.x - ctl.n_stream) * NT; f0 < nframes; f0 += ctl.n_fin * NT) {
            const uint32_t f = f0 + team < nframes ? f0 + team : nframes - 1u;
            const bool live = f0 + team < nframes;                   // (a team without a frame in the last round only keeps the barriers)
            // what the closing step needs besides the sums: requested now, in flight while the records are awaited
            const GrBox *bp = boxes + first_slot + f;
            GrBox lb;
            lb.ax = bp->ax; lb.by = bp->by; lb.cz = bp->cz; lb.bx = bp->bx; lb.cx = bp->cx; lb.cy = bp->cy;
            lb.bcx = bp->bcx; lb.bcy = bp->bcy; lb.bcz = bp->bcz; lb.r_ws = bp->r_ws; lb.ortho = bp->ortho;
            float g0x, g0y, g0z;
            gr_pos_load(frames + (size_t)(first_slot + f) * frame_stride, sel.start, g0x, g0y, g0z);
            const int pre_status = state[f].status;
            // The wave's 32 records are 8 KiB in a row, and the wave reads them AS 8 KiB: eight 16-byte loads per lane, each instruction one
            // contiguous KiB (round 5; until then a lane read its own record's words 8 bytes at a time -- 1024 cache lines touched per wave
            // and look, 16 instructions of 64 scattered 8-byte accesses: a look took microseconds and the looks of all finalizers were a
            // sixth of the launch's memory requests).  Load k hands lane l the words 2 (l % 16) and 2 (l % 16) + 1 of record 4 k + l / 16.
            // (a team may have more waves than the frame has blocks of 32 records: such a wave reads a resource of zero bytes)
            const uint32_t rec0 = wt * RPW < n_pad ? wt * RPW : 0u, nrec = wt * RPW < n_pad ? min(RPW, n_pad - wt * RPW) : 0u;
            const unsigned long long *src = ctl.wgrec + ((size_t)f * n_pad + rec0) * GR_RES_REC_WORDS;
            const __amdgpu_buffer_rsrc_t rr = gr_buf_rsrc(src, nrec * GR_RES_REC_WORDS * 8u);
            const uint32_t w0 = (lane & 15u) * 2u, rbase = wt * RPW + (lane >> 4);
            gr_i4 q[8];
            uint32_t polls = 0;
            const unsigned long long t0 = wall_clock64();
            if (live && f == ctl.test_abort_frame) { if (lane == 0) { gr_st_agent(ctl.abort, 1u); __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } polls = 0xFFFFFFFFu; }
            while (polls != 0xFFFFFFFFu) {
                bool ok = true;
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k) q[k] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)((k * 64u + lane) * 16u), 0, 16 /* sc1: agent scope */);
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k) {
                    const bool mine = live && rbase + 4u * k < ctl.wgs_frame;
                    ok = ok && (!mine || ((uint32_t)q[k].y == ctl.epoch && ((uint32_t)q[k].w == ctl.epoch || w0 == 30u)));   // (word 31 is never written)
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (++polls > GR_RES_PATIENCE || ((polls & 255u) == 0 && (gr_ld_agent(ctl.abort) != 0u || wall_clock64() - t0 > ctl.patience_ticks))) {
                    if (lane == 0) { gr_st_agent(ctl.abort, 1u); __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    polls = 0xFFFFFFFFu;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#ifdef GR_EXP_TIMELINE
            if (ctl.tl && wt == 0 && lane == 0 && live) { ctl.tl[(size_t)f * 8 + 1] = wall_clock64(); ctl.tl[(size_t)f * 8 + 7] = polls; }
#endif
            // (a wave that gave up still meets the others at the barriers; the frame is then published as ABORTED, never as closed)
            // sums in fp64, extents as maxima (words 0..18 sums, 19..30 maxima): the lane's eight records in order, then the four lanes
            // that hold the same words (l, l ^ 16, l ^ 32, l ^ 48) -- a fixed order: results do not depend on the order of arrival
            const bool mx0 = w0 >= 19u, mx1 = w0 + 1u >= 19u;
            double a0 = mx0 ? (double)-3.0e38f : 0.0, a1 = mx1 ? (double)-3.0e38f : 0.0;
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) {
                const bool have = live && rbase + 4u * k < ctl.wgs_frame && polls != 0xFFFFFFFFu;
                const double x0 = have ? (double)__int_as_float(q[k].x) : (mx0 ? (double)-3.0e38f : 0.0);
                const double x1 = (have && w0 != 30u) ? (double)__int_as_float(q[k].z) : (mx1 ? (double)-3.0e38f : 0.0);
                a0 = mx0 ? fmax(a0, x0) : a0 + x0;
                a1 = mx1 ? fmax(a1, x1) : a1 + x1;
            }
#pragma unroll
            for (uint32_t off = 16u; off < 64u; off <<= 1) {
                const double o0 = __shfl_xor(a0, (int)off, 64), o1 = __shfl_xor(a1, (int)off, 64);
                a0 = mx0 ? fmax(a0, o0) : a0 + o0;
                a1 = mx1 ? fmax(a1, o1) : a1 + o1;
            }
            if (lane < 16u) { wtot[wave * 32u + w0] = a0; wtot[wave * 32u + w0 + 1u] = a1; }
            __syncthreads();                                          // (also publishes gave_up)
            if (wt == 0 && lane < 31u) {     // totals over the team's waves, in wave order: lane k owns word k
                double a = wtot[wave * 32u + lane];
                const bool mx = lane >= 19u;
                for (uint32_t wv = 1; wv < TW; ++wv) { const double o = wtot[(wave + wv) * 32u + lane]; a = mx ? fmax(a, o) : a + o; }
                wtot[(WAVES + team) * 32u + lane] = a;
            }
            gr_wave_sync();
            if (wt == 0 && lane == 0 && live) {
#ifdef GR_EXP_TIMELINE
                if (ctl.tl) ctl.tl[(size_t)f * 8 + 2] = wall_clock64();
#endif
                GrFrameState st = state[f];                            // (closed in registers, stored once: results are read back below)
#ifdef GR_EXP_TIMELINE
                if (ctl.tl) { if (st.status > 1000000) ctl.tl[0] = 0; ctl.tl[(size_t)f * 8 + 3] = wall_clock64(); }
#endif
                const bool lost = *gave_up != 0u;                      // ANY wave of the workgroup gave up on a record of this round or an earlier one
                const double *t = wtot + (WAVES + team) * 32u;
                if (lost) {
                    if (pre_status == 0) st.status = GR_ST_ABORTED;
                } else if (MODE == 1) {
                    if (pre_status == 0) {
                        // the group's Bai-Breen sums (hi + lo words), closed as k_center_finalize closes them; a frame with an atom without
                        // position / mass, or with sums that are not finite, goes back to the host untouched (the two passes name the atom)
                        double acc[GR_CEN_K];
                        bool fine = !(t[7] + t[15] > 0.0);
#pragma unroll
                        for (int k = 0; k < GR_CEN_K; ++k) acc[k] = k < 6 ? t[k] + t[8 + k] : 0.0;
#pragma unroll
                        for (int k = 0; k < 6; ++k) fine = fine && (fabs(acc[k]) <= 1.0e300);
                        if (!fine) st.status = GR_ST_FALLBACK;
                        else gr_center_close(acc, GR_NOIDX, GR_NOIDX, lb, 1, (int)ctl.cen_weighted, 1, 0, sel.n, st, 0);
                        // shift = filter(box centre - centre, Dimension) (utility.rs:116-119; k_translate_wrap's own subtraction)
                        st.shift[0] = (ctl.cen_dim_mask & 1u) ? lb.bcx - st.center[0] : 0.0f;
                        st.shift[1] = (ctl.cen_dim_mask & 2u) ? lb.bcy - st.center[1] : 0.0f;
                        st.shift[2] = (ctl.cen_dim_mask & 4u) ? lb.bcz - st.center[2] : 0.0f;
#pragma unroll
                        for (int k = 0; k < 9; ++k) st.R[k] = 0.0f;
                    }
                } else if (pre_status == 0) {
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
#ifdef GR_EXP_NOFINMATH
                    st.R[0] = st.R[4] = st.R[8] = 1.0f; st.R[1] = st.R[2] = st.R[3] = st.R[5] = st.R[6] = st.R[7] = 0.0f; st.shift[0] = st.shift[1] = st.shift[2] = (float)(acc[1] * 1e-9); st.status = 0;
#else
                    gr_finalize_math<0, true, false>(acc, mn, mx3, fmn, fmx, GR_NOIDX, GR_NOIDX, lb, plan, g, sel.n, st);
#endif
                }
#ifdef GR_EXP_TIMELINE
                if (ctl.tl) { if (st.R[0] > 1.0e30f) ctl.tl[0] = 0; ctl.tl[(size_t)f * 8 + 4] = wall_clock64(); }
#endif
                unsigned long long *o = ctl.rec + (size_t)f * 16;
                // (a frame that was not closed is published as failed: the streaming waves leave it unmodified and move on)
                gr_st_agent(o + 0, tagv | (uint32_t)st.status);
#pragma unroll
                for (int k = 0; k < 3; ++k) gr_st_agent(o + 1 + k, tagv | __float_as_uint(st.shift[k]));
#pragma unroll
                for (int k = 0; k < 9; ++k) gr_st_agent(o + 4 + k, tagv | __float_as_uint(st.R[k]));
                // t0 = -R cv, cv = COM - first atom (the sums are relative to the first atom): what the V kernel adds to R v; formed from
                // the f32 R that every wave applies, so that R v + t0 is R (v - cv) to rounding
                const double cvx = t[1] / t[0], cvy = t[2] / t[0], cvz = t[3] / t[0];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float t0 = (float)(-((double)st.R[a] * cvx + (double)st.R[3 + a] * cvy + (double)st.R[6 + a] * cvz));
                    gr_st_agent(o + 13 + a, tagv | __float_as_uint(t0));
                }
                state[f] = st;
#ifdef GR_EXP_TIMELINE
                if (ctl.tl) ctl.tl[(size_t)f * 8 + 5] = wall_clock64();
#endif
            }
            __syncthreads();                                          // wtot is free again
        }
        return;
    }

    // ------------------------------------------------------------------------------------------ streaming workgroups
    const uint32_t ngroups = ((n_atoms + 255u) >> 8) << 6;            // the slot is padded to whole tiles (a multiple of 64 groups)
    // Frames smaller than half the chip run as several STREAMS side by side: workgroup b belongs to stream b / wgs_frame and owns
    // the atoms of workgroup b % wgs_frame of every frame of its stream (frames s, s + streams, ...): each stream is the pipeline
    // described above on its own share of the CUs, the finalizers serve them all.  `kf(k)` = the frame of the stream's k-th turn.
    // A workgroup owns ctl.groups_wg consecutive groups of the frame -- 1024 (two per lane) when whole frames fill the chip, fewer
    // (a multiple of 64: whole waves) when a frame cut into 1024s would leave CUs idle: the first 512 are the lanes' groups A, the
    // rest groups B of the first waves -- one wave per SIMD first, so 768 groups load every SIMD with 3 group-units instead of 4.
    const uint32_t wg_all = blockIdx.x, stream = wg_all / ctl.wgs_frame, wg = wg_all % ctl.wgs_frame, base = wg * ctl.groups_wg;
    const uint32_t glimit = min(base + ctl.groups_wg, ngroups);   // one past the workgroup's last group
    const uint32_t n_turns = nframes > stream ? (nframes - stream + ctl.streams - 1u) / ctl.streams : 0u;
    auto kf = [&](uint32_t k) { return stream + k * ctl.streams; };
    if (base + wave * 64u >= glimit) {                                // every chunk of this wave lies behind the workgroup's last group: nothing to fit
        if (lane == 0) ctl.progress[wg_all * WAVES + wave] = GR_RES_IDLE_WAVE;   // (not "all turns done": the host skips this word)
        if (MODE == 0 && fit_partials) for (uint32_t k = lane; k < n_turns; k += 64u) fit_partials[((size_t)kf(k) * ctl.wgs_frame + wg) * WAVES + wave] = 0.0;   // (its words of the frames' fit sums)
        return;
    }
    const uint32_t n_waves = min(WAVES, (glimit - base) >> 6);
    float4 *park = smem;
    float *wsum = reinterpret_cast<float *>(smem + S::PARK_F4);
    double *fsum = reinterpret_cast<double *>(wsum + S::WSUM_F);
    uint32_t *cnt_s = reinterpret_cast<uint32_t *>(fsum + R * WAVES);   // [R][2]: arrivals at the sums hand-over of a slot (cumulative), combines done for it
    uint32_t *prog = cnt_s + 3 * R, *simd_of = prog + WAVES;             // iterations each wave has begun; the SIMD each wave runs on
#ifdef GR_EXP_TIMELINE
    unsigned long long *tls = reinterpret_cast<unsigned long long *>(simd_of + WAVES);   // [256][4]
    const bool tl_wg = ctl.tl != nullptr && wg_all == 0u;
#endif
    if (tid < 3 * R) cnt_s[tid] = 0u;
    if (lane == 0) { prog[wave] = 0u; simd_of[wave] = (uint32_t)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) /* HW_ID.SIMD_ID */; }
    __syncthreads();                                                  // the only barrier: before the first frame
    // the other wave of this workgroup on the same SIMD: the two share the SIMD's issue slots, and whichever of them is BEHIND gets
    // them first (see `balance` below)
    uint32_t partner = WAVES;
    if (GR_RES_BAL) {
        const uint32_t mine = simd_of[wave];
        uint32_t found = 0;
        for (uint32_t w = 0; w < n_waves; ++w) if (w != wave && simd_of[w] == mine) { partner = w; ++found; }
        if (found != 1u) partner = WAVES;
        partner = (uint32_t)__builtin_amdgcn_readfirstlane((int)partner);
    }

    // (a masked selection -- GrSel::masked, gr_hot.h -- is its span with a bit per atom: the lane's membership flags take the bits in,
    //  once, and nothing else in the launch knows the difference; plan.p is then the plan's by-atom copy over the span)
    const uint32_t first = sel.start, last = sel.start + (sel.masked ? sel.span : sel.n), g0 = sel.g0 << 6;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // the lane's groups: A = chunk `wave` of the workgroup's first LANES groups, B = the same chunk of its second LANES
    auto setup = [&](uint32_t g, GrResGroup &Gr) {
        const uint32_t i0 = g << 2;
        Gr.valid = g < glimit;                                        // wave-uniform
        bool in0 = Gr.valid && (i0 >= first) && (i0 < last), in1 = Gr.valid && (i0 + 1u >= first) && (i0 + 1u < last);
        bool in2 = Gr.valid && (i0 + 2u >= first) && (i0 + 2u < last), in3 = Gr.valid && (i0 + 3u >= first) && (i0 + 3u < last);
        if (!V && sel.masked && (in0 || in1 || in2 || in3)) {
            const uint32_t nib = (sel.mask[g >> 3] >> ((g & 7u) * 4u)) & 15u;
            in0 = in0 && (nib & 1u); in1 = in1 && (nib & 2u); in2 = in2 && (nib & 4u); in3 = in3 && (nib & 8u);
        }
        const bool in_sel = in0 || in1 || in2 || in3, full = in0 && in1 && in2 && in3;
        Gr.flags = (in0 ? GR_RG_IN0 : 0u) | (in1 ? GR_RG_IN1 : 0u) | (in2 ? GR_RG_IN2 : 0u) | (in3 ? GR_RG_IN3 : 0u) | (in_sel ? GR_RG_ANY : 0u) | (full ? GR_RG_FULL : 0u);
        if (MODE == 1) {
            const bool e0 = Gr.valid && i0 < n_atoms, e1 = Gr.valid && i0 + 1u < n_atoms, e2 = Gr.valid && i0 + 2u < n_atoms, e3 = Gr.valid && i0 + 3u < n_atoms;
            Gr.flags |= (e0 ? GR_RG_EX0 : 0u) | (e1 ? GR_RG_EX1 : 0u) | (e2 ? GR_RG_EX2 : 0u) | (e3 ? GR_RG_EX3 : 0u) | ((e0 && e1 && e2 && e3) ? GR_RG_EXALL : 0u);
        }
        Gr.b = (uint32_t)gr_row_index(Gr.valid ? g : 0u, 0);
        float4 pa = zero4, pb = zero4, pc = zero4;
        Gr.mm = zero4; Gr.ww = zero4;
        if (MODE == 1) {   // the group's weights: its masses, or one per atom (gr_center_atom: `weighted`)
            if (in_sel) {
                Gr.mm = ctl.cen_weighted ? reinterpret_cast<const float4 *>(masses)[g] : make_float4(1.f, 1.f, 1.f, 1.f);
                if (!in0) Gr.mm.x = 0.f;
                if (!in1) Gr.mm.y = 0.f;
                if (!in2) Gr.mm.z = 0.f;
                if (!in3) Gr.mm.w = 0.f;
            }
        } else if (in_sel) {
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
    setup(base + LANES + tid, GB);
    const float cx = plan.ref_com[0], cy = plan.ref_com[1], cz = plan.ref_com[2];
    // sum of the masses of the wave's atoms inside the selection: the same for every frame of the launch (an SGPR)
    const float m_wave = gr_first_f(gr_wave_allsum_f32(((GA.mm.x + GA.mm.y) + (GA.mm.z + GA.mm.w)) + ((GB.mm.x + GB.mm.y) + (GB.mm.z + GB.mm.w))));
    // MODE 1: an atom of the group without mass (NaN; only a weighted centre looks at masses): every frame of the launch goes back to the host
    const bool cen_mass_bad = MODE == 1 && ((GA.mm.x != GA.mm.x) || (GA.mm.y != GA.mm.y) || (GA.mm.z != GA.mm.z) || (GA.mm.w != GA.mm.w) ||
                                            (GB.mm.x != GB.mm.x) || (GB.mm.y != GB.mm.y) || (GB.mm.z != GB.mm.z) || (GB.mm.w != GB.mm.w));
    // does the wave hold any atom of the selection?  (V: always; otherwise most waves of a small selection do not, and skip the sums arithmetic)
    const bool wave_sel = V || __builtin_amdgcn_ballot_w64(((GA.flags | GB.flags) & GR_RG_ANY) != 0u) != 0ull;

    struct Rows { float4 r0, r1, r2; };
    struct Landing { Rows a, b; float gx, gy, gz; };
    // (unconditional: a turn behind the stream's last one reads a resource of zero records, a missing group B an offset nowhere)
    const uint32_t slot_bytes = (uint32_t)(frame_stride * sizeof(float));
    const uint32_t offA = GA.b * 16u, offB = GB.valid ? GB.b * 16u : GR_BUF_NOWHERE;
    const uint32_t offgx = (uint32_t)gr_tile_index(first, 0) * 4u, offgy = (uint32_t)gr_tile_index(first, 1) * 4u, offgz = (uint32_t)gr_tile_index(first, 2) * 4u;
    auto request = [&](uint32_t k, Landing &L) {
        const bool live = k < n_turns;
        const __amdgpu_buffer_rsrc_t rs = gr_buf_rsrc(frames + (size_t)(first_slot + (live ? kf(k) : 0u)) * frame_stride, live ? slot_bytes : 0u);
        L.a.r0 = gr_buf_load_stream(rs, offA); L.a.r1 = gr_buf_load_stream(rs, offA + 1024u); L.a.r2 = gr_buf_load_stream(rs, offA + 2048u);
        L.b.r0 = gr_buf_load_stream(rs, offB); L.b.r1 = gr_buf_load_stream(rs, offB + 1024u); L.b.r2 = gr_buf_load_stream(rs, offB + 2048u);
        L.gx = gr_buf_load_f32(rs, offgx); L.gy = gr_buf_load_f32(rs, offgy); L.gz = gr_buf_load_f32(rs, offgz);   // provisional centre: the first atom of the selection
    };
    auto request_rec = [&](uint32_t k) -> unsigned long long { return lane < 16u ? gr_ld_agent(ctl.rec + (size_t)kf(k) * 16 + lane) : 0ull; };
    bool bail = false;
    uint32_t n_fitted = 0;                                             // frames whose fit stage this wave has been through (-> ctl.progress)
    // Two waves share a SIMD, and the frame rate of the whole launch is the rate of its SLOWEST wave (every frame's record needs
    // every wave's sums).  Left to the hardware's tie-break -- the older wave first -- the older wave of each pair runs ahead until
    // it has to wait for a record, and the younger one gets the issue slots that are left over.  So each wave publishes how many
    // iterations it has begun, looks at its partner's count once per iteration, and takes priority 3 when it is behind, 2 level, 1 ahead
    // (0 while it polls for a record): the pair advances together and neither waits for the other.
    // (Measured: 6.26 -> 5.44 us per frame.  The counts are hints: relaxed LDS accesses, nothing depends on their order.)
    uint32_t my_prio = 2u;
    auto set_prio = [&]() {
        if (my_prio == 3u) __builtin_amdgcn_s_setprio(3); else if (my_prio == 2u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
    };
    auto balance = [&](uint32_t i) {
        if (!GR_RES_BAL || partner >= WAVES) return;
        if (lane == 0) __hip_atomic_store(prog + wave, i + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t pp = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(prog + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        my_prio = pp > i + 1u ? 3u : (pp == i + 1u ? 2u : 1u);
        set_prio();
    };

    // THE METRONOME.  Left alone, the waves issue their row requests whenever their own turn comes round: at any moment the chip asks for
    // 1 KiB pieces from all over the frame in no order, and HBM delivers a copy at 5.2-5.5 TB/s.  A copy whose requests sweep memory
    // in ADDRESS ORDER -- fresh single-load waves handed out by the dispatcher -- runs at 6.5 (tools/copy_matrix*.hip, profiles/
    // r05_copy_matrix.md).  So the walk is given a clock: wave w of streaming workgroup b may request the rows of turn k no earlier than
    //      t0 + (k + (8 b + w) / (8 n_stream)) T        (device clock; t0 set by the workgroup that opens the launch, T = ctl.metro_t16 / 16)
    // -- the chip's requests, and six turns later its stores, then sweep every frame from its first byte to its last once per period.
    // The same walk without arithmetic: 4.61 us per 1e6-atom frame free-running, 3.70 with the clock at T = 3.7 us (6.4 TB/s), 4.04 with
    // a T it cannot keep (3.6).  A wave that reaches its slot late does not wait (it is counted); T is the host's to choose (gr_api.hip).
    const unsigned long long metro_base = metro_t0;
#ifndef GR_RES_METRO_ANTI
#define GR_RES_METRO_ANTI 0        /* experiment: waves 4-7 (the second wave of every SIMD) half a period behind waves 0-3 */
#endif
    const unsigned long long metro_phase16 = ((unsigned long long)(wg_all * WAVES + wave) * ctl.metro_t16) / ((unsigned long long)ctl.n_stream * WAVES)
                                           + (GR_RES_METRO_ANTI ? (unsigned long long)(wave >> 2) * (ctl.metro_t16 >> 1) : 0ull);
    uint32_t n_late = 0, n_waited = 0;
    // (workgroup 0's first wave also reads the SHADER clock at both ends of its walk: with the device clock beside it the host knows the
    //  frequency the launch ran at -- under the power cap the pass, streaming and computing at once, does not run at the nominal 2.4 GHz)
#ifdef GR_EXP_STEPTIME
    // experiment: where a wave's turn goes (shader-clock ticks summed over the walk, per segment): 0 the metronome's wait, 1 request + balance,
    // 2 the wait for the rows + images + sums arithmetic, 3 reductions + hand-over (+ the combine, when this wave claims one), 4 the wait
    // for the record (first look / polls), 5 fit arithmetic + stores + hand-over of the fit sum, 6 parking; 7 = fit stages that had to poll
    unsigned long long st_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, st_last = (unsigned long long)clock64(), st_combines = 0ull;
#define GR_STEP_STAMP(k) do { const unsigned long long now_ = (unsigned long long)clock64(); st_acc[k] += now_ - st_last; st_last = now_; } while (0)
#else
#define GR_STEP_STAMP(k) do { } while (0)
#endif
    const bool clk_wave = wg_all == 0u && wave == 0u;
    unsigned long long clk_c0 = 0ull, clk_w0 = 0ull;
    if (clk_wave) { clk_c0 = (unsigned long long)clock64(); clk_w0 = (unsigned long long)wall_clock64(); }
    auto gate = [&](uint32_t turn) {
        if (ctl.metro_t16 == 0u || turn >= n_turns) return;
        const unsigned long long target = metro_base + (((unsigned long long)turn * ctl.metro_t16 + metro_phase16) >> 4);
        long long early = (long long)(target - wall_clock64());
        if (early <= 0) { n_late += early < -(long long)(ctl.metro_t16 >> 6) ? 1u : 0u; return; }   // (late by more than a quarter of a period)
        if (early > (long long)ctl.metro_t16 * 4) return;             // (farther ahead than any wave can be, 64 periods: the origin is not this launch's; run free)
        n_waited++;
        __builtin_amdgcn_s_setprio(0);
        uint32_t polls = 0;
        do { __builtin_amdgcn_s_sleep(1); early = (long long)(target - wall_clock64()); } while (early > 0 && ++polls < 100000u);
        set_prio();
    };

    // the image vectors of one group: v = image of (x - first atom) nearest to the first atom
    auto images = [&](const GrResGroup &Gr, const Rows &rw, const GrBoxU &B, const GrBox *boxp, float gx, float gy, float gz) -> GrP4 {
        GrP4 q = gr_pairs_rows(rw.r0, rw.r1, rw.r2);
        // atoms outside the selection (V: the pad atoms behind the last one) become copies of the first atom: v = 0 adds nothing and
        // lies inside every extent (a whole wave of complete groups skips this on one scalar branch)
        if (__builtin_amdgcn_ballot_w64((Gr.flags & GR_RG_FULL) == 0u) != 0ull) {
            if (!(Gr.flags & GR_RG_IN0)) { q.x01.x = gx; q.y01.x = gy; q.z01.x = gz; }
            if (!(Gr.flags & GR_RG_IN1)) { q.x01.y = gx; q.y01.y = gy; q.z01.y = gz; }
            if (!(Gr.flags & GR_RG_IN2)) { q.x23.x = gx; q.y23.x = gy; q.z23.x = gz; }
            if (!(Gr.flags & GR_RG_IN3)) { q.x23.y = gx; q.y23.y = gy; q.z23.y = gz; }
        }
        GrP4 v;
        v.x01 = q.x01 - gr_v2(gx); v.y01 = q.y01 - gr_v2(gy); v.z01 = q.z01 - gr_v2(gz);
        v.x23 = q.x23 - gr_v2(gx); v.y23 = q.y23 - gr_v2(gy); v.z23 = q.z23 - gr_v2(gz);
        gr_res_image_pair(v.x01, v.y01, v.z01, B, boxp);
        gr_res_image_pair(v.x23, v.y23, v.z23, B, boxp);
        return v;
    };

    // WHO ADDS THE WAVE RECORDS UP.  Until round 5 the wave whose arrival completed the count did -- by construction the slowest wave of its
    // workgroup at that moment, and the extra work kept it the slowest: measured (-DGR_EXP_STEPTIME), one wave of a workgroup ran half of its
    // combines (an even share is an eighth; the younger wave of every SIMD pair twice as often as the older), every other wave polled for
    // records in 45-50 % of its fits, and the pace of the launch is the pace of its slowest wave.  Now (GR_RES_LATE_COMBINE) arrivals are only
    // counted -- cumulatively: slot s is complete for its use u when cnt_s[s] = n_waves (u + 1), nothing is ever reset -- and the record of
    // frame f is put together by the FIRST wave that passes its own hand-over of frame f + 1 and finds frame f complete and unclaimed
    // (done_s[s]: u -> u + 1 by compare-and-swap): a wave that is ahead does the chore.  The last wave to pass finds every earlier arrival
    // in, so somebody always does; the stream's last frame is claimed in the first turn after it.  (A slot is reused 8 frames later: a wave
    // that writes it then has fitted frame f, so frame f's record -- and with it this combine -- was complete long before.)  Measured:
    // combines spread evenly (the busiest wave 21 %), fits that poll 45 % -> 2 %, RMSD-fit 4.12 -> 4.04 us per frame, atoms_center 4.31 -> 4.21.
#ifndef GR_RES_LATE_COMBINE
#define GR_RES_LATE_COMBINE 1
#endif
    auto claim = [&](uint32_t f) -> bool {        // wave-uniform: may this wave put the workgroup's record of frame f together?
        const uint32_t rs = f % R, u = f / R;
        // (arrivals and combines of a slot sit side by side: one 64-bit LDS read answers both questions)
        const unsigned long long both = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(cnt_s + 2u * rs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)both != n_waves * (u + 1u) || (uint32_t)(both >> 32) != u) return false;
        uint32_t won = 0u;
        if (lane == 0) { uint32_t expect = u; won = __hip_atomic_compare_exchange_strong(cnt_s + 2u * rs + 1u, &expect, u + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u; }
        if (__builtin_amdgcn_readfirstlane((int)won) == 0) return false;
        gr_lds_acquire();
        return true;
    };
    // ---- the sums stage of frame i (rows in L): the lane's 18 + 12 values -> wave (reduce-scatter) -> workgroup (LDS, the last
    // wave to arrive adds the wave records in wave order) -> the frame's tagged record.  V: `va`, `vb` receive the image vectors.
    // the workgroup's record of frame f: the wave records in wave order, 31 tagged words
    // (the lane's number is taken afresh inside the combines, and the record leaves through a buffer store off a wave-uniform base: with the
    //  lane's LDS offsets and 64-bit record address as loop invariants the compiler, out of registers, kept them in scratch memory, and their
    //  reload -- a vector memory load -- made the combining wave wait for ALL its outstanding loads, the next turn's rows included)
    auto fresh_lane = []() { uint32_t l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; };
    auto publish_word = [&](uint32_t f, uint32_t l, float v) {
        const __amdgpu_buffer_rsrc_t rr = gr_buf_rsrc(ctl.wgrec + ((size_t)kf(f) * n_pad + wg) * GR_RES_REC_WORDS, 31u * 8u);
        typedef int gr_i2 __attribute__((ext_vector_type(2)));
        gr_i2 w; w.x = __float_as_int(v); w.y = (int)ctl.epoch;
        __builtin_amdgcn_raw_buffer_store_b64(w, rr, (int)(l * 8u), 0, 16 /* sc1: agent scope */);
    };
    auto combine_rmsd = [&](uint32_t f) {
        const uint32_t rs = f % R;
#ifdef GR_EXP_STEPTIME
        st_combines += 1ull;
#endif
        const float *all = wsum + rs * WAVES * 32;
        const uint32_t lane = fresh_lane();
        if (lane < 31u) {
            float v;
            if (n_waves == WAVES) {
                // a full workgroup (all but the last one of a frame): the wave records are requested together -- one LDS round trip
                // instead of a chain of dependent ones (5.44 -> 5.30 us)
                float t[WAVES];
#pragma unroll
                for (uint32_t w = 0; w < WAVES; ++w) t[w] = all[w * 32 + lane];
                v = t[0];
                if (lane < 19u) {
#pragma unroll
                    for (uint32_t w = 1; w < WAVES; ++w) v += t[w];
                } else {
#pragma unroll
                    for (uint32_t w = 1; w < WAVES; ++w) v = gr_fmaxf(v, t[w]);
                }
            } else {
                v = all[lane];
                if (lane < 19u) { for (uint32_t w = 1; w < n_waves; ++w) v += all[w * 32 + lane]; }
                else { for (uint32_t w = 1; w < n_waves; ++w) v = gr_fmaxf(v, all[w * 32 + lane]); }
            }
            publish_word(f, lane, v);
        }
#ifdef GR_EXP_TIMELINE
        if (tl_wg && lane == 0) tls[(f & 127u) * 4u + 0u] = wall_clock64();
#endif
    };
    auto sums = [&](uint32_t i, const Landing &L, const GrBoxU &B, Rows &va, Rows &vb) {
        float s32[32], e32[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) { s32[k] = 0.0f; e32[k] = -3.0e38f; }
        if (wave_sel) {
            const GrBox *boxp = boxes + first_slot + kf(i);
            const float gx = gr_first_f(L.gx), gy = gr_first_f(L.gy), gz = gr_first_f(L.gz);    // wave-uniform: SGPR operands
            // the lane's two groups together (a group that does not exist contributes v = 0 with zero mass and reference): every sum
            // is ONE chain of four packed FMAs over the 8 atoms + one fold, every extent three v_min3 / v_max3 + one v_min / v_max
            const GrP4 a = images(GA, L.a, B, boxp, gx, gy, gz);
            GrP4 b;
            b.x01 = b.y01 = b.z01 = b.x23 = b.y23 = b.z23 = gr_v2(0.0f);
            if (GB.valid) b = images(GB, L.b, B, boxp, gx, gy, gz);
            if (V) { gr_rows_pairs(a, va.r0, va.r1, va.r2); gr_rows_pairs(b, vb.r0, vb.r1, vb.r2); }
            auto fold = [](gr_v2f v) { return v.x + v.y; };
            auto dot8 = [&](gr_v2f w0, gr_v2f w1, gr_v2f w2, gr_v2f w3, gr_v2f x0, gr_v2f x1, gr_v2f x2, gr_v2f x3) {
                return fold(gr_v2_fma(w3, x3, gr_v2_fma(w2, x2, gr_v2_fma(w1, x1, w0 * x0))));
            };
            auto min8 = [](gr_v2f p0, gr_v2f p1, gr_v2f p2, gr_v2f p3) { return gr_fminf(gr_min3f(gr_min3f(gr_min3f(p0.x, p0.y, p1.x), p1.y, p2.x), p2.y, p3.x), p3.y); };
            auto max8 = [](gr_v2f p0, gr_v2f p1, gr_v2f p2, gr_v2f p3) { return gr_fmaxf(gr_max3f(gr_max3f(gr_max3f(p0.x, p0.y, p1.x), p1.y, p2.x), p2.y, p3.x), p3.y); };
            // fractional coordinates of v, axis by axis (c, then b, then a: each axis' values are dead once its four results are
            // formed -- the lane is close to its register budget here): moments + extents feed the image proof (gr_finalize_math)
#ifndef GR_EXP_NOPROOF   /* experiment: what the image proof's share of the instruction stream costs (results are then wrong) */
            {
                const gr_v2f icz = gr_v2(B.icz), ncy = gr_v2(-B.cy), ncx = gr_v2(-B.cx);
                const gr_v2f fc0 = a.z01 * icz, fc1 = a.z23 * icz, fc2 = b.z01 * icz, fc3 = b.z23 * icz;
                gr_v2f ub0 = gr_v2_fma(fc0, ncy, a.y01), ub1 = gr_v2_fma(fc1, ncy, a.y23), ub2 = gr_v2_fma(fc2, ncy, b.y01), ub3 = gr_v2_fma(fc3, ncy, b.y23);
                gr_v2f ua0 = gr_v2_fma(fc0, ncx, a.x01), ua1 = gr_v2_fma(fc1, ncx, a.x23), ua2 = gr_v2_fma(fc2, ncx, b.x01), ua3 = gr_v2_fma(fc3, ncx, b.x23);
                s32[14] = fold((fc0 + fc1) + (fc2 + fc3)); s32[17] = dot8(fc0, fc1, fc2, fc3, fc0, fc1, fc2, fc3);
                e32[8] = -min8(fc0, fc1, fc2, fc3); e32[11] = max8(fc0, fc1, fc2, fc3);
                const gr_v2f iby = gr_v2(B.iby), nbx = gr_v2(-B.bx);
                ub0 *= iby; ub1 *= iby; ub2 *= iby; ub3 *= iby;                       // = fb
                ua0 = gr_v2_fma(ub0, nbx, ua0); ua1 = gr_v2_fma(ub1, nbx, ua1); ua2 = gr_v2_fma(ub2, nbx, ua2); ua3 = gr_v2_fma(ub3, nbx, ua3);
                s32[13] = fold((ub0 + ub1) + (ub2 + ub3)); s32[16] = dot8(ub0, ub1, ub2, ub3, ub0, ub1, ub2, ub3);
                e32[7] = -min8(ub0, ub1, ub2, ub3); e32[10] = max8(ub0, ub1, ub2, ub3);
                const gr_v2f iax = gr_v2(B.iax);
                ua0 *= iax; ua1 *= iax; ua2 *= iax; ua3 *= iax;                       // = fa
                s32[12] = fold((ua0 + ua1) + (ua2 + ua3)); s32[15] = dot8(ua0, ua1, ua2, ua3, ua0, ua1, ua2, ua3);
                e32[6] = -min8(ua0, ua1, ua2, ua3); e32[9] = max8(ua0, ua1, ua2, ua3);
            }
            e32[0] = -min8(a.x01, a.x23, b.x01, b.x23); e32[1] = -min8(a.y01, a.y23, b.y01, b.y23); e32[2] = -min8(a.z01, a.z23, b.z01, b.z23);
            e32[3] = max8(a.x01, a.x23, b.x01, b.x23); e32[4] = max8(a.y01, a.y23, b.y01, b.y23); e32[5] = max8(a.z01, a.z23, b.z01, b.z23);
#endif
            const gr_v2f m0 = gr_v2p(GA.mm.x, GA.mm.y), m1 = gr_v2p(GA.mm.z, GA.mm.w), m2 = gr_v2p(GB.mm.x, GB.mm.y), m3 = gr_v2p(GB.mm.z, GB.mm.w);
            // value k (1..18) lives at s32[k - 1]; value 0 = sum m does not depend on the frame (m_wave)
            s32[0] = dot8(m0, m1, m2, m3, a.x01, a.x23, b.x01, b.x23); s32[1] = dot8(m0, m1, m2, m3, a.y01, a.y23, b.y01, b.y23); s32[2] = dot8(m0, m1, m2, m3, a.z01, a.z23, b.z01, b.z23);
            s32[3] = dot8(GA.P.x01, GA.P.x23, GB.P.x01, GB.P.x23, a.x01, a.x23, b.x01, b.x23); s32[4] = dot8(GA.P.x01, GA.P.x23, GB.P.x01, GB.P.x23, a.y01, a.y23, b.y01, b.y23);
            s32[5] = dot8(GA.P.x01, GA.P.x23, GB.P.x01, GB.P.x23, a.z01, a.z23, b.z01, b.z23);
            s32[6] = dot8(GA.P.y01, GA.P.y23, GB.P.y01, GB.P.y23, a.x01, a.x23, b.x01, b.x23); s32[7] = dot8(GA.P.y01, GA.P.y23, GB.P.y01, GB.P.y23, a.y01, a.y23, b.y01, b.y23);
            s32[8] = dot8(GA.P.y01, GA.P.y23, GB.P.y01, GB.P.y23, a.z01, a.z23, b.z01, b.z23);
            s32[9] = dot8(GA.P.z01, GA.P.z23, GB.P.z01, GB.P.z23, a.x01, a.x23, b.x01, b.x23); s32[10] = dot8(GA.P.z01, GA.P.z23, GB.P.z01, GB.P.z23, a.y01, a.y23, b.y01, b.y23);
            s32[11] = dot8(GA.P.z01, GA.P.z23, GB.P.z01, GB.P.z23, a.z01, a.z23, b.z01, b.z23);
        }
        GR_STEP_STAMP(2);
        const uint32_t rs = i % R;
        float *mine = wsum + (rs * WAVES + wave) * 32;
        if (wave_sel) {
            // 18 sums per frame (the 19th, sum m, does not depend on the frame: m_wave) = a reduce-scatter of 16 + two plain wave
            // sums (a 32-wide scatter would push 14 zeros through its two widest steps); 12 extents = a 16-wide scatter with max
            const float tot = gr_wave_sum_scatter16(s32, lane);                       // values 1..16
            const float t17 = gr_wave_allsum_f32(s32[16]), t18 = gr_wave_allsum_f32(s32[17]);
#ifndef GR_EXP_NOPROOF
            const float emax = gr_wave_max_scatter16(e32, lane);
#else
            const float emax = e32[0];
#endif
            if ((lane & 3u) == 0) mine[1 + (lane >> 2)] = tot;                         // sums 1..16
            if (lane == 1u) { mine[0] = m_wave; mine[17] = t17; mine[18] = t18; }      // sum m (the same every frame), sums 17, 18
            if ((lane & 3u) == 0 && lane < 48u) mine[19 + (lane >> 2)] = emax;         // extents 0..11
        } else if (lane < 31u) {
            mine[lane] = lane < 19u ? 0.0f : -3.0e38f;                                 // nothing of the selection here: the neutral record
        }
        // hand-off to the wave that completes the count: the record is RELEASED before the arrival is counted (LDS-only fence: one
        // s_waitcnt lgkmcnt(0)); the completing wave ACQUIRES before it reads the other waves' records
        gr_lds_release();
#if GR_RES_LATE_COMBINE
        if (lane == 0) (void)__hip_atomic_fetch_add(cnt_s + 2u * rs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (i > 0u && claim(i - 1u)) combine_rmsd(i - 1u);
#else
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt_s + 2u * rs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)old) != n_waves * (i / R + 1u) - 1u) return;
        gr_lds_acquire();
        combine_rmsd(i);
#endif
    };

    // ---- MODE 1, the sums stage of frame i: the Bai-Breen sums of the reference group (iterators.rs:1152-1191,1314-1357).  Every atom's terms are
    // gr_center_atom<1>'s -- the arithmetic of the two-pass path's k_center_sums<1>, bit for bit -- gathered per 4-atom group in f32 and widened
    // to fp64 there, as that kernel does per trip; lanes -> wave (fp64 reduce-scatter) -> LDS (fp64) -> the workgroup's record: every sum as
    // TWO tagged words, v = hi + lo (f32 each: 48 bits of the fp64 value), so that the finalizer's fp64 total differs from the two-pass path's
    // only by the order of fp64 additions.  Word 7 / 15: atoms without position (x is NaN) among ALL the workgroup's atoms -- the frame is
    // then left alone (see the finalizer).
    auto combine_cen = [&](uint32_t f) {          // MODE 1: the workgroup's record of frame f, every fp64 sum as hi + lo words
        const uint32_t rs = f % R;
        const double *all = reinterpret_cast<const double *>(wsum + rs * WAVES * 32);
        const uint32_t lane = fresh_lane();
        if (lane < 31u) {
            float word = 0.0f;
            if (lane < 16u) {
                double v = all[lane & 7u];
                for (uint32_t w = 1; w < n_waves; ++w) v += all[w * 16u + (lane & 7u)];      // wave order
                const float hi = (float)v;
                word = lane < 8u ? hi : (float)(v - (double)hi);
            }
            publish_word(f, lane, word);
        }
    };
    auto sums_cen = [&](uint32_t i, const Landing &L, const GrBoxU &B) {
        double d32[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d32[k] = 0.0;
        const GrBox *boxp = boxes + first_slot + kf(i);
        bool bad = cen_mass_bad;
        const float PI_X2 = 3.14159265358979323846f * 2.0f;       // auxiliary.rs:15 (as k_center_sums)
        const float scx = PI_X2 / B.ax, scy = PI_X2 / B.by, scz = PI_X2 / B.cz;
        // gr_center_atom<1>'s arithmetic for two atoms at a time (gr_bb_angles_pair: the same IEEE operations in the same order, so the same
        // bits), after the positions that lie outside the cell have been wrapped into it (one wave-wide test; gr_wrap is the identity inside)
        GrBbBox bb;
        bb.by = B.by; bb.cz = B.cz; bb.bx = B.bx; bb.cx = B.cx; bb.cy = B.cy; bb.iby = B.iby; bb.icz = B.icz; bb.scx = scx; bb.scy = scy; bb.scz = scz; bb.tric = B.tric;
        auto angles = [&](gr_v2f x, gr_v2f y, gr_v2f z, gr_v2f (&sn)[3], gr_v2f (&cs)[3]) { gr_bb_angles_pair(x, y, z, bb, sn, cs); };
        auto group = [&](const GrResGroup &Gr, const Rows &rw) {
            GrP4 q = gr_pairs_rows(rw.r0, rw.r1, rw.r2);
            // atoms behind the system's last one hold anything: they become the origin (their weight is zero)
            if (__builtin_amdgcn_ballot_w64((Gr.flags & GR_RG_EXALL) == 0u) != 0ull) {
                if (!(Gr.flags & GR_RG_EX0)) { q.x01.x = 0.f; q.y01.x = 0.f; q.z01.x = 0.f; }
                if (!(Gr.flags & GR_RG_EX1)) { q.x01.y = 0.f; q.y01.y = 0.f; q.z01.y = 0.f; }
                if (!(Gr.flags & GR_RG_EX2)) { q.x23.x = 0.f; q.y23.x = 0.f; q.z23.x = 0.f; }
                if (!(Gr.flags & GR_RG_EX3)) { q.x23.y = 0.f; q.y23.y = 0.f; q.z23.y = 0.f; }
            }
            // an atom without position (x is NaN: atom.rs) anywhere in the system: the frame goes back to the host
            bad = bad || (q.x01.x != q.x01.x) || (q.x01.y != q.x01.y) || (q.x23.x != q.x23.x) || (q.x23.y != q.x23.y);
            if (!wave_sel) return;    // no atom of the group in this wave
            const float xl = gr_fminf(gr_min3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y), xh = gr_fmaxf(gr_max3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y);
            const float yl = gr_fminf(gr_min3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y), yh = gr_fmaxf(gr_max3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y);
            const float zl = gr_fminf(gr_min3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y), zh = gr_fmaxf(gr_max3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y);
            const bool inside = (xl >= 0.0f) & (xh <= B.ax) & (yl >= 0.0f) & (yh <= B.by) & (zl >= 0.0f) & (zh <= B.cz);
            if (__builtin_amdgcn_ballot_w64(!inside) != 0ull) {
                float4 r0, r1, r2;
                gr_rows_pairs(q, r0, r1, r2);
                q = gr_res_wrap_slow(r0, r1, r2, 0.0f, 0.0f, 0.0f, boxp);
            }
            gr_v2f s01[3], c01[3], s23[3], c23[3];
            angles(q.x01, q.y01, q.z01, s01, c01);
            angles(q.x23, q.y23, q.z23, s23, c23);
            // the group's 4-atom partials, atom by atom as k_center_sums adds them (p = fma(m, c, p)), widened once
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float pc = fmaf(Gr.mm.w, c23[a].y, fmaf(Gr.mm.z, c23[a].x, fmaf(Gr.mm.y, c01[a].y, fmaf(Gr.mm.x, c01[a].x, 0.0f))));
                const float ps = fmaf(Gr.mm.w, s23[a].y, fmaf(Gr.mm.z, s23[a].x, fmaf(Gr.mm.y, s01[a].y, fmaf(Gr.mm.x, s01[a].x, 0.0f))));
                d32[a] += (double)pc; d32[3 + a] += (double)ps;
            }
        };
        if (i < n_turns) {
            group(GA, L.a);
            if (GB.valid) group(GB, L.b);
        }
        d32[7] = bad ? 1.0 : 0.0;
        GR_STEP_STAMP(2);
        const uint32_t rs = i % R;
        double *mine = reinterpret_cast<double *>(wsum + (rs * WAVES + wave) * 32);     // 16 doubles per wave record
        const double dt = gr_wave_sum_scatter8_f64(d32, lane);                           // lane l: the wave total of value l >> 3 (no LDS crossbar)
        if ((lane & 7u) == 0) mine[lane >> 3] = dt;
        gr_lds_release();
#if GR_RES_LATE_COMBINE
        if (lane == 0) (void)__hip_atomic_fetch_add(cnt_s + 2u * rs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (i > 0u && claim(i - 1u)) combine_cen(i - 1u);
#else
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt_s + 2u * rs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)old) != n_waves * (i / R + 1u) - 1u) return;
        gr_lds_acquire();
        combine_cen(i);
#endif
    };

    // V fit of one group: R v + t0, sum w |R q - p|^2 (rmsd.rs:592-599; pad atoms weigh nothing), + reference COM
    auto fit_group_v = [&](const GrResGroup &Gr, const Rows &rw, const GrResRot &T, float t0x, float t0y, float t0z, Rows &o, float &rs) {
        const GrP4 v = gr_pairs_rows(rw.r0, rw.r1, rw.r2);
        GrP4 n;
        n.x01 = gr_v2_fma(gr_v2(T.r02), v.z01, gr_v2_fma(gr_v2(T.r01), v.y01, gr_v2_fma(gr_v2(T.r00), v.x01, gr_v2(t0x))));
        n.y01 = gr_v2_fma(gr_v2(T.r12), v.z01, gr_v2_fma(gr_v2(T.r11), v.y01, gr_v2_fma(gr_v2(T.r10), v.x01, gr_v2(t0y))));
        n.z01 = gr_v2_fma(gr_v2(T.r22), v.z01, gr_v2_fma(gr_v2(T.r21), v.y01, gr_v2_fma(gr_v2(T.r20), v.x01, gr_v2(t0z))));
        n.x23 = gr_v2_fma(gr_v2(T.r02), v.z23, gr_v2_fma(gr_v2(T.r01), v.y23, gr_v2_fma(gr_v2(T.r00), v.x23, gr_v2(t0x))));
        n.y23 = gr_v2_fma(gr_v2(T.r12), v.z23, gr_v2_fma(gr_v2(T.r11), v.y23, gr_v2_fma(gr_v2(T.r10), v.x23, gr_v2(t0y))));
        n.z23 = gr_v2_fma(gr_v2(T.r22), v.z23, gr_v2_fma(gr_v2(T.r21), v.y23, gr_v2_fma(gr_v2(T.r20), v.x23, gr_v2(t0z))));
        {
            const gr_v2f w01 = WMASS ? gr_v2p(Gr.mm.x, Gr.mm.y) : gr_v2p(Gr.ww.x, Gr.ww.y), w23 = WMASS ? gr_v2p(Gr.mm.z, Gr.mm.w) : gr_v2p(Gr.ww.z, Gr.ww.w);
            gr_v2f dx = n.x01 - Gr.P.x01, dy = n.y01 - Gr.P.y01, dz = n.z01 - Gr.P.z01;
            gr_v2f part = w01 * gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz));
            dx = n.x23 - Gr.P.x23; dy = n.y23 - Gr.P.y23; dz = n.z23 - Gr.P.z23;
            part = gr_v2_fma(w23, gr_v2_fma(dx, dx, gr_v2_fma(dy, dy, dz * dz)), part);
            rs += part.x + part.y;
        }
        n.x01 += gr_v2(cx); n.y01 += gr_v2(cy); n.z01 += gr_v2(cz); n.x23 += gr_v2(cx); n.y23 += gr_v2(cy); n.z23 += gr_v2(cz);
        gr_rows_pairs(n, o.r0, o.r1, o.r2);
    };

    // ---- the fit stage of frame j: `rv` = the frame's record as requested earlier (lanes 0..15); rows / image vectors as they were parked.
    // The record was requested one turn ago -- BEFORE this turn's row request -- so the common case (it has arrived) needs only "all but
    // the loads issued since".  The compiler can count that only when the first look is not the head of the polling loop (whose
    // re-requests are younger than everything: a merged loop head waits for vmcnt(0), i.e. for the rows just requested) and when the
    // record is taken apart into scalars on each path separately (a register read behind the join inherits the slow path's wait).
    struct Rec { int status; float sx, sy, sz, r00, r10, r20, r01, r11, r21, r02, r12, r22, t0x, t0y, t0z; };
    auto take = [&](unsigned long long rv) {
        Rec r;
        r.status = __builtin_amdgcn_readlane((int)(uint32_t)rv, 0);
        r.sx = gr_lane_f(rv, 1); r.sy = gr_lane_f(rv, 2); r.sz = gr_lane_f(rv, 3);
        r.r00 = gr_lane_f(rv, 4); r.r10 = gr_lane_f(rv, 5); r.r20 = gr_lane_f(rv, 6); r.r01 = gr_lane_f(rv, 7); r.r11 = gr_lane_f(rv, 8); r.r21 = gr_lane_f(rv, 9);
        r.r02 = gr_lane_f(rv, 10); r.r12 = gr_lane_f(rv, 11); r.r22 = gr_lane_f(rv, 12);
        r.t0x = gr_lane_f(rv, 13); r.t0y = gr_lane_f(rv, 14); r.t0z = gr_lane_f(rv, 15);
        return r;
    };
    auto fit = [&](uint32_t j, unsigned long long rv, const Rows &ra, const Rows &rb, const GrBoxU &B) {
        Rec rec;
#ifdef GR_EXP_TIMELINE
        uint32_t tl_polls = 0;
        if (tl_wg && wave == 0 && lane == 0) tls[(j & 127u) * 4u + 1u] = wall_clock64();
#endif
        if (__builtin_amdgcn_ballot_w64(lane < 16u && (uint32_t)(rv >> 32) != ctl.epoch) == 0ull) {
            rec = take(rv);
            asm volatile("; record had arrived");
        } else {
            uint32_t polls = 0;
            const unsigned long long t0 = wall_clock64();
            __builtin_amdgcn_s_setprio(0);                             // a wave that is ahead waits below the waves it shares the SIMD with
            do {
                if (++polls > GR_RES_PATIENCE || ((polls & 255u) == 0 && (gr_ld_agent(ctl.abort) != 0u || wall_clock64() - t0 > ctl.patience_ticks))) { if (lane == 0) gr_st_agent(ctl.abort, 1u); bail = true; return; }
                __builtin_amdgcn_s_sleep(GR_RES_SLEEP);
                rv = request_rec(j);
            } while (__builtin_amdgcn_ballot_w64(lane < 16u && (uint32_t)(rv >> 32) != ctl.epoch) != 0ull);
            rec = take(rv);
            asm volatile("; record polled for");
#ifdef GR_EXP_STEPTIME
            st_acc[7] += 1ull;
#endif
#ifdef GR_EXP_TIMELINE
            tl_polls = polls;
#endif
        }
#ifdef GR_EXP_TIMELINE
        if (tl_wg && wave == 0 && lane == 0) { tls[(j & 127u) * 4u + 2u] = tl_polls; tls[(j & 127u) * 4u + 3u] = wall_clock64(); }
#endif
        set_prio();
        GR_STEP_STAMP(4);
        const int status = rec.status;
        float rs = 0.0f;
        Rows oa, ob;
        oa.r0 = oa.r1 = oa.r2 = ob.r0 = ob.r1 = ob.r2 = zero4;
        if (MODE == 1) {
            // x <- wrap(x + shift) for every atom of the system (k_translate_wrap: iterators.rs:1520-1553): the one-turn closed form per axis
            // (c, then b, then a) is the general wrap's own arithmetic whenever no stage turns more than once and every result lies inside
            // (0, L] -- one wave-wide test; otherwise the wave redoes its atoms with gr_wrap itself
            if (status == 0) {
                const GrBox *boxp = boxes + first_slot + kf(j);
                auto center_group = [&](const GrResGroup &Gr, const Rows &rw, Rows &o) {
                    GrP4 q = gr_pairs_rows(rw.r0, rw.r1, rw.r2);
                    q.x01 += gr_v2(rec.sx); q.y01 += gr_v2(rec.sy); q.z01 += gr_v2(rec.sz); q.x23 += gr_v2(rec.sx); q.y23 += gr_v2(rec.sy); q.z23 += gr_v2(rec.sz);
                    // (the pad atoms behind the system's last one keep what they hold, see below: here they sit at the box centre, so that whatever
                    //  they hold -- zeros, as a rule: a point on a face -- never sends their wave through the general wrap, turn after turn,
                    //  with every other workgroup of the launch waiting for that wave's next record)
                    if (__builtin_amdgcn_ballot_w64((Gr.flags & GR_RG_EXALL) == 0u) != 0ull) {
                        if (!(Gr.flags & GR_RG_EX0)) { q.x01.x = B.bcx; q.y01.x = B.bcy; q.z01.x = B.bcz; }
                        if (!(Gr.flags & GR_RG_EX1)) { q.x01.y = B.bcx; q.y01.y = B.bcy; q.z01.y = B.bcz; }
                        if (!(Gr.flags & GR_RG_EX2)) { q.x23.x = B.bcx; q.y23.x = B.bcy; q.z23.x = B.bcz; }
                        if (!(Gr.flags & GR_RG_EX3)) { q.x23.y = B.bcx; q.y23.y = B.bcy; q.z23.y = B.bcz; }
                    }
                    float kmax = 0.0f;
                    auto wrap1 = [&](gr_v2f &x, gr_v2f &y, gr_v2f &z) {
                        gr_v2f k = gr_v2_floor(z * gr_v2(B.icz));
                        kmax = gr_max3f(kmax, __builtin_fabsf(k.x), __builtin_fabsf(k.y));
                        x = gr_v2_fma(-k, gr_v2(B.cx), x); y = gr_v2_fma(-k, gr_v2(B.cy), y); z = gr_v2_fma(-k, gr_v2(B.cz), z);
                        k = gr_v2_floor(y * gr_v2(B.iby));
                        kmax = gr_max3f(kmax, __builtin_fabsf(k.x), __builtin_fabsf(k.y));
                        x = gr_v2_fma(-k, gr_v2(B.bx), x); y = gr_v2_fma(-k, gr_v2(B.by), y);
                        k = gr_v2_floor(x * gr_v2(B.iax));
                        kmax = gr_max3f(kmax, __builtin_fabsf(k.x), __builtin_fabsf(k.y));
                        x = gr_v2_fma(-k, gr_v2(B.ax), x);
                    };
                    wrap1(q.x01, q.y01, q.z01);
                    wrap1(q.x23, q.y23, q.z23);
                    const float xl = gr_fminf(gr_min3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y), xh = gr_fmaxf(gr_max3f(q.x01.x, q.x01.y, q.x23.x), q.x23.y);
                    const float yl = gr_fminf(gr_min3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y), yh = gr_fmaxf(gr_max3f(q.y01.x, q.y01.y, q.y23.x), q.y23.y);
                    const float zl = gr_fminf(gr_min3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y), zh = gr_fmaxf(gr_max3f(q.z01.x, q.z01.y, q.z23.x), q.z23.y);
                    const bool ok = (xl > 0.0f) & (xh <= B.ax) & (yl > 0.0f) & (yh <= B.by) & (zl > 0.0f) & (zh <= B.cz) & (kmax <= 1.0f);
                    if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) q = gr_res_wrap_slow(rw.r0, rw.r1, rw.r2, rec.sx, rec.sy, rec.sz, boxp);
                    gr_rows_pairs(q, o.r0, o.r1, o.r2);
                    // the pad atoms behind the system's last one keep what they hold (k_translate_wrap does not touch them)
                    if (__builtin_amdgcn_ballot_w64((Gr.flags & GR_RG_EXALL) == 0u) != 0ull) {
                        if (!(Gr.flags & GR_RG_EX0)) { o.r0.x = rw.r0.x; o.r0.z = rw.r0.z; o.r1.x = rw.r1.x; }
                        if (!(Gr.flags & GR_RG_EX1)) { o.r0.y = rw.r0.y; o.r0.w = rw.r0.w; o.r1.y = rw.r1.y; }
                        if (!(Gr.flags & GR_RG_EX2)) { o.r1.z = rw.r1.z; o.r2.x = rw.r2.x; o.r2.z = rw.r2.z; }
                        if (!(Gr.flags & GR_RG_EX3)) { o.r1.w = rw.r1.w; o.r2.y = rw.r2.y; o.r2.w = rw.r2.w; }
                    }
                };
                center_group(GA, ra, oa);
                if (GB.valid) center_group(GB, rb, ob);
            }
        } else if (status == 0) {
            GrResRot T;
            T.r00 = rec.r00; T.r10 = rec.r10; T.r20 = rec.r20; T.r01 = rec.r01; T.r11 = rec.r11; T.r21 = rec.r21;
            T.r02 = rec.r02; T.r12 = rec.r12; T.r22 = rec.r22;
            if (V) {
                const float t0x = rec.t0x, t0y = rec.t0y, t0z = rec.t0z;
                T.sx = T.sy = T.sz = 0.f;
                fit_group_v(GA, ra, T, t0x, t0y, t0z, oa, rs);
                if (GB.valid) fit_group_v(GB, rb, T, t0x, t0y, t0z, ob, rs);
            } else {
                T.sx = rec.sx; T.sy = rec.sy; T.sz = rec.sz;
                const GrBox *boxp = boxes + first_slot + kf(j);
                gr_res_fit_group<WMASS>(GA, ra.r0, ra.r1, ra.r2, T, B, boxp, cx, cy, cz, oa.r0, oa.r1, oa.r2, rs);
                if (GB.valid) gr_res_fit_group<WMASS>(GB, rb.r0, rb.r1, rb.r2, T, B, boxp, cx, cy, cz, ob.r0, ob.r1, ob.r2, rs);
            }
        }
        // The six stores are issued on EVERY path -- a frame that was not closed gets a resource of zero records, a group B that does
        // not exist the offset nowhere: the hardware drops them -- so that the number of memory operations between the row request of
        // this turn and the first use of those rows (next turn's sums stage) is the same whatever happened here, and the compiler's
        // s_waitcnt before that use is exact instead of a lower bound that drains part of the prefetch.
        {
#ifdef GR_EXP_NOSTORE
            const __amdgpu_buffer_rsrc_t out = gr_buf_rsrc(frames + (size_t)(first_slot + kf(j)) * frame_stride, 0u);
#else
            const __amdgpu_buffer_rsrc_t out = gr_buf_rsrc(frames + (size_t)(first_slot + kf(j)) * frame_stride, status == 0 ? slot_bytes : 0u);
#endif
            gr_buf_store_stream(out, offA, oa.r0); gr_buf_store_stream(out, offA + 1024u, oa.r1); gr_buf_store_stream(out, offA + 2048u, oa.r2);
            gr_buf_store_stream(out, offB, ob.r0); gr_buf_store_stream(out, offB + 1024u, ob.r1); gr_buf_store_stream(out, offB + 2048u, ob.r2);
        }
        n_fitted = j + 1u;
        if (MODE == 1) return;                                    // (no sum to hand over)
        // the wave's share of sum w |R q - p|^2: the lane's eight atoms in f32, the wave in f32 (no LDS crossbar) -- and straight to memory,
        // one fp64 word per wave and frame: k_rmsd_close adds the waves' words of a frame in a fixed order.  (Until round 5 the waves met in
        // LDS and the last one to arrive added the eight up: a fence, an atomic and a chain of LDS reads on the workgroup's slowest wave,
        // every turn.)
        const float wtot = gr_wave_allsum_f32(rs);
        if (lane == 0) fit_partials[((size_t)kf(j) * ctl.wgs_frame + wg) * WAVES + wave] = (double)wtot;
    };

    // ---- the walk: iteration i = fit of frame i - K, then sums of frame i.  A frame (its rows; V: its image vectors) waits for its
    // record with group A in LDS slot i % K and group B in register set i % K.  The register set is picked by a scalar branch
    // ladder (12 moves out, 12 in): a queue that moves up by one set every iteration costs 60 moves (5.30 -> 5.27 us), and naming
    // the sets by unrolling the loop K times does not fit the instruction cache.  The sets are six separately named variables and
    // every copy is followed by a comment-only asm statement that differs per branch: without them the optimiser merges the six
    // copies into one copy from a selected ADDRESS and the sets end up in scratch memory (measured: 7.3 us per frame).
    Landing L0, L1;
    L0.a.r0 = L0.a.r1 = L0.a.r2 = L0.b.r0 = L0.b.r1 = L0.b.r2 = zero4; L0.gx = L0.gy = L0.gz = 0.f;
    L1 = L0;
    Rows Q0, Q1, Q2, Q3, Q4, Q5;
    Q0.r0 = Q0.r1 = Q0.r2 = zero4; Q1 = Q0; Q2 = Q0; Q3 = Q0; Q4 = Q0; Q5 = Q0;
    unsigned long long rv = 0ull;
    gate(0);
    if (n_turns) request(0, L0);
    const uint32_t n_iter = n_turns + K;
    const GrBoxU B0 = gr_box_uniform(boxes + first_slot);
    auto lds_put = [&](uint32_t slot, const Rows &rw) { park[(slot * 3 + 0) * LANES + tid] = rw.r0; park[(slot * 3 + 1) * LANES + tid] = rw.r1; park[(slot * 3 + 2) * LANES + tid] = rw.r2; };
    auto lds_get = [&](uint32_t slot) { Rows rw; rw.r0 = park[(slot * 3 + 0) * LANES + tid]; rw.r1 = park[(slot * 3 + 1) * LANES + tid]; rw.r2 = park[(slot * 3 + 2) * LANES + tid]; return rw; };
    // Order of a step, FL (round 5; the host takes it where the streaming workgroups fill the chip): the sums of frame i FIRST, then the fit of frame i - K.  A frame's record is the end of a chain --
    // every workgroup's sums -> their records visible to a finalizer -> its tree -> the closing arithmetic -> the record visible here:
    // 17-20 us (tools/timeline_bench.sh) -- and with the fit at the head of the step the chain had (K - 1) turns: workgroup 0 found
    // the record missing in 30-50 % of its turns and polled, a memory round trip per look.  Sums first publishes a turn's record ~0.35
    // turns earlier and needs the old one ~0.65 turns later: the chain has K turns.  Registers: the image vectors of frame i wait in the
    // landing registers the sums have just emptied while frame i - K is fitted, and are parked behind it (the slot is free then).
    auto step_sums_first = [&](uint32_t i, Landing &cur, Landing &nxt) {
        // the record this step's fit needs: requested before everything else of the step (the wait for it then leaves the rows
        // requested below out); unconditional -- the record of the stream's first frame when there is nothing to fit -- see below
        rv = request_rec(i >= K && i < n_iter ? i - K : 0u);
#ifdef GR_EXP_NOLOAD
        if (i < 2) request(i + 1, nxt);
#else
        GR_STEP_STAMP(6);
        gate(i + 1);
        GR_STEP_STAMP(0);
        request(i + 1, nxt);
#endif
        balance(i);
        const GrBoxU Bf = UBOX ? B0 : gr_box_uniform(boxes + first_slot + kf(i >= K ? i - K : 0u));
        const GrBoxU Bs = UBOX ? B0 : gr_box_uniform(boxes + first_slot + kf(i < n_turns ? i : 0u));
        Rows va = cur.a, vb = cur.b;                            // what gets parked: the rows, or (V: set by sums) the image vectors
        const uint32_t ps = i % K;
        GR_STEP_STAMP(1);
        if (i < n_turns) { if constexpr (MODE == 1) sums_cen(i, cur, Bs); else sums(i, cur, Bs, va, vb); }
#if GR_RES_LATE_COMBINE
        // (the stream's last frame has no hand-over after it: its record is put together in the first turn behind it)
        if (i == n_turns && n_turns > 0u && claim(n_turns - 1u)) { if constexpr (MODE == 1) combine_cen(n_turns - 1u); else combine_rmsd(n_turns - 1u); }
#endif
        GR_STEP_STAMP(3);
        if (i >= K) {
            Rows qb;
            if (ps == 0u) { qb = Q0; asm volatile("; set 0 out"); } else if (ps == 1u) { qb = Q1; asm volatile("; set 1 out"); } else if (ps == 2u) { qb = Q2; asm volatile("; set 2 out"); }
            else if (ps == 3u) { qb = Q3; asm volatile("; set 3 out"); } else if (ps == 4u) { qb = Q4; asm volatile("; set 4 out"); } else { qb = Q5; asm volatile("; set 5 out"); }
            fit(i - K, rv, lds_get(ps), qb, Bf);
            if (bail) return;
        }
        GR_STEP_STAMP(5);
        if (i < n_turns) {                                      // (the slot's old content has been read by the fit above)
            lds_put(ps, va);
            if (ps == 0u) { Q0 = vb; asm volatile("; set 0 in"); } else if (ps == 1u) { Q1 = vb; asm volatile("; set 1 in"); } else if (ps == 2u) { Q2 = vb; asm volatile("; set 2 in"); }
            else if (ps == 3u) { Q3 = vb; asm volatile("; set 3 in"); } else if (ps == 4u) { Q4 = vb; asm volatile("; set 4 in"); } else { Q5 = vb; asm volatile("; set 5 in"); }
        }
    };
    // ... and the order of rounds 2-4, the fit first: still the faster one for a single stream that leaves a quarter of the chip idle
    auto step_fit_first = [&](uint32_t i, Landing &cur, Landing &nxt) {
#ifdef GR_EXP_NOLOAD
        if (i < 2) request(i + 1, nxt);
#elif !defined(GR_EXP_LATE_REQUEST)
        gate(i + 1);
        request(i + 1, nxt);
#endif
        balance(i);
        // both boxes of the iteration are requested here (scalar loads): they arrive while the record is checked
        const GrBoxU Bf = UBOX ? B0 : gr_box_uniform(boxes + first_slot + kf(i >= K ? i - K : 0u));
        const GrBoxU Bs = UBOX ? B0 : gr_box_uniform(boxes + first_slot + kf(i < n_turns ? i : 0u));
        Rows va = cur.a, vb = cur.b;                            // what gets parked: the rows, or (V: set by sums) the image vectors
        const uint32_t ps = i % K;
        if (i >= K) {
            Rows qb;
            if (ps == 0u) { qb = Q0; asm volatile("; set 0 out"); } else if (ps == 1u) { qb = Q1; asm volatile("; set 1 out"); } else if (ps == 2u) { qb = Q2; asm volatile("; set 2 out"); }
            else if (ps == 3u) { qb = Q3; asm volatile("; set 3 out"); } else if (ps == 4u) { qb = Q4; asm volatile("; set 4 out"); } else { qb = Q5; asm volatile("; set 5 out"); }
            fit(i - K, rv, lds_get(ps), qb, Bf);
            if (bail) return;
        }
        // (unconditional -- the record of the stream's first frame when there is nothing to fit next turn: an `if` here makes the new
        // value depend on the old register, and the compiler waits for the rows just requested before it reads that)
        rv = request_rec(i + 1 >= K && i + 1 < n_iter ? i + 1 - K : 0u);
#ifdef GR_EXP_LATE_REQUEST
        request(i + 1, nxt);                                    // (experiment: the next frame's rows requested after this step's stores)
#endif
        if (i < n_turns) {
            if (MODE == 1) sums_cen(i, cur, Bs);
            else if (V) sums(i, cur, Bs, va, vb);               // (the slot's old content has been read by the fit above)
            lds_put(ps, va);
            if (ps == 0u) { Q0 = vb; asm volatile("; set 0 in"); } else if (ps == 1u) { Q1 = vb; asm volatile("; set 1 in"); } else if (ps == 2u) { Q2 = vb; asm volatile("; set 2 in"); }
            else if (ps == 3u) { Q3 = vb; asm volatile("; set 3 in"); } else if (ps == 4u) { Q4 = vb; asm volatile("; set 4 in"); } else { Q5 = vb; asm volatile("; set 5 in"); }
            if (!V && MODE != 1) sums(i, cur, Bs, va, vb);
        }
#if GR_RES_LATE_COMBINE
        // (the stream's last frame has no hand-over after it: its record is put together in the first turn behind it)
        if (i == n_turns && n_turns > 0u && claim(n_turns - 1u)) { if constexpr (MODE == 1) combine_cen(n_turns - 1u); else combine_rmsd(n_turns - 1u); }
#endif
    };
    auto step = [&](uint32_t i, Landing &cur, Landing &nxt) { if constexpr (FL) step_sums_first(i, cur, nxt); else step_fit_first(i, cur, nxt); };
    for (uint32_t i = 0; i < n_iter && !bail; i += 2) {
        step(i, L0, L1);
        if (!bail && i + 1 < n_iter) step(i + 1, L1, L0);
    }
    if (lane == 0) {
        ctl.progress[wg_all * WAVES + wave] = n_fitted;
        // what the host sets the metronome by: when the last wave left, how many slots were reached late / waited for
        (void)__hip_atomic_fetch_max(reinterpret_cast<unsigned long long *>(ctl.abort + 6), (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n_late) (void)__hip_atomic_fetch_add(ctl.abort + 8, n_late, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n_waited) (void)__hip_atomic_fetch_add(ctl.abort + 9, n_waited, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef GR_EXP_STEPTIME
        if (ctl.dbg) {
            unsigned long long *o = ctl.dbg + (size_t)(wg_all * WAVES + wave) * 4u;
            unsigned long long tot = 0; for (int k = 0; k < 7; ++k) tot += st_acc[k];
            o[0] = st_acc[4]; o[1] = st_acc[7] | (st_combines << 32);
            o[2] = (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u) | (simd_of[wave] << 8);
            o[3] = tot;
        }
        if ((wg_all == 0u || wg_all == ctl.n_stream / 2u) && (wave == 0u || wave == 5u)) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(ctl.abort + 16) + ((wg_all ? 2u : 0u) + (wave ? 1u : 0u)) * 8u;
            for (int k = 0; k < 8; ++k) o[k] = st_acc[k];
        }
#endif
        if (clk_wave) {
            const unsigned long long dc = (unsigned long long)clock64() - clk_c0, dw = (unsigned long long)wall_clock64() - clk_w0;
            gr_st_agent(ctl.abort + 10, (uint32_t)(dc >> 8)); gr_st_agent(ctl.abort + 11, (uint32_t)(dw >> 8));   // shader / device clock ticks of the walk, / 256
        }
    }
#ifdef GR_EXP_TIMELINE
    // workgroup 0's stamps of its last 256 turns: frame f's publication of the workgroup record -> slot 0, first look at the frame's
    // record -> slot 6 (slot 5 is the finalizer's), polls -> word 0 of the frame's row is left alone (slot 0 = publication)
    if (tl_wg && wave == 0 && !bail) {
        for (uint32_t t = (n_turns > 128u ? n_turns - 128u : 0u) + lane; t < n_turns; t += 64u) {
            const size_t f = kf(t);
            ctl.tl[f * 8 + 0] = tls[(t & 127u) * 4u + 0u];
            ctl.tl[f * 8 + 6] = tls[(t & 127u) * 4u + 1u] | (tls[(t & 127u) * 4u + 2u] << 48);
        }
    }
#endif
}
