// gr_persist.h -- persistent, software-pipelined RMSD-fit kernel for gfx950.
//
// The three-kernel path (k_rmsd_accum -> k_rmsd_finalize -> k_fit) moves 37 MB of HBM traffic per 1e6-atom frame: the
// frame is read twice (12 MB each) because the rotation is only known after the whole frame has been reduced, and the
// 256 MiB Infinity Cache does not make the second read cheaper (tools/mall_bench.hip).  This kernel reads every frame
// ONCE: the chip's LDS (256 CUs x 160 KiB = 40 MB) holds a whole frame (12 MB) three times over, so each workgroup
// keeps ITS slice of the frame in LDS between the sums and the fit:
//
//   one workgroup of 16 waves per CU; workgroup w owns tiles [w*ntiles/nwg, (w+1)*ntiles/nwg) (<= 16) of EVERY frame and
//   each wave owns one of them, so a wave's reference coordinates, masses and weights are loaded once per launch and
//   live in registers
//   A(f): wave: load my tile of frame f (coalesced, 12 B/atom from HBM) -> LDS ring slot f % D, read it back transposed
//         (4 atoms per lane), the sums of gr_flush4<0, LITE>, wave reduce-scatter (f32) -> LDS, LDS arrival counter
//         comm wave (wave 0): when all waves of the workgroup have arrived: sum them (fp64), publish the workgroup's
//         record, global arrival counter (fire and forget)
//   the LAST workgroup of the grid owns no tiles: its 16 waves are the finalizers, wave v for frames f = v (mod 16):
//         wait for all arrivals of frame f, sum the 255 records, image proof + rotation (gr_finalize_math<0, LITE>),
//         publish shift / R for the frame, then a ready flag.  (Letting the last workgroup to ARRIVE do this instead
//         makes the slowest workgroup slower still: it then closes every frame and the whole pipeline runs at its pace.)
//   C(f): comm wave: wait for the ready flag of frame f, copy the frame state to LDS, raise the LDS ready counter
//         wave: wait for the LDS ready counter, transform my LDS tile, sum w |R q - p|^2 on the way (the reference's
//         final loop, rmsd.rs:592-599), write the tile out (coalesced, 12 B/atom to HBM), one rmsd partial per wave
//   each wave runs  A(0) A(1) .. | A(k) C(k-D+1) | .. C(F-1)  on its own: no workgroup barrier in the loop; the time a
//   frame needs to become "ready" is spent on the next frames.  k_rmsd_close turns the partials into the rmsd.
// HBM traffic: 24 MB/frame instead of 37.
//
// Hand-off between workgroups (MI355X guide, all-sc1 form): every handed-off byte is written with agent-scope relaxed
// atomic stores (global_store ... sc1) by ONE wave, that wave drains them (s_waitcnt vmcnt(0)), then one lane signals
// (atomic add / flag store, agent scope); the consumer polls with agent-scope loads and reads the payload with sc1
// loads after its poll has matched.  Hand-off inside a workgroup goes through LDS with workgroup-scope release/acquire.
// No placement assumption: which XCD a workgroup runs on only affects L2 locality.
// Progress: a wave only ever waits for "all workgroups finished A(f)" of a frame it has itself finished, and the
// finalizer never waits, so the workgroup that is furthest behind is never blocked; all workgroups must be co-resident,
// which one-workgroup-per-CU (1024 threads, > 80 KiB LDS each) guarantees and a start-up handshake verifies.  Every
// spin is bounded (wall clock); on a timeout the abort flag makes every wave leave and the host reports an error.
#pragma once
#include "gr_kernels.h"

#define GR_PS_THREADS 1024
#define GR_PS_WAVES (GR_PS_THREADS / 64)
#define GR_PS_REC 32                      // doubles per workgroup record: [0..18] sums, [19..30] extents (as maxima), [31] spare
#define GR_PS_NSUM 19                     // sum m, sum m v (3), A (9), moments of the fractional coordinates (6)
#define GR_PS_TIMEOUT_TICKS 20000000ull   // s_memrealtime runs at 100 MHz: 0.2 s

struct GrPersistArgs {
    float *frames; size_t frame_stride; uint32_t first_slot, n_frames, n_atoms;
    const float *masses; GrSel sel; const GrBox *boxes; GrPlanDev plan;
    GrFrameState *state;       // [n_frames] in: status prefilled by the host, out: results (rmsd by k_rmsd_close)
    double *partials;          // [n_frames][n_wg][GR_PS_REC]
    double *rmsd_partials;     // [n_frames][n_wg * GR_PS_WAVES], zeroed by the host
    uint32_t *sync;            // [0] present counter, [1] abort flag, [2 + f] arrivals of frame f, [2 + n_frames + f] ready flag of frame f
    uint32_t tiles_per_wg, depth;
    unsigned long long *trace; // debugging (GR_PS_TRACE): [n_frames][n_wg][GR_PS_TRACE_N] s_memrealtime stamps, or nullptr
};
#define GR_PS_TRACE_N 16
// stamp k of (frame f, workgroup w), comm wave: 0 A start, 1 all waves arrived, 2 arrived globally, 3 C wait start,
// 4 ready seen, 5 C done; finalizer only: 6 records summed, 7 math done, 8 published; wave 15's durations (ticks):
// 9 tile landed in LDS, 10 sums, 11 wave reduce + LDS arrive, 12 wait for the LDS ready counter, 13 transform, 14 store + rmsd
#define GR_PS_LAP(f, k) do { if (A.trace && wave == GR_PS_WAVES - 1) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); if (lane == 0) A.trace[((size_t)(f) * nwg + w) * GR_PS_TRACE_N + (k)] = now_ - lap_; lap_ = now_; } } while (0)
#define GR_PS_STAMP(f, k) do { if (A.trace && lane == 0) A.trace[((size_t)(f) * nwg + w) * GR_PS_TRACE_N + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

__device__ __forceinline__ uint32_t gr_ld_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gr_st_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gr_st_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gr_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t gr_lds_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void gr_lds_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Eight 16-byte agent-scope (sc1) loads in flight, one wait: the compiler waits after every atomic load it emits itself,
// which turns the finalizer's 64 loads per lane into 64 round trips.  Loads p + 1024 q, q = 0 .. 7.
typedef double gr_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gr_ld_sc1_x8(const unsigned char *p, gr_d2 (&v)[8]) {
    const unsigned char *p2 = p + 4096;
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %8, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %2, %8, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %3, %8, off offset:3072 sc1\n\t"
        "global_load_dwordx4 %4, %9, off sc1\n\t"
        "global_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\t"
        "global_load_dwordx4 %7, %9, off offset:3072 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p), "v"(p2)
        : "memory");
}

// bounded wait (one lane): true when *flag >= want, false on abort / timeout (timeout raises the abort flag)
__device__ __forceinline__ bool gr_wait_ge(const uint32_t *flag, uint32_t want, uint32_t *abort_flag) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (gr_ld_u32(flag) >= want) return true;
        if (gr_ld_u32(abort_flag)) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > GR_PS_TIMEOUT_TICKS) { gr_st_u32(abort_flag, 1u); return false; }
        __builtin_amdgcn_s_sleep(8);
    }
}
// the same on an LDS counter; `abort_l` is the workgroup's copy of the abort flag
__device__ __forceinline__ bool gr_wait_lds_ge(const uint32_t *flag, uint32_t want, const uint32_t *abort_l, uint32_t *abort_flag) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t spin = 0;; ++spin) {
        if (gr_lds_ld(flag) >= want) return true;
        if (gr_lds_ld(abort_l)) return false;
        if ((spin & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t0 > GR_PS_TIMEOUT_TICKS) { gr_st_u32(abort_flag, 1u); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
}

// LDS carve-up (bytes), shared by the kernel and the host
struct GrPersistLds {
    uint32_t ring, wsum, wmax, fin, st, ctl, total;
    __host__ __device__ GrPersistLds(uint32_t T, uint32_t D) {
        ring = 0;
        wsum = ring + D * T * GR_TILE_F4 * 16;            // [D][waves][32] float
        wmax = wsum + D * GR_PS_WAVES * 32 * 4;           // [D][waves][16] float
        fin = wmax + D * GR_PS_WAVES * 16 * 4;            // [GR_PS_REC] double
        st = fin + GR_PS_REC * 8 + 128;                   // (+ one GrFrameState of finalizer scratch)  [D][32] dwords: published frame state
        ctl = st + D * 32 * 4;                            // [0] go, [1] abort, [2] frames ready, [4 + s] arrivals of partial slot s
        total = ctl + (4 + D) * 4 + 16;
    }
};

// The cell part of a frame's box (its first 16 dwords) as wave-uniform values: one 64-byte load by 16 lanes, then
// v_readlane.  (Going through `const GrBox &` in memory instead costs a dependent global load per field after every
// store the compiler cannot prove unrelated: ~30 serialized round trips per frame and wave.)  The minimum-image table
// stays in memory: it is touched by the rare atom beyond r_ws only.
__device__ __forceinline__ uint32_t gr_box_head_load(const GrBox *b, uint32_t lane) { return reinterpret_cast<const uint32_t *>(b)[lane & 15u]; }
__device__ __forceinline__ void gr_box_head_unpack(GrBox &h, uint32_t bw) {
    auto F = [&](int q) { return __uint_as_float(__builtin_amdgcn_readlane(bw, q)); };
    h.ax = F(0); h.by = F(1); h.cz = F(2); h.bx = F(3); h.cx = F(4); h.cy = F(5); h.bcx = F(6); h.bcy = F(7); h.bcz = F(8);
    h.iax = F(9); h.iby = F(10); h.icz = F(11); h.r_ws = F(12);
    h.ortho = (int)__builtin_amdgcn_readlane(bw, 13); h.ncand = (int)__builtin_amdgcn_readlane(bw, 14); h.valid = (int)__builtin_amdgcn_readlane(bw, 15);
}

// Sums of one atom of the selection for the persistent kernel: the arithmetic of gr_flush4<0, LITE> with every lane
// holding only its 4 atoms of the frame, so the 4-atom f32 partials ARE the lane's sums (s[0..18]: sum m, sum m v, A,
// moments; e[0..11]: -min v, max v, -min f, max f).  A missing position (NaN x) or mass poisons the sums: the finalizer
// then sends the frame to the multi-pass path, which names the atom in the reference's order.
__device__ __forceinline__ void gr_lite_atom(float (&s)[32], float (&e)[32], float x, float y, float z, float px, float py, float pz, float m,
                                             bool in_sel, const GrBox &box, const GrBox &cand_box, const GrFrameConst &fc) {
    if (!in_sel) return;
    float vx, vy, vz, f_a, f_b, f_c;
    gr_image_about(vx, vy, vz, f_a, f_b, f_c, x, y, z, box, cand_box, fc);
    s[0] += m; s[1] = fmaf(m, vx, s[1]); s[2] = fmaf(m, vy, s[2]); s[3] = fmaf(m, vz, s[3]);
    s[4] = fmaf(px, vx, s[4]); s[5] = fmaf(px, vy, s[5]); s[6] = fmaf(px, vz, s[6]);
    s[7] = fmaf(py, vx, s[7]); s[8] = fmaf(py, vy, s[8]); s[9] = fmaf(py, vz, s[9]);
    s[10] = fmaf(pz, vx, s[10]); s[11] = fmaf(pz, vy, s[11]); s[12] = fmaf(pz, vz, s[12]);
    s[13] += f_a; s[14] += f_b; s[15] += f_c;
    s[16] = fmaf(f_a, f_a, s[16]); s[17] = fmaf(f_b, f_b, s[17]); s[18] = fmaf(f_c, f_c, s[18]);
    e[0] = gr_fmaxf(e[0], -vx); e[1] = gr_fmaxf(e[1], -vy); e[2] = gr_fmaxf(e[2], -vz);
    e[3] = gr_fmaxf(e[3], vx); e[4] = gr_fmaxf(e[4], vy); e[5] = gr_fmaxf(e[5], vz);
    e[6] = gr_fmaxf(e[6], -f_a); e[7] = gr_fmaxf(e[7], -f_b); e[8] = gr_fmaxf(e[8], -f_c);
    e[9] = gr_fmaxf(e[9], f_a); e[10] = gr_fmaxf(e[10], f_b); e[11] = gr_fmaxf(e[11], f_c);
}

// The last-arriving workgroup's closing step for frame f (one lane; cold code kept out of line so its registers do not
// weigh on the streaming loop): fin = the 19 sums + 12 maxima of the whole frame.
__device__ __noinline__ void gr_persist_close(const GrFrameState *state_f, const float *xyz, const GrBox *box, GrPlanDev plan, uint32_t first, uint32_t n_sel,
                                              const double *fin, GrFrameState *out) {
    GrFrameState st = {};
    st.err_index = GR_NOIDX;
    st.status = (int)gr_ld_u32(reinterpret_cast<const uint32_t *>(&state_f->status));   // host pre-check result
    if (st.status == 0) {
        double acc[GR_ACC_K];
        for (int q = 0; q < GR_ACC_K; ++q) acc[q] = 0.0;
        for (int q = 0; q < 13; ++q) acc[q] = fin[q];
        for (int q = 0; q < 6; ++q) acc[26 + q] = fin[13 + q];
        float mn[3], mx[3], fmn[3], fmx[3];
        for (int a = 0; a < 3; ++a) {
            mn[a] = -(float)fin[GR_PS_NSUM + a]; mx[a] = (float)fin[GR_PS_NSUM + 3 + a];
            fmn[a] = -(float)fin[GR_PS_NSUM + 6 + a]; fmx[a] = (float)fin[GR_PS_NSUM + 9 + a];
        }
        const double gc[3] = { xyz[3 * (size_t)first], xyz[3 * (size_t)first + 1], xyz[3 * (size_t)first + 2] };
        gr_finalize_math<0, true>(acc, mn, mx, fmn, fmx, GR_NOIDX, GR_NOIDX, *box, plan, gc, n_sel, st);
    }
    *out = st;
}

__global__ __launch_bounds__(GR_PS_THREADS) void k_rmsd_fit_persist(const GrPersistArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gr_ps_lds[];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t w = blockIdx.x, nwg = gridDim.x;
    const uint32_t T = A.tiles_per_wg, D = A.depth, F = A.n_frames;
    const GrPersistLds L(T, D);
    float4 *ring = reinterpret_cast<float4 *>(gr_ps_lds + L.ring);
    float *wsum = reinterpret_cast<float *>(gr_ps_lds + L.wsum);
    float *wmax = reinterpret_cast<float *>(gr_ps_lds + L.wmax);
    uint32_t *st_l = reinterpret_cast<uint32_t *>(gr_ps_lds + L.st);
    uint32_t *ctl = reinterpret_cast<uint32_t *>(gr_ps_lds + L.ctl);
    uint32_t *go_l = ctl, *abort_l = ctl + 1, *ready_l = ctl + 2, *cnt_l = ctl + 4;
    uint32_t *present = A.sync, *abort_flag = A.sync + 1, *arrive = A.sync + 2, *ready = A.sync + 2 + F;

    const uint32_t ntiles = (A.n_atoms + 255u) >> 8, ncomp = nwg - 1;   // the last workgroup finalizes, the others stream
    const bool finalizer = w == ncomp;
    const uint32_t t_begin = finalizer ? 0u : (uint32_t)(((uint64_t)w * ntiles) / ncomp), t_end = finalizer ? 0u : (uint32_t)(((uint64_t)(w + 1) * ntiles) / ncomp);
    const uint32_t ntw = t_end - t_begin;                       // my tiles: 1 .. 16, one per wave, wave 15 first
    const uint32_t j = (GR_PS_WAVES - 1) - wave;                // my tile within the workgroup
    const bool has_tile = j < ntw, comm = wave == 0;
    if (!has_tile && !comm && !finalizer) return;
    const uint32_t t = t_begin + j, g = (t << 6) + lane, i0 = g << 2;   // my tile, float4 group and first atom
    const uint32_t first = A.sel.start, last = A.sel.start + A.sel.n, g0 = A.sel.g0 << 6;
    const bool wm = A.plan.w_is_mass != 0;
    const bool touches = has_tile && (i0 + 3 >= first) && (i0 < last), interior = (i0 >= first) && (i0 + 3 < last);

    if (comm && lane == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; for (uint32_t s = 0; s < D; ++s) cnt_l[s] = 0; }
    // my reference coordinates, masses and weights: the same for every frame
    float4 pa = make_float4(0, 0, 0, 0), pb = pa, pc = pa, mm = pa, ww = pa;
    if (touches) {
        const size_t pg = (size_t)(g - g0);
        const float4 *p4 = reinterpret_cast<const float4 *>(A.plan.p);
        pa = p4[3 * pg]; pb = p4[3 * pg + 1]; pc = p4[3 * pg + 2];
        mm = reinterpret_cast<const float4 *>(A.masses)[g];
        ww = wm ? mm : reinterpret_cast<const float4 *>(A.plan.w)[pg];
    }
    __syncthreads();   // the only workgroup barrier (waves without work have left already): the control words are initialised

    // ---- residency handshake: nothing is written to the frames unless every workgroup is running
    {
        uint32_t ok = 0;
        if (comm) {
            if (lane == 0) {
                __hip_atomic_fetch_add(present, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = gr_wait_ge(present, nwg, abort_flag) ? 1u : 0u;
                if (!ok) gr_lds_st(abort_l, 1u);
                gr_lds_st(go_l, 1u);
            }
        } else if (lane == 0) {
            ok = (gr_wait_lds_ge(go_l, 1u, abort_l, abort_flag) && !gr_lds_ld(abort_l)) ? 1u : 0u;
        }
        if (!__builtin_amdgcn_readfirstlane(ok)) return;
    }


    if (finalizer) {
        // ================================================================ finalizer waves: frames f = wave (mod 16)
        for (uint32_t f = wave; f < F; f += GR_PS_WAVES) {
            uint32_t ok = 0;
            if (lane == 0) ok = gr_wait_ge(arrive + f, ncomp, abort_flag) ? 1u : 0u;
            if (!__builtin_amdgcn_readfirstlane(ok)) return;
            GR_PS_STAMP(f, 2);
            // Lane l sums 16-byte chunk (l & 15) of records (l >> 4) + 4 i; doubles [0, 19) of a record are sums,
            // [19, 31) maxima (chunk 9 holds one of each)
            const uint32_t chunk = lane & 15u;
            const bool xmax = chunk * 2 >= GR_PS_NSUM, ymax = chunk * 2 + 1 >= GR_PS_NSUM;
            double ax = xmax ? -3.0e38 : 0.0, ay = ymax ? -3.0e38 : 0.0;
            const unsigned char *base = reinterpret_cast<const unsigned char *>(A.partials + (size_t)f * nwg * GR_PS_REC) +
                                        (size_t)(lane >> 4) * (GR_PS_REC * 8) + chunk * 16;
            for (uint32_t i = 0; i < nwg / 4; i += 8) {          // records 1024 bytes apart, eight per batch (nwg % 32 == 0)
                gr_d2 v[8];
                gr_ld_sc1_x8(base + (size_t)i * 1024, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if ((lane >> 4) + 4u * (i + q) >= ncomp) continue;   // the finalizer workgroup has no record
                    ax = xmax ? fmax(ax, v[q].x) : ax + v[q].x;
                    ay = ymax ? fmax(ay, v[q].y) : ay + v[q].y;
                }
            }
            // lanes l, l^16, l^32, l^48 hold the same chunk of different records: fixed combination order
            { const double ox = __shfl_xor(ax, 16, 64), oy = __shfl_xor(ay, 16, 64); ax = xmax ? fmax(ax, ox) : ax + ox; ay = ymax ? fmax(ay, oy) : ay + oy; }
            { const double ox = __shfl_xor(ax, 32, 64), oy = __shfl_xor(ay, 32, 64); ax = xmax ? fmax(ax, ox) : ax + ox; ay = ymax ? fmax(ay, oy) : ay + oy; }
            double *fin = reinterpret_cast<double *>(gr_ps_lds) + (size_t)wave * (GR_PS_REC + 16);   // this wave's scratch (the ring is unused here)
            if (lane < 16) { fin[2 * lane] = ax; fin[2 * lane + 1] = ay; }
            gr_wave_sync();
            GR_PS_STAMP(f, 6);
            if (lane == 0) {
                GrFrameState *stp = reinterpret_cast<GrFrameState *>(fin + GR_PS_REC);
                gr_persist_close(A.state + f, A.frames + (size_t)(A.first_slot + f) * A.frame_stride, A.boxes + A.first_slot + f, A.plan, first, A.sel.n, fin, stp);
                GR_PS_STAMP(f, 7);
                uint32_t *dst = reinterpret_cast<uint32_t *>(A.state + f);
                const uint32_t *srcw = reinterpret_cast<const uint32_t *>(stp);
                for (uint32_t q = 0; q < sizeof(GrFrameState) / 4; ++q) gr_st_u32(dst + q, srcw[q]);
                gr_drain_stores();
                gr_st_u32(ready + f, 1u);
                GR_PS_STAMP(f, 8);
            }
            gr_wave_sync();
        }
        return;
    }

    for (uint32_t k = 0; k < F + D - 1; ++k) {
        // ================================================================ A(k)
        if (k < F) {
            const uint32_t f = k, slot = f % D;
            if (comm) GR_PS_STAMP(f, 0);
            unsigned long long lap_ = (A.trace && wave == GR_PS_WAVES - 1) ? __builtin_amdgcn_s_memrealtime() : 0ull;
            if (has_tile) {
                float4 *tile = ring + ((size_t)slot * T + j) * GR_TILE_F4;
                float *xyz = A.frames + (size_t)(A.first_slot + f) * A.frame_stride;
                const float4 *src = reinterpret_cast<const float4 *>(xyz) + (size_t)t * GR_TILE_F4;
                const float4 r0 = src[lane], r1 = src[lane + 64], r2 = src[lane + 128];
                const GrBox &cbox = A.boxes[A.first_slot + f];           // only its minimum-image table is read through this
                const uint32_t bw = gr_box_head_load(&cbox, lane);
                const uint32_t gw = reinterpret_cast<const uint32_t *>(xyz)[3 * (size_t)first + (lane & 3u)];   // provisional centre: the first atom of the selection
                GrBox box;
                gr_box_head_unpack(box, bw);
                GrFrameConst fc;
                fc.gx = __uint_as_float(__builtin_amdgcn_readlane(gw, 0)); fc.gy = __uint_as_float(__builtin_amdgcn_readlane(gw, 1)); fc.gz = __uint_as_float(__builtin_amdgcn_readlane(gw, 2));
                fc.sx = fc.sy = fc.sz = 0.f;
                fc.iax = box.iax; fc.iby = box.iby; fc.icz = box.icz; fc.rws2 = box.r_ws * box.r_ws; fc.tric = !box.ortho; fc.wm = wm;
                tile[lane] = r0; tile[lane + 64] = r1; tile[lane + 128] = r2;   // memory order: float4 q of the tile <-> src[q]
                gr_wave_sync();
                GR_PS_LAP(f, 9);
                float s32[32], e32[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) { s32[q] = 0.0f; e32[q] = -3.0e38f; }
                if (touches) {
                    const float4 a = tile[3 * lane], b = tile[3 * lane + 1], c = tile[3 * lane + 2];
                    gr_lite_atom(s32, e32, a.x, a.y, a.z, pa.x, pa.y, pa.z, mm.x, interior || (i0 >= first && i0 < last), box, cbox, fc);
                    gr_lite_atom(s32, e32, a.w, b.x, b.y, pa.w, pb.x, pb.y, mm.y, interior || (i0 + 1 >= first && i0 + 1 < last), box, cbox, fc);
                    gr_lite_atom(s32, e32, b.z, b.w, c.x, pb.z, pb.w, pc.x, mm.z, interior || (i0 + 2 >= first && i0 + 2 < last), box, cbox, fc);
                    gr_lite_atom(s32, e32, c.y, c.z, c.w, pc.y, pc.z, pc.w, mm.w, interior || (i0 + 3 >= first && i0 + 3 < last), box, cbox, fc);
                }
                if (A.trace) asm volatile("" : "+v"(s32[0]), "+v"(s32[18]), "+v"(e32[0]));   // keep the lap honest: the sums are done here
                GR_PS_LAP(f, 10);
                const float tot = gr_wave_sum_scatter32(s32, lane);
                const float emax = gr_wave_max_scatter16(e32, lane);
                if ((lane & 1u) == 0) wsum[(slot * GR_PS_WAVES + wave) * 32 + (lane >> 1)] = tot;
                if ((lane & 3u) == 0) wmax[(slot * GR_PS_WAVES + wave) * 16 + (lane >> 2)] = emax;
                gr_wave_sync();
                if (lane == 0) __hip_atomic_fetch_add(cnt_l + slot, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            GR_PS_LAP(f, 11);
            if (comm) {
                // ---- all my workgroup's waves have delivered frame f: publish the workgroup record
                uint32_t ok = 0;
                if (lane == 0) ok = gr_wait_lds_ge(cnt_l + slot, ntw * (f / D + 1u), abort_l, abort_flag) ? 1u : 0u;
                if (!__builtin_amdgcn_readfirstlane(ok)) { if (lane == 0) gr_lds_st(abort_l, 1u); return; }
                GR_PS_STAMP(f, 1);
                double *rec = A.partials + ((size_t)f * nwg + w) * GR_PS_REC;
                const uint32_t w0 = GR_PS_WAVES - ntw;                     // waves w0 .. 15 own tiles
                if (lane < GR_PS_NSUM) {
                    double s = 0.0;
                    for (uint32_t q = w0; q < GR_PS_WAVES; ++q) s += (double)wsum[(slot * GR_PS_WAVES + q) * 32 + lane];
                    gr_st_f64(rec + lane, s);
                } else if (lane < GR_PS_NSUM + 12) {
                    float m = -3.0e38f;
                    for (uint32_t q = w0; q < GR_PS_WAVES; ++q) m = gr_fmaxf(m, wmax[(slot * GR_PS_WAVES + q) * 16 + (lane - GR_PS_NSUM)]);
                    gr_st_f64(rec + lane, (double)m);
                }
                gr_drain_stores();                                        // every lane's record stores have left before the signal
                if (lane == 0) __hip_atomic_fetch_add(arrive + f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // fire and forget
                GR_PS_STAMP(f, 2);
            }
        }
        // ================================================================ C(k - D + 1)
        if (k + 1 >= D) {
            const uint32_t f = k + 1 - D, slot = f % D;
            if (comm) {
                GR_PS_STAMP(f, 3);
                uint32_t ok = 0;
                if (lane == 0) ok = gr_wait_ge(ready + f, 1u, abort_flag) ? 1u : 0u;
                ok = __builtin_amdgcn_readfirstlane(ok);
                if (!ok) { if (lane == 0) gr_lds_st(abort_l, 1u); return; }
                if (lane < sizeof(GrFrameState) / 4) st_l[slot * 32 + lane] = gr_ld_u32(reinterpret_cast<const uint32_t *>(A.state + f) + lane);
                gr_wave_sync();
                if (lane == 0) gr_lds_st(ready_l, f + 1u);
                GR_PS_STAMP(f, 4);
            }
            if (has_tile) {
                unsigned long long lap_ = (A.trace && wave == GR_PS_WAVES - 1) ? __builtin_amdgcn_s_memrealtime() : 0ull;
                if (!comm) {
                    uint32_t ok = 0;
                    if (lane == 0) ok = gr_wait_lds_ge(ready_l, f + 1u, abort_l, abort_flag) ? 1u : 0u;
                    if (!__builtin_amdgcn_readfirstlane(ok)) return;
                }
                GR_PS_LAP(f, 12);
                const uint32_t bw = gr_box_head_load(A.boxes + A.first_slot + f, lane);
                const uint32_t sw = st_l[slot * 32 + (lane & 31u)];       // the frame state as wave-uniform values, too
                auto SF = [&](int q) { return __uint_as_float(__builtin_amdgcn_readlane(sw, q)); };
                const int st_status = (int)__builtin_amdgcn_readlane(sw, 19);
                if (st_status == 0) {
                    GrBox box;
                    gr_box_head_unpack(box, bw);
                    float4 *tile = ring + ((size_t)slot * T + j) * GR_TILE_F4;
                    float4 *dst = reinterpret_cast<float4 *>(A.frames + (size_t)(A.first_slot + f) * A.frame_stride) + (size_t)t * GR_TILE_F4;
                    const float sx = SF(6), sy = SF(7), sz = SF(8);           // GrFrameState: center[3] com[3] shift[3] R[9] rmsd status ...
                    const float r00 = SF(9), r10 = SF(10), r20 = SF(11), r01 = SF(12), r11 = SF(13), r21 = SF(14), r02 = SF(15), r12 = SF(16), r22 = SF(17);
                    const float cx = A.plan.ref_com[0], cy = A.plan.ref_com[1], cz = A.plan.ref_com[2];
                    auto rot = [&](float &x, float &y, float &z) {
                        x += sx; y += sy; z += sz;
                        gr_wrap(x, y, z, box);
                        x -= box.bcx; y -= box.bcy; z -= box.bcz;
                        const float nx = r00 * x + r01 * y + r02 * z;
                        const float ny = r10 * x + r11 * y + r12 * z;
                        const float nz = r20 * x + r21 * y + r22 * z;
                        x = nx; y = ny; z = nz;
                    };
                    float4 a = tile[3 * lane], b = tile[3 * lane + 1], c = tile[3 * lane + 2];
                    rot(a.x, a.y, a.z); rot(a.w, b.x, b.y); rot(b.z, b.w, c.x); rot(c.y, c.z, c.w);
                    float part = 0.0f;
                    if (touches) {
                        auto d2 = [](float x, float y, float z, float px, float py, float pz) { const float dx = x - px, dy = y - py, dz = z - pz; return fmaf(dx, dx, fmaf(dy, dy, dz * dz)); };
                        if (interior) {
                            part = ww.x * d2(a.x, a.y, a.z, pa.x, pa.y, pa.z);
                            part = fmaf(ww.y, d2(a.w, b.x, b.y, pa.w, pb.x, pb.y), part);
                            part = fmaf(ww.z, d2(b.z, b.w, c.x, pb.z, pb.w, pc.x), part);
                            part = fmaf(ww.w, d2(c.y, c.z, c.w, pc.y, pc.z, pc.w), part);
                        } else {
                            if (i0 >= first && i0 < last) part = ww.x * d2(a.x, a.y, a.z, pa.x, pa.y, pa.z);
                            if (i0 + 1 >= first && i0 + 1 < last) part = fmaf(ww.y, d2(a.w, b.x, b.y, pa.w, pb.x, pb.y), part);
                            if (i0 + 2 >= first && i0 + 2 < last) part = fmaf(ww.z, d2(b.z, b.w, c.x, pb.z, pb.w, pc.x), part);
                            if (i0 + 3 >= first && i0 + 3 < last) part = fmaf(ww.w, d2(c.y, c.z, c.w, pc.y, pc.z, pc.w), part);
                        }
                    }
                    a.x += cx; a.y += cy; a.z += cz; a.w += cx; b.x += cy; b.y += cz; b.z += cx; b.w += cy; c.x += cz; c.y += cx; c.z += cy; c.w += cz;
                    if (A.trace) asm volatile("" : "+v"(a.x), "+v"(c.w), "+v"(part));
                    GR_PS_LAP(f, 13);
                    gr_tile_store(dst, tile, lane, a, b, c);
                    const double rs = gr_wave_sum((double)part);
                    if (lane == 0) A.rmsd_partials[((size_t)f * nwg + w) * GR_PS_WAVES + wave] = rs;
                    GR_PS_LAP(f, 14);
                }
            }
            if (comm) GR_PS_STAMP(f, 5);
        }
    }
}
