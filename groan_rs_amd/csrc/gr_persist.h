// gr_persist.h -- persistent, software-pipelined RMSD-fit kernel for gfx950.
//
// The three-kernel path (k_rmsd_accum -> k_rmsd_finalize -> k_fit) moves 37 MB of HBM traffic per 1e6-atom frame: the
// frame is read twice (12 MB each) because the rotation is only known after the whole frame has been reduced, and the
// 256 MiB Infinity Cache does not make the second read cheaper (tools/mall_bench.hip).  This kernel reads every frame
// ONCE: the chip's LDS (256 CUs x 160 KiB = 40 MB) holds a whole frame (12 MB) several times over, so each workgroup
// keeps ITS slice of the frame in LDS between the accumulate and the fit:
//
//   one workgroup per CU, workgroup w owns tiles [w*ntiles/nwg, (w+1)*ntiles/nwg) of every frame (its slice of the reference
//   coordinates and masses therefore stays in its XCD's L2 for the whole launch)
//   A(f): load my slice of frame f (12 B/atom from HBM) -> LDS ring slot f % D and, on the way through the registers,
//         the single-pass sums of gr_flush4; reduce in the workgroup; publish my partial record; arrival counter;
//         the LAST workgroup to arrive sums the 256 records, runs the image proof + Kabsch (gr_finalize_math) and
//         publishes R / shift / rmsd for the frame, then a ready flag
//   C(f): wait for the ready flag of frame f, transform my LDS slice, write it out (12 B/atom to HBM)
//   each workgroup runs  A(0) A(1) .. | A(k) C(k-D+1) | .. C(F-1)   with D = 2 or 3 ring slots, so the ~4 us a frame
//   needs to become "ready" (slowest arrival + finalize) are spent loading and summing the next frames.
// HBM traffic: 24 MB/frame instead of 37.
//
// Inter-workgroup hand-off (MI355X guide, "Valid forms" / Guideline 16, all-sc1 form): every handed-off byte is written
// with agent-scope relaxed atomic stores (global_store ... sc1) by ONE wave, that wave drains them (s_waitcnt vmcnt(0)),
// then one lane signals (atomic add / flag store, agent scope); the consumer polls with agent-scope relaxed loads and
// reads the payload with agent-scope relaxed loads after its poll has matched (other waves after a workgroup barrier).
// No placement assumption: which XCD a workgroup runs on only affects L2 locality.
// Progress: a workgroup only ever waits for "all workgroups finished A(f)" of a frame it has itself finished, and the
// finalizer never waits, so the workgroup that is furthest behind is never blocked; all workgroups must be co-resident,
// which one-workgroup-per-CU (LDS > 80 KiB each) guarantees and a start-up handshake verifies.  Every spin is bounded
// (wall clock); on a timeout the abort flag makes every workgroup leave and the host reports an error.
#pragma once
#include "gr_kernels.h"

#define GR_PS_THREADS 512
#define GR_PS_WAVES (GR_PS_THREADS / 64)
#define GR_PS_REC 48                      // doubles per partial record: 32 sums, 12 extents, 4 spare
#define GR_PS_TIMEOUT_TICKS 20000000ull   // s_memrealtime runs at 100 MHz: 0.2 s

struct GrPersistArgs {
    float *frames; size_t frame_stride; uint32_t first_slot, n_frames, n_atoms;
    const float *masses; GrSel sel; const GrBox *boxes; GrPlanDev plan;
    GrFrameState *state;       // [n_frames] in: status prefilled by the host, out: results
    double *partials;          // [n_frames][n_wg][GR_PS_REC]
    uint32_t *sync;            // [0] present counter, [1] abort flag, [2 + f] arrivals of frame f, [2 + n_frames + f] ready flag of frame f
    uint32_t tiles_per_wg, depth;
    unsigned long long *trace; // debugging (GR_PS_TRACE): [n_frames][n_wg][GR_PS_TRACE_N] s_memrealtime stamps, or nullptr
};
#define GR_PS_TRACE_N 10
// stamp k of (frame f, workgroup w): 0 A start, 1 sums done, 2 arrived, 3 C wait start, 4 ready seen, 5 C done,
// finalizer only: 6 records summed, 7 math done, 8 published
#define GR_PS_STAMP(f, k) do { if (A.trace && lane == 0 && (wave == 0)) A.trace[((size_t)(f) * nwg + w) * GR_PS_TRACE_N + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

__device__ __forceinline__ uint32_t gr_ld_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gr_st_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double gr_ld_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gr_st_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gr_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// bounded wait (one lane): true when *flag >= want, false on abort / timeout (timeout raises the abort flag)
__device__ __forceinline__ bool gr_wait_ge(const uint32_t *flag, uint32_t want, uint32_t *abort_flag) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (gr_ld_u32(flag) >= want) return true;
        if (gr_ld_u32(abort_flag)) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > GR_PS_TIMEOUT_TICKS) { gr_st_u32(abort_flag, 1u); return false; }
        __builtin_amdgcn_s_sleep(16);
    }
}

// Wave reduce-scatter of 32 doubles: afterwards lane l holds the wave total of value (l >> 1).
// Each step halves the values a lane still carries (it sends the half its partner keeps): 16+8+4+2+1+1 exchanges
// instead of 32 x 6.  Steps are template instances so every register-array index is a compile-time constant.
template <int HALF, int MASK>
__device__ __forceinline__ void gr_rs_step(double (&a)[32], const uint32_t lane) {
    const bool hi = (lane & MASK) != 0;
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
        const double send = hi ? a[k] : a[k + HALF];
        const double keep = hi ? a[k + HALF] : a[k];
        a[k] = keep + __shfl_xor(send, MASK, 64);
    }
}
__device__ __forceinline__ double gr_wave_reduce_scatter32(double (&a)[32], const uint32_t lane) {
    gr_rs_step<16, 32>(a, lane); gr_rs_step<8, 16>(a, lane); gr_rs_step<4, 8>(a, lane); gr_rs_step<2, 4>(a, lane); gr_rs_step<1, 2>(a, lane);
    a[0] += __shfl_xor(a[0], 1, 64);
    return a[0];
}
// the same with max over 16 floats: lane l ends with the wave maximum of value (l >> 2)
template <int HALF, int MASK>
__device__ __forceinline__ void gr_ms_step(float (&a)[16], const uint32_t lane) {
    const bool hi = (lane & MASK) != 0;
#pragma unroll
    for (int k = 0; k < HALF; ++k) {
        const float send = hi ? a[k] : a[k + HALF];
        const float keep = hi ? a[k + HALF] : a[k];
        a[k] = fmaxf(keep, __shfl_xor(send, MASK, 64));
    }
}
__device__ __forceinline__ float gr_wave_max_scatter16(float (&a)[16], const uint32_t lane) {
    gr_ms_step<8, 32>(a, lane); gr_ms_step<4, 16>(a, lane); gr_ms_step<2, 8>(a, lane); gr_ms_step<1, 4>(a, lane);
    a[0] = fmaxf(a[0], __shfl_xor(a[0], 2, 64));
    a[0] = fmaxf(a[0], __shfl_xor(a[0], 1, 64));
    return a[0];
}

__global__ __launch_bounds__(GR_PS_THREADS) void k_rmsd_fit_persist(const GrPersistArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gr_ps_lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w = blockIdx.x, nwg = gridDim.x;
    const uint32_t T = A.tiles_per_wg, D = A.depth, F = A.n_frames;
    // ---- LDS carve-up
    float4 *ring = reinterpret_cast<float4 *>(gr_ps_lds);                                    // [D][T][192]
    GrBox *boxes_l = reinterpret_cast<GrBox *>(gr_ps_lds + (size_t)D * T * GR_TILE_F4 * 16);   // [D]
    double *wsum = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(boxes_l) + D * ((sizeof(GrBox) + 15) & ~15u));   // [waves][32]
    float *wmax = reinterpret_cast<float *>(wsum + GR_PS_WAVES * 32);                         // [waves][16]
    double *fin = reinterpret_cast<double *>(wmax + GR_PS_WAVES * 16);                        // [GR_PS_REC]
    uint32_t *st_l = reinterpret_cast<uint32_t *>(fin + GR_PS_REC);                            // [D][32] published frame state (as dwords)
    int *flag_l = reinterpret_cast<int *>(st_l + D * 32);                                      // [2]
    uint32_t *present = A.sync, *abort_flag = A.sync + 1, *arrive = A.sync + 2, *ready = A.sync + 2 + F;

    const uint32_t ntiles = (A.n_atoms + 255u) >> 8;
    const uint32_t t_begin = (uint32_t)(((uint64_t)w * ntiles) / nwg), t_end = (uint32_t)(((uint64_t)(w + 1) * ntiles) / nwg);   // my tiles: balanced split, at most T
    const uint32_t first = A.sel.start, last = A.sel.start + A.sel.n, g0 = A.sel.g0 << 6;
    const float4 *p4 = reinterpret_cast<const float4 *>(A.plan.p);
    const float4 *m4 = reinterpret_cast<const float4 *>(A.masses);
    const float4 *w4 = reinterpret_cast<const float4 *>(A.plan.w);
    const bool wm = A.plan.w_is_mass != 0;

    // ---- residency handshake: nothing is written to the frames unless every workgroup is running
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(present, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag_l[0] = gr_wait_ge(present, nwg, abort_flag) ? 1 : 0;
    }
    __syncthreads();
    if (!flag_l[0]) return;
    __syncthreads();

    for (uint32_t k = 0; k < F + D - 1; ++k) {
        // ================================================================ A(k)
        if (k < F) {
            const uint32_t f = k, slot = f % D;
            GR_PS_STAMP(f, 0);
            float4 *buf = ring + (size_t)slot * T * GR_TILE_F4;
            float *xyz = A.frames + (size_t)(A.first_slot + f) * A.frame_stride;
            const float4 *f4 = reinterpret_cast<const float4 *>(xyz);
            __syncthreads();                                            // every wave has left C(f - D), the last reader of this box slot
            gr_stage_box(boxes_l + slot, A.boxes + A.first_slot + f);   // includes a workgroup barrier
            const GrBox &box = boxes_l[slot];
            GrLaneAcc L;
            L.reset();
            GrFrameConst fc;
            fc.gx = xyz[3 * (size_t)first]; fc.gy = xyz[3 * (size_t)first + 1]; fc.gz = xyz[3 * (size_t)first + 2];
            fc.sx = fc.sy = fc.sz = 0.f;
            fc.iax = box.iax; fc.iby = box.iby; fc.icz = box.icz; fc.rws2 = box.r_ws * box.r_ws; fc.tric = !box.ortho; fc.wm = wm;
            for (uint32_t t = t_begin + wave; t < t_end; t += GR_PS_WAVES) {
                const uint32_t g = (t << 6) + lane;
                const float4 a = f4[3 * (size_t)g], b = f4[3 * (size_t)g + 1], c = f4[3 * (size_t)g + 2];
                float4 *dst = buf + (size_t)(t - t_begin) * GR_TILE_F4 + 3 * lane;
                dst[0] = a; dst[1] = b; dst[2] = c;                     // memory order: float4 j of the tile <-> f4[t*192 + j]
                const uint32_t i = g << 2;
                if (i + 3 >= first && i < last) {                        // the group touches the selection
                    const size_t pg = (size_t)(g - g0);
                    const float4 pa = p4[3 * pg], pb = p4[3 * pg + 1], pc = p4[3 * pg + 2];
                    const float4 mm = m4[g];
                    const float4 ww = wm ? mm : w4[pg];
                    GrA4 q;
                    gr_unpack4(q, a, b, c, pa, pb, pc, mm, ww, i, first, last);
                    if (i >= first && i + 3 < last) gr_flush4<0>(L, q, false, box, fc); else gr_flush4<0>(L, q, true, box, fc);
                }
            }
            L.close(wm);
            GR_PS_STAMP(f, 1);
            if (L.bad_pos != GR_NOIDX || L.bad_mass != GR_NOIDX) L.acc[0] = __builtin_nan("");   // -> poisoned -> multi-pass path names the atom
            // ---- workgroup reduction: wave reduce-scatter -> LDS -> wave 0
            const double tot = gr_wave_reduce_scatter32(L.acc, lane);
            float ex[16];
#pragma unroll
            for (int a = 0; a < 3; ++a) { ex[a] = -L.mn[a]; ex[3 + a] = L.mx[a]; ex[6 + a] = -L.fmn[a]; ex[9 + a] = L.fmx[a]; }
            ex[12] = ex[13] = ex[14] = ex[15] = -3.0e38f;
            const float emax = gr_wave_max_scatter16(ex, lane);
            if ((lane & 1u) == 0) wsum[wave * 32 + (lane >> 1)] = tot;
            if ((lane & 3u) == 0) wmax[wave * 16 + (lane >> 2)] = emax;
            __syncthreads();
            if (wave == 0) {
                double *rec = A.partials + ((size_t)f * nwg + w) * GR_PS_REC;
                if (lane < 32) {
                    double s = 0.0;
#pragma unroll
                    for (int q = 0; q < GR_PS_WAVES; ++q) s += wsum[q * 32 + lane];
                    gr_st_f64(rec + lane, s);
                } else if (lane < 48) {
                    float m = -3.0e38f;
#pragma unroll
                    for (int q = 0; q < GR_PS_WAVES; ++q) m = fmaxf(m, wmax[q * 16 + (lane - 32)]);
                    gr_st_f64(rec + lane, (double)m);
                }
                gr_drain_stores();                                        // every lane's record stores have left before the signal
                uint32_t old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(arrive + f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                old = __builtin_amdgcn_readfirstlane(old);
                GR_PS_STAMP(f, 2);
                if (old == nwg - 1) {
                    // ---- I am the last to arrive: close frame f
                    double s[32];
                    float e[16];
#pragma unroll
                    for (int q = 0; q < 32; ++q) s[q] = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) e[q] = -3.0e38f;
                    for (uint32_t r = lane; r < nwg; r += 64) {
                        const double *src = A.partials + ((size_t)f * nwg + r) * GR_PS_REC;
#pragma unroll
                        for (int q = 0; q < 32; ++q) s[q] += gr_ld_f64(src + q);
#pragma unroll
                        for (int q = 0; q < 12; ++q) e[q] = fmaxf(e[q], (float)gr_ld_f64(src + 32 + q));
                    }
                    GR_PS_STAMP(f, 6);
                    const double ts = gr_wave_reduce_scatter32(s, lane);
                    const float te = gr_wave_max_scatter16(e, lane);
                    if ((lane & 1u) == 0) fin[lane >> 1] = ts;
                    if ((lane & 3u) == 0) fin[32 + (lane >> 2)] = (double)te;
                    gr_wave_sync();
                    if (lane == 0) {
                        GrFrameState st = {};
                        st.err_index = GR_NOIDX;
                        st.status = (int)gr_ld_u32(reinterpret_cast<const uint32_t *>(&A.state[f].status));   // host pre-check result
                        if (st.status == 0) {
                            float mn[3], mx[3], fmn[3], fmx[3];
                            for (int a = 0; a < 3; ++a) { mn[a] = -(float)fin[32 + a]; mx[a] = (float)fin[35 + a]; fmn[a] = -(float)fin[38 + a]; fmx[a] = (float)fin[41 + a]; }
                            const double g[3] = { fc.gx, fc.gy, fc.gz };
                            gr_finalize_math<0>(fin, mn, mx, fmn, fmx, GR_NOIDX, GR_NOIDX, A.boxes[A.first_slot + f], A.plan, g, A.sel.n, st);
                        }
                        GR_PS_STAMP(f, 7);
                        uint32_t *dst = reinterpret_cast<uint32_t *>(A.state + f);
                        const uint32_t *srcw = reinterpret_cast<const uint32_t *>(&st);
                        for (uint32_t q = 0; q < sizeof(GrFrameState) / 4; ++q) gr_st_u32(dst + q, srcw[q]);
                        gr_drain_stores();
                        gr_st_u32(ready + f, 1u);
                        GR_PS_STAMP(f, 8);
                    }
                }
            }
        }
        // ================================================================ C(k - D + 1)
        if (k + 1 >= D) {
            const uint32_t f = k + 1 - D, slot = f % D;
            GR_PS_STAMP(f, 3);
            if (threadIdx.x == 0) flag_l[1] = gr_wait_ge(ready + f, 1u, abort_flag) ? 1 : 0;
            __syncthreads();
            const int okflag = flag_l[1];
            GR_PS_STAMP(f, 4);
            if (okflag && threadIdx.x < sizeof(GrFrameState) / 4)
                st_l[slot * 32 + threadIdx.x] = gr_ld_u32(reinterpret_cast<const uint32_t *>(A.state + f) + threadIdx.x);
            __syncthreads();
            if (!okflag) return;   // abort: uniform over the workgroup
            const GrFrameState &st = *reinterpret_cast<const GrFrameState *>(st_l + slot * 32);
            if (st.status == 0) {
                const GrBox &box = boxes_l[slot];
                float4 *buf = ring + (size_t)slot * T * GR_TILE_F4;
                float4 *f4 = reinterpret_cast<float4 *>(A.frames + (size_t)(A.first_slot + f) * A.frame_stride);
                const float sx = st.shift[0], sy = st.shift[1], sz = st.shift[2];
                const float r00 = st.R[0], r10 = st.R[1], r20 = st.R[2], r01 = st.R[3], r11 = st.R[4], r21 = st.R[5], r02 = st.R[6], r12 = st.R[7], r22 = st.R[8];
                const float cx = A.plan.ref_com[0], cy = A.plan.ref_com[1], cz = A.plan.ref_com[2];
                auto tf = [&](float &x, float &y, float &z) {
                    x += sx; y += sy; z += sz;
                    gr_wrap(x, y, z, box);
                    x -= box.bcx; y -= box.bcy; z -= box.bcz;
                    const float nx = r00 * x + r01 * y + r02 * z;
                    const float ny = r10 * x + r11 * y + r12 * z;
                    const float nz = r20 * x + r21 * y + r22 * z;
                    x = nx + cx; y = ny + cy; z = nz + cz;
                };
                for (uint32_t t = t_begin + wave; t < t_end; t += GR_PS_WAVES) {
                    float4 *tile = buf + (size_t)(t - t_begin) * GR_TILE_F4;
                    float4 a = tile[3 * lane], b = tile[3 * lane + 1], c = tile[3 * lane + 2];
                    tf(a.x, a.y, a.z); tf(a.w, b.x, b.y); tf(b.z, b.w, c.x); tf(c.y, c.z, c.w);
                    gr_tile_store(f4 + (size_t)t * GR_TILE_F4, tile, lane, a, b, c);
                }
            }
            GR_PS_STAMP(f, 5);
        }
    }
}

// dynamic LDS bytes of k_rmsd_fit_persist for (tiles per workgroup, ring depth)
static inline size_t gr_persist_lds_bytes(uint32_t T, uint32_t D) {
    return (size_t)D * T * GR_TILE_F4 * 16 + (size_t)D * ((sizeof(GrBox) + 15) & ~15u) + GR_PS_WAVES * 32 * 8 + GR_PS_WAVES * 16 * 4 + GR_PS_REC * 8 +
           (size_t)D * 32 * 4 + 16;
}
