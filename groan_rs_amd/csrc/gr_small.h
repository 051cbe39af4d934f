// gr_small.h -- ONE per-frame call on a SMALL selection (the reference's per-frame traits on a protein of a few hundred atoms:
// BASELINE configs[0] / [1], `FrameAnalyze::analyze` once per frame, traj_convert.rs:76-83) as ONE dispatch whose result the host
// reads out of host-mapped memory.
//
// The batched kernels answer such a call with a chain of dispatches -- state upload, sums over a handful of workgroups, a finalize
// workgroup with ~30 block-wide reductions, state download -- and a stream synchronisation: 36 us for a centre of mass, 55 us for an
// RMSD of 363 atoms in a 32 817-atom system (tools/latency_trace.py: 29 us of that inside the two kernels), where the reference's
// CPU code needs a few microseconds.  What such a call is made of on this box (tools/microbench/launch_latency.hip): one dispatch +
// hipStreamSynchronize 11.7-12.5 us, every further dispatch or small copy + 2.5-2.8 us, but one dispatch whose kernel stores its
// result into coherent host memory that the host polls 6.2 us -- the end-of-kernel signal and the wake-up are the other half.
// So: a single wave walks the selection, reduces inside the wave (reduce-scatter over the lanes, no barrier), lane 0 closes the
// frame with the SAME closing functions as the batched path (gr_center_close, gr_finalize_math), writes the frame's state to the
// device copy and to the host-mapped copy, fences at system scope and publishes the call's sequence number; the host spins on that
// word (and falls back to hipStreamSynchronize when it does not come).  By then the kernel has nothing left to do: everything it
// read it has read, everything it writes it has written.
// Per-atom arithmetic is the batched kernels' own (gr_center_atom, gr_flush4); only the order of the fp64 additions differs -- so
// EVERY path of a small selection goes through these waves (centre stages of batches and of atoms_center, RMSD batches, the literal redo of
// a frame whose image proof failed: k_center_small_stage / _pbc, k_rmsd_small<0 / 1> with one wave per frame), and a batch keeps
// equalling its per-frame calls bit for bit.
#pragma once
#include "gr_kernels.h"

#define GR_SMALL_MAX_DEFAULT 4096u     /* atoms of a selection the single-wave kernels take (GR_TUNE_SMALL_CALLS: 0 = never) */

__device__ __forceinline__ uint32_t gr_wave_min_u32(uint32_t x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)x, off, 64); x = y < x ? y : x; }
    return x;
}
// lane 0 publishes the frame's state: device copy, host-mapped copy, then -- released at system scope -- the call's sequence number
__device__ __forceinline__ void gr_small_publish(const GrFrameState &st, GrFrameState *state_dev, GrFrameState *state_host, uint32_t *flag_host, uint32_t seq) {
    *state_dev = st;
    *state_host = st;
    __threadfence_system();
    __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one stage of a centre (k_center_sums<KIND> + k_center_finalize) by one wave: the totals end in lane 0, which closes the stage
template <int KIND>
__device__ __forceinline__ void gr_small_center_stage(const float *__restrict__ xyz, const float *__restrict__ masses, const GrSel &sel, const GrBox &box,
                                                      const int weighted, const int mass_first, const int target, GrFrameState &st, double *lds_tot, const uint32_t lane, const int only_status) {
    const float PI_X2 = 3.14159265358979323846f * 2.0f;   // auxiliary.rs:15
    const float scx = KIND == 1 ? PI_X2 / box.ax : 0.f, scy = KIND == 1 ? PI_X2 / box.by : 0.f, scz = KIND == 1 ? PI_X2 / box.cz : 0.f;
    // (the centre of the unwrapping stage is lane 0's: every lane takes it from there)
    const float cx = KIND == 2 ? __shfl(st.center[0], 0, 64) : 0.f, cy = KIND == 2 ? __shfl(st.center[1], 0, 64) : 0.f, cz = KIND == 2 ? __shfl(st.center[2], 0, 64) : 0.f;
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    uint32_t bad_pos = GR_NOIDX, bad_mass = GR_NOIDX;
    // (a single wave has nobody to hide its load latency behind: the loads of eight trips are requested together, their atoms then added
    //  in trip order)
    for (uint32_t j0 = lane; j0 < sel.n; j0 += 8u * 64u) {
        float x[8], y[8], z[8], m[8];
        uint32_t ii[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t j = j0 + (uint32_t)k * 64u;
            const uint32_t jj = j < sel.n ? j : 0u;
            ii[k] = sel.contiguous ? sel.start + jj : sel.idx[jj];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { gr_pos_load(xyz, ii[k], x[k], y[k], z[k]); m[k] = weighted ? masses[ii[k]] : 1.0f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (j0 + (uint32_t)k * 64u >= sel.n) continue;
            float p[7] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
            gr_center_atom<KIND>(ii[k], x[k], y[k], z[k], m[k], weighted, box, scx, scy, scz, cx, cy, cz, bad_pos, bad_mass, p);
#pragma unroll
            for (int q = 0; q < (KIND == 1 ? 7 : 4); ++q) acc[q] += (double)p[q];
        }
    }
    const double t = gr_wave_sum_scatter16_f64(acc, lane);           // lane l: the wave total of value l >> 2
    bad_pos = gr_wave_min_u32(bad_pos); bad_mass = gr_wave_min_u32(bad_mass);
    if ((lane & 3u) == 0u && lane < 32u) lds_tot[lane >> 2] = t;
    gr_wave_sync();
    if (lane == 0) {
        double tot[GR_CEN_K];
#pragma unroll
        for (int k = 0; k < GR_CEN_K; ++k) tot[k] = lds_tot[k];
        gr_center_close(tot, bad_pos, bad_mass, box, KIND, weighted, mass_first, target, sel.n, st, only_status);
    }
    gr_wave_sync();
}

// (The frame's box is read through the uniform, read-only pointer: scalar loads that are in flight together with the first rows --
//  staging it through LDS, as the batched kernels do for their 4-16 waves, would put a load round trip and a fence in front of everything.)
// ONE stage of a centre for a batch of frames, one wave per frame: what center_stage (gr_api.hip) launches instead of
// k_center_sums<KIND> + k_center_finalize when the selection is small -- every caller of a centre stage (the per-frame calls, the
// batched ones, atoms_center's estimate) then adds a small selection's terms in the same order, and "a batch equals its per-frame
// calls bit for bit" (tests/test_gpu_batch_calls.py) keeps holding
__global__ __launch_bounds__(64) void k_center_small_stage(
    const float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, const float *__restrict__ masses, GrSel sel,
    const GrBox *__restrict__ boxes, int kind, int weighted, int mass_first, int target, GrFrameState *state, int only_status,
    uint32_t *__restrict__ fresh_bad = nullptr /* one-frame gr_atoms_center: the state starts fresh HERE (no reset launch before this one) and the
                                                  translate kernel's "first atom without position" words are set for it (no memset launch) */) {
    __shared__ double lds_tot[GR_CEN_K];
    const uint32_t lane = threadIdx.x, frame = blockIdx.x;
    GrFrameState st = {};
    if (fresh_bad) { st.err_index = GR_NOIDX; if (lane < 4u) fresh_bad[4 * frame + lane] = GR_NOIDX; }
    else st = state[frame];
    if (st.status != only_status) return;             // an earlier stage of this frame already failed / not one of the frames asked for
    const GrBox &box = boxes[first_slot + frame];
    const float *xyz = frames + (size_t)(first_slot + frame) * frame_stride;
    // (gr_center_close sees only_status as k_center_finalize does: the AMBIG placement, "the frame's last stage: done")
    if (kind == 0) gr_small_center_stage<0>(xyz, masses, sel, box, weighted, mass_first, target, st, lds_tot, lane, only_status);
    else if (kind == 1) gr_small_center_stage<1>(xyz, masses, sel, box, weighted, mass_first, target, st, lds_tot, lane, only_status);
    else gr_small_center_stage<2>(xyz, masses, sel, box, weighted, mass_first, target, st, lds_tot, lane, only_status);
    if (lane == 0) state[frame] = st;
}

// get_center / get_com of a batch of frames (pbc_center_stages, gr_api.hip): both stages of every frame in one launch, one wave per frame --
// the same two stage functions in the same order as k_center_small below and as two k_center_small_stage launches
__global__ __launch_bounds__(64) void k_center_small_pbc(
    const float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, const float *__restrict__ masses, GrSel sel,
    const GrBox *__restrict__ boxes, int weighted, GrFrameState *state) {
    __shared__ double lds_tot[GR_CEN_K];
    const uint32_t lane = threadIdx.x, frame = blockIdx.x;
    GrFrameState st = state[frame];
    if (st.status != 0) return;
    const GrBox &box = boxes[first_slot + frame];
    const float *xyz = frames + (size_t)(first_slot + frame) * frame_stride;
    gr_small_center_stage<1>(xyz, masses, sel, box, 0, 0, 0, st, lds_tot, lane, 0);                               // the unweighted estimate (iterators.rs:1405-1407)
    if (__shfl(st.status, 0, 64) == 0) gr_small_center_stage<2>(xyz, masses, sel, box, weighted, 0, 1, st, lds_tot, lane, 0);
    if (lane == 0) state[frame] = st;
}

// ONE frame, the whole centre in one dispatch, the result left in host-mapped memory (center_core, gr_api.hip):
// kind 0 naive, 1 Bai-Breen estimate, 2 estimate + unwrapped mean (get_center / get_com) -- the same stages in the same order
// (blockIdx.x = 1: the SECOND selection of a call that wants the centres of two groups of one frame -- group_distance, analysis.rs:348-360 --
//  with its own record and sequence word behind the first)
__global__ __launch_bounds__(64) void k_center_small(
    const float *__restrict__ frames, size_t frame_stride, uint32_t slot, const float *__restrict__ masses, GrSel sel_a, GrSel sel_b,
    const GrBox *__restrict__ boxes, int kind, int weighted, GrFrameState *state_dev, GrFrameState *state_host, uint32_t *flag_host, uint32_t seq) {
    __shared__ double lds_tot[GR_CEN_K];
    const uint32_t lane = threadIdx.x;
    const GrSel &sel = blockIdx.x ? sel_b : sel_a;
    state_dev += blockIdx.x; state_host += blockIdx.x; flag_host += blockIdx.x;
    const GrBox &box = boxes[slot];
    const float *xyz = frames + (size_t)slot * frame_stride;
    GrFrameState st = {};
    st.err_index = GR_NOIDX;
    if (kind == 0) gr_small_center_stage<0>(xyz, masses, sel, box, weighted, 0, 1, st, lds_tot, lane, 0);          // position first (iterators.rs:946-958)
    else if (kind == 1) gr_small_center_stage<1>(xyz, masses, sel, box, weighted, 1, 1, st, lds_tot, lane, 0);     // mass first (:1324-1339)
    else {
        gr_small_center_stage<1>(xyz, masses, sel, box, 0, 0, 0, st, lds_tot, lane, 0);                           // the unweighted estimate (:1405-1407)
        if (__shfl(st.status, 0, 64) == 0) gr_small_center_stage<2>(xyz, masses, sel, box, weighted, 0, 1, st, lds_tot, lane, 0);
    }
    if (lane == 0) gr_small_publish(st, state_dev, state_host, flag_host, seq);
}

// RMSD without fit of a small selection (calc_rmsd, rmsd.rs:75-129,141-166): k_rmsd_accum<0> + k_rmsd_finalize<0> by one wave
// (one wave per frame of the segment; have_state: the frames' states were initialised with the host-side checks -- a frame that failed
//  them is left alone; state_host / flag_host: the one-frame call's mapped record, NULL for a batch, whose states are fetched as usual)
// MODE 0: the closed-form single pass (k_rmsd_accum<0> + k_rmsd_finalize<0>); MODE 1: the literal sums about the frame's own COM, whose
// shift the centre stages have left in the state (k_rmsd_accum<1> + k_rmsd_finalize<1>: the redo of a frame whose image proof failed,
// rmsd_exact in gr_api.hip) -- have_state is then always set
template <int MODE>
__global__ __launch_bounds__(64) void k_rmsd_small(
    const float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, const float *__restrict__ masses, GrSel sel,
    const GrBox *__restrict__ boxes, GrPlanDev plan, GrFrameState *state_dev, int have_state, GrFrameState *state_host, uint32_t *flag_host, uint32_t seq) {
    __shared__ double lds_tot[GR_ACC_K];
    __shared__ float lds_ext[16];
    const uint32_t lane = threadIdx.x, frame = blockIdx.x, slot = first_slot + frame;
    state_dev += frame;
    GrFrameState st0 = {};
    st0.err_index = GR_NOIDX;
    if (have_state) { st0 = *state_dev; if (st0.status != 0) return; }
    const GrBox &box = boxes[slot];
    const float *xyz = frames + (size_t)slot * frame_stride;
    GrLaneAcc L;
    L.reset();
    GrFrameConst fc;
    gr_pos_load(xyz, sel.contiguous ? sel.start : sel.idx[0], fc.gx, fc.gy, fc.gz);
    fc.sx = fc.sy = fc.sz = 0.f;
    if (MODE == 1) { fc.sx = st0.shift[0]; fc.sy = st0.shift[1]; fc.sz = st0.shift[2]; }
    fc.iax = box.iax; fc.iby = box.iby; fc.icz = box.icz;
    fc.rws2 = box.r_ws * box.r_ws;
    fc.tric = !box.ortho;
    fc.wm = plan.w_is_mass != 0;
    const bool wm = fc.wm;
    if (sel.contiguous) {
        const uint32_t first = sel.start, last = sel.start + sel.n;
        const uint32_t g0 = sel.g0 << 6, g1 = (last + 3u) >> 2;
        const float4 *f4 = reinterpret_cast<const float4 *>(xyz);
        const float4 *p4 = reinterpret_cast<const float4 *>(plan.p);
        const float4 *m4 = reinterpret_cast<const float4 *>(masses);
        const float4 *w4 = reinterpret_cast<const float4 *>(plan.w);
        // (from the selection's first group, not its tile's: a small group sits anywhere in its tile; the rows of two trips are requested
        //  together -- a single wave has nobody to hide its load latency behind)
        for (uint32_t ga = (first >> 2) + lane; ga < g1; ga += 128u) {
            float4 r0[2], r1[2], r2[2], q0[2], q1[2], q2[2], mm[2], ww[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t g = ga + (uint32_t)t * 64u < g1 ? ga + (uint32_t)t * 64u : ga;
                gr_rows_load(f4, g, r0[t], r1[t], r2[t]);
                const size_t pg = (size_t)(g - g0);
                gr_rows_load(p4, pg, q0[t], q1[t], q2[t]);
                mm[t] = m4[g];
                ww[t] = wm ? mm[t] : w4[pg];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t g = ga + (uint32_t)t * 64u;
                if (g >= g1) continue;
                const uint32_t i = g << 2;
                GrA4 q;
                gr_rows_unpack(r0[t], r1[t], r2[t], q.x, q.y, q.z);
                gr_rows_unpack(q0[t], q1[t], q2[t], q.px, q.py, q.pz);
                q.m[0] = mm[t].x; q.m[1] = mm[t].y; q.m[2] = mm[t].z; q.m[3] = mm[t].w;
                q.w[0] = ww[t].x; q.w[1] = ww[t].y; q.w[2] = ww[t].z; q.w[3] = ww[t].w;
#pragma unroll
                for (int k = 0; k < 4; ++k) { q.i[k] = i + k; q.ok[k] = (i + k >= first) && (i + k < last); }
                if (MODE == 0 && i >= first && i + 3 < last) gr_flush4<MODE>(L, q, false, box, fc); else gr_flush4<MODE>(L, q, true, box, fc);
            }
        }
    } else {
        const uint32_t n4 = (sel.n + 3u) >> 2;
        for (uint32_t j4 = lane; j4 < n4; j4 += 64u) {
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
    const double t = gr_wave_sum_scatter32_f64(L.acc, lane);          // lane l: the wave total of value l >> 1
    const uint32_t bad_pos = gr_wave_min_u32(L.bad_pos), bad_mass = gr_wave_min_u32(L.bad_mass);
    float e[32];
#pragma unroll
    for (int a = 0; a < 3; ++a) { e[a] = -L.mn[a]; e[3 + a] = L.mx[a]; e[6 + a] = -L.fmn[a]; e[9 + a] = L.fmx[a]; }
#pragma unroll
    for (int k = 12; k < 32; ++k) e[k] = -3.0e38f;
    const float em = gr_wave_max_scatter16(e, lane);                   // lane l: the wave maximum of value l >> 2
    if ((lane & 1u) == 0u) lds_tot[lane >> 1] = t;
    if ((lane & 3u) == 0u) lds_ext[lane >> 2] = em;
    gr_wave_sync();
    if (lane == 0) {
        GrFrameState st = st0;
        double acc[GR_ACC_K];
#pragma unroll
        for (int k = 0; k < GR_ACC_K; ++k) acc[k] = lds_tot[k];
        const float mn[3] = { -lds_ext[0], -lds_ext[1], -lds_ext[2] }, mx[3] = { lds_ext[3], lds_ext[4], lds_ext[5] };
        const float fmn[3] = { -lds_ext[6], -lds_ext[7], -lds_ext[8] }, fmx[3] = { lds_ext[9], lds_ext[10], lds_ext[11] };
        const double g[3] = { fc.gx, fc.gy, fc.gz };
        gr_finalize_math<MODE>(acc, mn, mx, fmn, fmx, bad_pos, bad_mass, box, plan, g, sel.n, st);
        if (state_host) gr_small_publish(st, state_dev, state_host, flag_host, seq); else *state_dev = st;
    }
}
