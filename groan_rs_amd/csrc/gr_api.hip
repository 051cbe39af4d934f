// gr_api.hip -- C ABI of libgroan_hip.so (include/groan_hip.h): host control flow of the MI355X
// geometry engine.  Mirrors the reference's System-level entry points (src/system/{analysis,rmsd,
// modifying,utility}.rs) including the order of their checks; all per-atom work runs in the HIP
// kernels of gr_kernels.h on the context's own stream.  There is no CPU fallback anywhere in this
// file: without a usable device gr_ctx_create fails with GR_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/groan_hip.h"
#include "gr_container.h"
#include "gr_kernels.h"
#include "gr_hot.h"
#include "gr_resident.h"
#include "gr_small.h"
#include "gr_xtc.h"
#include "gr_shape.h"
#include "gr_xtc_dev.h"
#include "gr_xtc_enc_dev.h"
#include "gr_trr.h"
#include "gr_cellgrid.h"
#include "gr_textio.h"
#include "gr_pool.h"
#include <set>
#include <thread>
#include <atomic>
#include <system_error>
#include <chrono>

#define GR_MAX_BATCH 1024    // frames per batched call segment (workspace is sized for this; 82 MB of partial records)
#define GR_MAX_CHUNKS 256    // workgroups per frame in the reduction kernels

namespace {

struct Group {
    std::vector<grc::Block> blocks;
    uint64_t n = 0;
    bool contiguous = true;
    uint32_t start = 0;
    uint32_t *idx_dev = nullptr;
    uint32_t *mask_dev = nullptr;     // one bit per atom of the system (group-limited trajectory reads: built on first use; masked selections: built with the group)
    bool walk_rw = false;             // ... and translate / wrap walk the span as well (k_translate_wrap)
    bool masked = false;              // non-contiguous but DENSE (>= 1/8 of its span, >= 4096 atoms): the sums kernels walk the span with this mask
    uint32_t span = 0;                // atoms from the first to the last selected one
};

}  // namespace

struct gr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // H2D double buffering: uploads run on their own stream; per-slot "upload done" events gate the compute
    // stream, and a ring of "compute done" events gates the next upload into a slot that kernels still read
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> ev_ready;       // per slot, created on first upload
    std::vector<uint8_t> upload_pending;    // per slot: the compute stream has not yet waited on ev_ready
    hipEvent_t ev_done_ring[64] = {};
    uint64_t done_gen = 0;
    std::vector<uint64_t> slot_gen;         // per slot: generation of the last compute call that touched it
    uint32_t *fuse_cnt = nullptr;     // [2 * GR_MAX_BATCH] arrival counters of the fused finalize / close tails (self-resetting)
    int fuse = 1;                     // GR_TUNE_FUSE 0: separate k_rmsd_finalize_lite launch instead of the sums kernel's closing tail
    uint64_t center_fallbacks = 0;    // frames the one-pass centre handed to the estimate pass (copy selection) or to both passes (gr_center_fallbacks)
    uint32_t com_onepass_min = 4096;  // contiguous groups of at least this many atoms take the one-pass get_com / get_center (0 = never)
    uint64_t n = 0, n_pad = 0;
    uint32_t n_slots = 0;
    size_t frame_stride = 0;          // floats per slot
    float *frames = nullptr;          // [n_slots] pair-tiled slots of n_pad atoms (gr_layout.h)
    float *aos_up = nullptr;          // [n_pad][3] packed records: landing buffer of gr_frame_upload on the copy stream (-> k_tile)
    float *aos_dl = nullptr;          // [n_pad][3] packed records: k_untile's output on the compute stream (-> D2H)
    float *masses = nullptr;          // [n_pad]
    std::vector<float> masses_host;   // copy kept for plan bookkeeping (weights == masses test)
    GrBox *boxes_dev = nullptr;       // [n_slots]
    GrBox *boxes_host = nullptr;      // pinned [n_slots]
    std::vector<int> box_status;      // per slot: GR_OK / GR_E_NO_BOX / GR_E_ZERO_BOX / GR_E_UNSUPPORTED_BOX
    std::vector<uint8_t> box9_set;
    std::vector<float> box9_host;     // [n_slots][9]
    std::map<std::string, Group> groups;
    // workspace
    GrCenPartial *cen_partials = nullptr;
    GrAccPartial *acc_partials = nullptr;
    double *fit_partials = nullptr;   // [frames of a segment][fit workgroups per frame]: sum w |R q - p|^2 of k_fit<true>
    size_t fit_partials_cap = 0;
    int two_pass = 1;                 // GR_TUNE_TWO_PASS 0: RMSD-fit keeps the closed-form single-pass rmsd (k_rmsd_accum<0>)
    int masked_sel = 1;               // GR_TUNE_MASKED_SELECTIONS 0: dense scattered selections keep to their gather lists (groups built afterwards)
    int rmsd_fast = 1;                // GR_TUNE_RMSD_FAST 0: the RMSD without fit always takes the exact-product pass (k_rmsd_accum<0>)
    int rmsd_fast_sigmas = 6;         // GR_TUNE_RMSD_FAST_SIGMAS: multiples of the closing step's rounding estimate a frame's rmsd must stand clear of (0: no guard, calibration only)
    uint32_t rmsd_fast_min = 16384;   // GR_TUNE_RMSD_FAST_MIN: smallest contiguous mass-weighted selection that takes k_sums_pk<false, true>
    uint64_t rmsd_fast_frames = 0, rmsd_exact_redos = 0;   // frames closed by the f32-chain pass / handed back to the exact-product pass
    // resident RMSD fit (gr_resident.h): one launch per segment, the frame waits on chip for its rotation
    int resident = 1;                 // GR_TUNE_RESIDENT 0 never, 1 when the frame fills the chip, 2 whenever it fits (tests)
    int pd_sym = 1;                   // GR_TUNE_PAIRDIST_SYMMETRIC: the pair matrix of a selection with itself computes one triangle and mirrors it
    int res_wg_groups = 0;            // GR_TUNE_RESIDENT_WG_GROUPS 0: automatic, 64 .. 1024 (multiple of 64): 4-atom groups per streaming workgroup
    int res_streams = 0;              // GR_TUNE_RESIDENT_STREAMS 0 automatic, 1..GR_RES_MAX_STREAMS: at most so many frame streams per resident launch
    int res_fill16 = 1;               // GR_TUNE_RESIDENT_FILL: sixteenths of the chip the streaming workgroups must fill for the pass to be chosen (resident = 1); 10 until the late combine of round 5
    uint32_t res_last_streams = 0;    // frame streams of the last resident launch (gr_ctx_stat)
    // the resident pass's metronome (gr_resident.h): GR_TUNE_RESIDENT_METRO_NS 0 = the controller below, 1 = off, else a fixed period.
    // The controller keeps, per launch shape, the shortest period the launches have KEPT (a turn took no longer than the period and
    // next to no slot was reached late); it starts from a free-running launch, probes downwards 1.5 % at a time, steps back to the
    // best kept period when a probe fails, and switches the clock off for a shape whose free-running turn it cannot beat.
    int res_metro_ns = 1;             // (1 = off, the default since the end of round 5: with the workgroups' chores on waves that are ahead the pass is no longer paced by memory,
                                      //  no period beats the free-running turn -- and the controller (0) could settle on a SLOW period when a process's first, cold launch
                                      //  read long: 4.52 us kept as "3 % under the free-running 4.66" while the pass free-runs at 3.95: a 12 % loss, seen once in bench.py)
    // workgroups per CU of the grid-launched read-modify-write streams (k_translate_wrap, k_fit_pk): GR_TUNE_STREAM_WGS_PER_CU.  A copy runs
    // fastest with 20-32 KiB of loads in flight per CU (tools/copy_matrix3 --occ: 3 KiB per wave at 2 workgroups per CU 6.2 TB/s, at 8 -- what the
    // registers allow -- 5.77); the surplus workgroups are kept off the CU by LDS they do not use
    int stream_wgs_cu = 0;            // 0: chosen by the library (see stream_lds), 1..8
    int translate_rows = 1;           // GR_TUNE_TRANSLATE_ROWS 1: translate / wrap / centre of a contiguous selection in an orthorhombic cell as one float4 per lane (k_translate_wrap_rows), 0: the three-rows walk
    int center_resident = 1;          // GR_TUNE_CENTER_RESIDENT 0: atoms_center is always two passes; 1: the resident pass's atoms_center form where it pays (see center_resident)
    uint64_t cen_res_launches = 0, cen_res_redone = 0;   // gr_ctx_stat: resident atoms_center launches; frames they handed back to the two passes
    int res_fit_last = 0;             // GR_TUNE_RESIDENT_FIT_LAST 0: chosen by the launch's fill, 1: the fit first, 2: the sums first
    struct Metro { uint64_t shape = 0; double free_ns = 0, T_ns = 0, best_ns = 0, fail_ns = 0; uint32_t off_for = 0, held = 0; } metro;
    uint64_t res_metro_period_ns = 0, res_last_turn_ns = 0, res_late_permille = 0, res_sclk_mhz = 0;   // gr_ctx_stat: the last resident launch
    int wall_khz = 100000;            // rate of the device's wall_clock64() (hipDeviceAttributeWallClockRate)
    uint32_t res_max_wgs = 0;         // workgroups of k_fit_resident the device holds at once (0: the pass cannot run here)
    int res_test_no_start = 0;        // GR_TUNE_TEST_RESIDENT_NO_START (tests): the next resident launch finds its start verdict already "never started"
    uint32_t res_test_abort_at = 0xFFFFFFFFu;   // GR_TUNE_TEST_RESIDENT_ABORT_AT (tests): the finalizer of this frame of the next resident launch raises `abort`
    uint32_t res_skip = 0, res_backoff = 0;     // segments the pass sits out after a missed start handshake (doubles with every miss in a row)
    uint32_t *res_progress = nullptr;           // [GR_MAX_CHUNKS][8]: frames each streaming wave of the last resident launch had fitted when it left
    unsigned long long *res_wgrec = nullptr; size_t res_wgrec_cap = 0;   // [frames][streaming workgroups, padded to 16][32] tagged words
    unsigned long long *res_rec = nullptr;   // [GR_MAX_BATCH][16]
    uint32_t *res_abort = nullptr;    // device word
    uint32_t *res_words_host = nullptr;   // pinned: the three control words of the last resident launch, copied behind it on the stream
    GrShapeSet *shape_set_dev = nullptr;      // the shapes of the geometry selection in flight (k_shape_mask reads them through a uniform pointer)
    unsigned long long *shape_mask_dev = nullptr, *shape_mask_host = nullptr; size_t shape_mask_cap = 0;   // geometry selection: one bit per atom of the source group (device, pinned host), grown on demand
    uint32_t res_epoch = 0;
    bool res_in_use = false;          // the pending segment took the resident pass (segment_end checks the abort word)
    uint64_t res_launches = 0, res_handshake_misses = 0, res_aborts = 0, res_redone_frames = 0;   // gr_ctx_stat
    GrFrameState *state_dev = nullptr;
    GrFrameState *state_host = nullptr;   // pinned
    // single-wave kernels of per-frame calls on small selections (gr_small.h): the kernel leaves the frame's state and the call's
    // sequence number in coherent host memory, the host polls the word instead of synchronising the stream
    GrFrameState *small_state = nullptr, *small_state_dev = nullptr;   // host-mapped record and its device address
    uint32_t *small_flag = nullptr, *small_flag_dev = nullptr;         // ... and the sequence word
    uint32_t small_seq = 0;
    uint32_t small_max = GR_SMALL_MAX_DEFAULT;                         // GR_TUNE_SMALL_CALLS: largest selection (atoms) that takes them, 0 = never
    uint64_t small_calls = 0, small_sync_fallbacks = 0;                // gr_ctx_stat
    uint32_t *bad_dev = nullptr;          // [4 * GR_MAX_BATCH]: per frame, first atom without position (rows / columns)
    uint32_t *bad_host = nullptr;         // pinned, same size
    float *pd_out = nullptr; size_t pd_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // per-kernel HIP-event profile of the batched RMSD path (gr_profile_*): 0 accumulate, 1 finalize, 2 fit
    hipEvent_t pev[6 * GR_MAX_BATCH] = {};   // per group: begin / end of the sums, finalize and fit kernels
    // launch geometry of the batched RMSD path (gr_ctx_set_tuning)
    uint32_t sub_batch = 256;   // frames per sums->fit group.  The group's frames (12 MB each) leave the L2 / Infinity Cache
                                // between the two passes at any size worth launching, so the size only trades launch
                                // boundaries (6-11 us each) against the sums pass's chunk count per frame.  Measured at 1e6
                                // atoms, 1024 frames per call: 16 -> 116 k frames/s, 32 -> 134 k, 64 -> 145.6 k, 128 -> 148.3 k,
                                // 256 -> 150.5 k, 512 -> 121 k (one launch of 512 x 976 workgroups dispatches slower).
    uint32_t chunks = 0;        // workgroups per frame in the reductions (0 = auto)
    uint32_t fit_wgs = 0;       // workgroups per frame in k_fit (0 = auto)
    int profile = 0;
    double prof_ms[4] = { 0, 0, 0, 0 };
    uint64_t prof_launches[4] = { 0, 0, 0, 0 };
    uint64_t prof_frames[4] = { 0, 0, 0, 0 };
    uint32_t n_cus = 0;
    // device-side xtc unpacking (gr_xtc_read_frames_device): grow-only staging, pinned host mirror + device copy
    // two pinned staging banks used in turn: the host reads + skims batch k + 1 while the H2D copy of batch k drains the other
    // and two device banks: the H2D copy of batch k + 1 (copy stream) runs beside k_xtc_unpack of batch k (unpack stream)
    unsigned char *xtc_host[2] = { nullptr, nullptr }, *xtc_dev[2] = { nullptr, nullptr }; size_t xtc_host_cap[2] = { 0, 0 }, xtc_dev_cap[2] = { 0, 0 };
    hipEvent_t xtc_ev[2] = { nullptr, nullptr };        // the bank's last H2D has left the pinned bank / filled the device bank
    hipEvent_t xtc_unpacked[2] = { nullptr, nullptr };  // the bank's last unpack kernel has finished with the device bank
    hipStream_t unpack_stream = nullptr;
    uint32_t xtc_bank = 0;
    float *wr_host = nullptr; size_t wr_cap = 0;   // pinned landing buffer of gr_xtc_write_slots (grow-only)
    // device xtc encoder (gr_xtc_enc_dev.h), grow-only: quantised atoms, run words, run descriptors, streams
    void *xe_dev[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr }; size_t xe_cap[6] = { 0, 0, 0, 0, 0, 0 };   // ints, enc, runs, meta, hdr + offsets, out
    unsigned char *xe_host[2] = { nullptr, nullptr }; size_t xe_host_cap[2] = { 0, 0 };     // pinned, two banks (rounds in turn): headers + offsets + streams
    int xtc_dev_encode = 1;           // GR_TUNE_XTC_DEVICE_ENCODE
    uint64_t xtc_dev_frames = 0;      // frames gr_xtc_write_slots compressed on the device (GR_STAT_XTC_DEVICE_FRAMES)
    int strict = 0;
    gr_rmsd_plan *in_flight = nullptr;   // the plan whose gr_rmsd_batch_begin has not been ended yet (shared workspace: one at a time)
    uint32_t in_flight_s0 = 0, in_flight_n = 0;   // its slots
    uint64_t epoch = 1;                  // bumped whenever masses or groups change: plans re-resolve what they cached
    std::string err;
    uint64_t err_index = 0;
    uint64_t counts[2] = { 0, 0 };
};

struct Pending {   // a segment between gr_rmsd_batch_begin and gr_rmsd_batch_end
    bool active = false, any_ok = false, consistent = true, fused = false, resident = false, rmsd_fast = false;
    uint32_t res_metro_ns = 0; bool res_metro_auto = false;   // (resident launch: the metronome period it ran with; chosen by the controller?)
    uint32_t s0 = 0, nb = 0, n_prof_groups = 0, res_stream = 0, res_streams = 1;   // (resident launch: streaming workgroups, frame streams)
    int fit = 0;
    std::vector<int> pre;
    std::vector<uint64_t> pre_idx;
    std::vector<std::string> pre_msg;
    GrSel sel = {};        // the group as it was at begin (the context refuses group / mass changes while a batch is in flight)
    uint64_t group_n = 0;
    bool has_group = false;
    bool small = false; uint32_t small_seq = 0;   // the segment is one single-wave dispatch (gr_small.h): segment_end polls the mapped word
#ifdef GR_EXP_TIMELINE
    unsigned long long *tl_dev = nullptr;
#endif
};

struct gr_rmsd_plan {
    gr_ctx *target = nullptr;
    Pending pend;
    std::string group;
    uint64_t n_ref = 0;
    float *p_dev = nullptr, *w_dev = nullptr;
    float *p_span_dev = nullptr;            // masked selections: the reference coordinates once more, laid out by atom over the group's span
    std::vector<grc::Block> ref_blocks;     // ... valid for a target group with exactly these blocks
    bool span_ok = false;                   // (decided with w_is_mass whenever the target's groups / masses change)
    std::vector<float> w_host;   // reference masses of the group, selection order
    GrPlanDev dev = {};
    int exact = 0;
    bool fit_behind_small = false;          // the last segment queued a fit kernel behind a single-wave kernel and returned on that kernel's flag (gr_rmsd_plan_destroy)
    uint32_t last_fallbacks = 0;
    bool resolved = false;
    uint64_t resolved_epoch = 0;   // context epoch at which w_is_mass was decided
};

namespace {

thread_local std::string g_create_err;

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return GR_E_HIP;                                                                   \
        }                                                                                      \
    } while (0)

int fail(gr_ctx *c, int status, const std::string &msg, uint64_t index = 0) {
    c->err = msg; c->err_index = index; return status;
}

GrSel make_sel(const Group &g) {
    GrSel s;
    s.n = (uint32_t)g.n; s.contiguous = g.contiguous ? 1u : 0u; s.start = g.start; s.g0 = g.start >> 8; s.idx = g.idx_dev;
    s.masked = (g.masked && g.mask_dev) ? (g.walk_rw ? 3u : 1u) : 0u; s.span = g.contiguous ? (uint32_t)g.n : g.span; s.mask = g.mask_dev;
    return s;
}

uint32_t chunks_for(const GrSel &s) {
    const uint64_t units = s.contiguous ? ((uint64_t)s.n + 3) / 4 : (uint64_t)s.n;
    uint64_t c = units / ((uint64_t)GR_WG * 4);
    if (c < 1) c = 1;
    if (c > GR_MAX_CHUNKS) c = GR_MAX_CHUNKS;
    return (uint32_t)c;
}

// workgroups per frame of the reduction kernels when `nf` frames share one launch: enough workgroups to
// fill the chip (~3 resident 256-thread workgroups per CU at this register budget), but as few as that
// allows, so each lane streams many atoms per 32-value fp64 wave reduction
uint32_t batch_chunks(const gr_ctx *c, const GrSel &s, uint32_t nf) {
    if (c->chunks) return c->chunks;
    const uint64_t units = ((uint64_t)(s.masked ? s.span : s.n) + 3) / 4;      // (a masked selection is walked over its whole span)
    uint64_t by_work = units / GR_WG;            // at least one trip per lane
    if (by_work < 1) by_work = 1;
    uint64_t want = (1536 + nf - 1) / nf;        // ~2 rounds of 768 resident workgroups
    // (never fewer than 8 when the selection has the work for it: 8 is the smallest count that keeps chunk c on XCD c % 8 for every
    // frame -- 256-frame launches used to get 6 chunks per frame: 2.65 instead of 2.54 us per 1e6-atom frame in the RMSD pass)
    if (want < 8) want = 8;
    uint64_t ch = want < by_work ? want : by_work;
    if (ch > GR_MAX_CHUNKS) ch = GR_MAX_CHUNKS;
    // XCD-aware: workgroups are dealt round-robin over the 8 XCDs, and block (chunk, frame) has linear id
    // frame*ch + chunk; with ch a multiple of 8 chunk c lands on XCD c % 8 for EVERY frame, so each XCD's 4 MiB L2
    // keeps its own eighth of the reference coordinates + masses (16 MB total) resident across all frames of the
    // launch (measured: 13 MB/frame of HBM fetch for 28 MB/frame algorithmic; chunk counts that are not multiples
    // of 8 rotate the mapping and run ~25 % slower)
    if (ch >= 8) ch &= ~(uint64_t)7;
    return (uint32_t)ch;
}

// Workgroups per frame of k_fit: one 256-atom tile per wave (measured at 1e6 atoms x 64 frames: 3.8 us/frame with 976
// workgroups per frame, 4.05 with 256, 4.45 with 64 -- the read-modify-write stream wants every wave slot busy).  A
// multiple of 8 keeps workgroup x on XCD x % 8 for every frame, so each XCD's L2 keeps its own eighth of the reference
// coordinates + weights that k_fit<true> re-reads per frame.
uint32_t fit_grid(const gr_ctx *c, uint32_t nf) {
    (void)nf;
    if (c->fit_wgs) return c->fit_wgs;
    const uint64_t tiles = (c->n + 255) >> 8;
    uint64_t gx = (tiles + GR_WG / 64 - 1) / (GR_WG / 64);
    if (gx >= 8) gx &= ~(uint64_t)7;
    return (uint32_t)(gx < 1 ? 1 : gx);
}

// the sixteen variants of the resident kernel (weights are the masses / every frame has the same box / the selection is the whole
// system / sums before fit): allow their LDS size, hand out the one a launch needs
static const void *resident_fn(bool wmass, bool ubox, bool v, bool fl) {
#define GR_RES_FN(W, U, V, F) reinterpret_cast<const void *>(&k_fit_resident<W, U, V, F>)
    static const void *const fn[16] = {
        GR_RES_FN(false, false, false, false), GR_RES_FN(true, false, false, false), GR_RES_FN(false, true, false, false), GR_RES_FN(true, true, false, false),
        GR_RES_FN(false, false, true, false), GR_RES_FN(true, false, true, false), GR_RES_FN(false, true, true, false), GR_RES_FN(true, true, true, false),
        GR_RES_FN(false, false, false, true), GR_RES_FN(true, false, false, true), GR_RES_FN(false, true, false, true), GR_RES_FN(true, true, false, true),
        GR_RES_FN(false, false, true, true), GR_RES_FN(true, false, true, true), GR_RES_FN(false, true, true, true), GR_RES_FN(true, true, true, true) };
#undef GR_RES_FN
    return fn[(wmass ? 1 : 0) | (ubox ? 2 : 0) | (v ? 4 : 0) | (fl ? 8 : 0)];
}
// ... and the four of its atoms_center form (MODE 1: rows parked, sums first; the centre mass-weighted or not is a launch parameter)
static const void *resident_center_fn(bool ubox, bool fl) {
    static const void *const fn[4] = { reinterpret_cast<const void *>(&k_fit_resident<false, false, false, false, 1>), reinterpret_cast<const void *>(&k_fit_resident<false, true, false, false, 1>),
                                       reinterpret_cast<const void *>(&k_fit_resident<false, false, false, true, 1>), reinterpret_cast<const void *>(&k_fit_resident<false, true, false, true, 1>) };
    return fn[(ubox ? 1 : 0) | (fl ? 2 : 0)];
}
#ifndef GR_STREAM_WGS_CU_DEFAULT
#define GR_STREAM_WGS_CU_DEFAULT 8
#endif
static bool resident_prepare() {
    bool ok = true;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_translate_wrap), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 5120) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fit_pk<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 5120) == hipSuccess;
    for (int v = 0; v < 16; ++v)
        ok = ok && hipFuncSetAttribute(resident_fn((v & 1) != 0, (v & 2) != 0, (v & 4) != 0, (v & 8) != 0), hipFuncAttributeMaxDynamicSharedMemorySize, GrResShape::LDS_BYTES) == hipSuccess;
    for (int v = 0; v < 4; ++v)
        ok = ok && hipFuncSetAttribute(resident_center_fn((v & 1) != 0, (v & 2) != 0), hipFuncAttributeMaxDynamicSharedMemorySize, GrResShape::LDS_BYTES) == hipSuccess;
    return ok;
}

// waves of a finalizer workgroup that close one frame together (gr_resident.h, "teams"): enough for the frame's records (32 per
// wave), and no more frames per workgroup and round than there are streams -- the frames of a round are closed together, so they
// must be frames that become ready together (the same turn of different streams): frames of ONE stream in a round would hold the
// earlier one back until the later one's sums exist, and 7 or more of them would wait for each other for ever (a stream runs 6
// frames ahead)
static uint32_t resident_team_waves(uint32_t wgs, uint32_t streams) {
    uint32_t tw = wgs <= 32 ? 1u : wgs <= 64 ? 2u : wgs <= 128 ? 4u : 8u;
    while (GrResShape::WAVES / tw > streams) tw *= 2u;
    return tw;
}
static uint32_t resident_finalizers_needed(uint32_t wgs, uint32_t streams) {
    if (streams <= 1) return 2;
    const uint32_t teams = GrResShape::WAVES / resident_team_waves(wgs, streams);
    return std::min<uint32_t>(GR_RES_MAX_FIN, (4u * streams + teams - 1u) / teams);
}

// Workgroups per frame of the resident RMSD-fit pass (gr_resident.h) and the number of frame streams the launch runs side by
// side (*streams), or 0 when the two-pass path takes the segment: streams x workgroups must fit the device beside at least two
// finalizer workgroups, and -- unless forced -- fill most of the chip: a launch that leaves CUs idle streams slower than the
// two-pass kernels, which spread every frame over all of them.
uint32_t resident_wgs(gr_ctx *c, bool lite, uint32_t nb, const GrSel &sel, uint32_t *streams, uint32_t *groups_wg) {
    *streams = 1; *groups_wg = GR_RES_GROUPS;
    if (!lite || !c->resident || !c->res_max_wgs) return 0;
    const uint64_t groups = ((c->n + 255) >> 8) << 6;
    // groups per workgroup (GR_TUNE_RESIDENT_WG_GROUPS; 0: 1024, two per lane)
    const uint64_t gwg = c->res_wg_groups ? (uint64_t)c->res_wg_groups : (uint64_t)GR_RES_GROUPS;
    const uint64_t wgs = (groups + gwg - 1) / gwg;
    *groups_wg = (uint32_t)gwg;
    if (wgs + 2 > c->res_max_wgs || wgs > GR_MAX_CHUNKS) return 0;
    // frames that fill less than half of the chip: several of them in flight side by side, each on its own share of the CUs (at
    // least two finalizer workgroups however many streams; a segment gives every stream 16 frames or more, see below).  Forced
    // launches (tests) stay with one stream unless GR_TUNE_RESIDENT_STREAMS asks for more.
    uint64_t s_max = c->res_streams ? (uint64_t)c->res_streams : (c->resident == 1 ? (uint64_t)GR_RES_MAX_STREAMS : 1u);
    s_max = std::min<uint64_t>(s_max, std::min<uint64_t>(c->res_max_wgs - 2, GR_MAX_CHUNKS) / wgs);   // (res_progress holds GR_MAX_CHUNKS workgroups)
    // ... and the streams must leave the finalizers they need: a finalizer workgroup closes `teams` frames in ~15 us while the
    // streams deliver one frame each per ~5 us turn, so 4 S / teams workgroups keep up with a margin (measured without this rule:
    // 23 streams of 45 000 atoms squeezed the finalizers down to 3 and ran 20 % SLOWER than the two passes)
    while (s_max > 1 && s_max * wgs + resident_finalizers_needed((uint32_t)wgs, (uint32_t)s_max) > c->res_max_wgs) s_max--;
    if (c->resident == 1) s_max = std::min<uint64_t>(s_max, nb / 16u);
    else s_max = std::min<uint64_t>(s_max, nb);
    // ONE stream of a frame that fills between a half and two thirds of the chip (510-700 k atoms on MI355X: two of them do not fit):
    // cut into workgroups of 768 groups -- three group-units per SIMD instead of four -- the frame spreads over a third more CUs
    // and a turn gets shorter with it.  Measured (profiles/r04_hole_sweep.txt, frames/s, two passes / 1024 / 768): 520 k atoms
    // 274 k / 281 k / 291 k, 600 k 242 k / 271 k / 278 k, 650 k 229 k / 268 k / 273 k, 690 k 212 k / 265 k / 267 k.
    uint64_t wgs_used = wgs;
    // (the window is a share of the device's workgroups -- two thirds: 170 of MI355X's 256 -- not a constant of one part)
    if (c->resident == 1 && !c->res_wg_groups && s_max <= 1 && wgs * 3u <= (uint64_t)c->res_max_wgs * 2u) {
        const uint64_t w768 = (groups + 767) / 768;
        if (w768 + 2 <= c->res_max_wgs && w768 <= GR_MAX_CHUNKS) { wgs_used = w768; *groups_wg = 768; }
    }
    // the pass costs the same per turn whatever the frame's size (every CU runs its 4096 atoms' worth or idles): it only beats
    // the two passes, whose time shrinks with the frame, when the streams together fill enough of the chip (measured,
    // profiles/r03_size_sweep.txt: + 3 % at 0.67 of the chip, - 8 % at 0.57; the default asked for 10/16).  Round 5: with the workgroup's
    // chores on a wave that is ahead the pass wins at ANY fill -- 500 .. 16 000 atoms per frame, 32 streams on 32-128 CUs: 4.9-6.2 M frames/s
    // against 2.8-4.3 M for the two passes (profiles/r05_small_frames.txt); the default asks for 1/16
    if (c->resident == 1 && (s_max == 0 || s_max * wgs_used * 16 < (uint64_t)c->res_max_wgs * (uint64_t)c->res_fill16)) return 0;
    // ... and when the segment is long enough to pay for filling and draining the six-frame pipeline (measured at 16 frames per
    // call: 9.8 us per frame against 10.6 for the two passes; single frames are a chain of waits)
    if (c->resident == 1 && nb < 16) return 0;
    if (s_max == 0) s_max = 1;
    // ... and when the selection is (nearly) the whole system.  Every workgroup of the pass advances at the pace of the slowest one
    // (a frame's record needs all of them), and with a partial selection the workgroups that hold it run sums AND the literal fit
    // arithmetic while the others only fit: measured at 1e6 atoms with a tenth of them selected, 6.3 us per frame against 4.5 for
    // the two passes, whose sums pass shrinks with the selection (profiles/r03_secondary.json)
    // Round 4, with the pass's waits fixed (profiles/r04_selection.json, 1e6 atoms, us per frame, two passes / resident): 90 % of the atoms
    // selected 6.53 / 5.10, 70 % 6.00 / 5.11, 50 % 5.50 / 5.16, 40 % 5.26 / 5.24, 30 % 5.05 / 5.27, 10 % 4.66 / 5.30 -- the pass is flat
    // (every workgroup streams and fits whatever it holds of the selection), the two passes shrink with the selection: the
    // cross-over lies at ~40 %, the default takes the pass from 45 % (round 3: from 90 %)
    // (a masked selection costs the two passes its SPAN, not its atoms: every third atom of the whole system 6.7 us against the pass's 5.2)
    if (c->resident == 1 && (uint64_t)(sel.masked ? sel.span : sel.n) * 100 < c->n * 45) return 0;
    // a launch that missed its start handshake (the device was busy with somebody else's kernels) makes the context sit out a few
    // segments -- twice as many after every miss in a row -- instead of giving the pass up for good (gr_ctx_stat counts the misses)
    if (c->res_skip) { c->res_skip--; return 0; }
    *streams = (uint32_t)s_max;
    return (uint32_t)wgs_used;
}

#ifdef GR_EXP_STEPTIME
static unsigned long long *g_steptime_dbg = nullptr;
#endif
// dynamic LDS that keeps a streaming kernel at `per_cu` workgroups per CU (160 KiB per CU; the kernels' own static LDS is < 5 KiB)
static uint32_t stream_lds(const gr_ctx *c, int fallback_per_cu) {
    const int per_cu = c->stream_wgs_cu ? c->stream_wgs_cu : fallback_per_cu;
    return per_cu >= 8 ? 0u : (uint32_t)(160 * 1024 / per_cu - 5120);
}
// One resident launch at a time per device and process: its workgroups wait for one another, so two of them sharing the CUs could
// each hold half the chip and starve.  A context that finds the device taken lets the two-pass path handle its segment.
static std::atomic<int> g_resident_in_flight[64];
static bool resident_acquire(int device) { int z = 0; return device >= 0 && device < 64 && g_resident_in_flight[device].compare_exchange_strong(z, 1); }
static void resident_release(int device) { if (device >= 0 && device < 64) g_resident_in_flight[device].store(0); }
// the context's pending segment no longer owns the device's resident slot (normal end, failure, or the plan / context going away
// with a batch in flight): every path that ends a segment comes through here, so the slot cannot leak
static void resident_done(gr_ctx *c) { if (c && c->res_in_use) { c->res_in_use = false; resident_release(c->device); } }

// a batch begun with gr_rmsd_batch_begin is still in flight on this context: only uploads may run beside it
int busy_check(gr_ctx *c) {
    if (c && c->in_flight) return fail(c, GR_E_INVALID_ARG, "a batch begun with gr_rmsd_batch_begin is in flight on this context: only gr_frame_upload / gr_frame_upload_wait / gr_host_* / gr_*_read_frames_device may run before gr_rmsd_batch_end");
    return GR_OK;
}

// device memory of a group that is being replaced / removed (kernels that read it may still be running)
void group_release(gr_ctx *c, Group &g) {
    if (!g.idx_dev && !g.mask_dev) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->unpack_stream) (void)hipStreamSynchronize(c->unpack_stream);
    if (g.idx_dev) (void)hipFree(g.idx_dev);
    if (g.mask_dev) (void)hipFree(g.mask_dev);
    g.idx_dev = nullptr; g.mask_dev = nullptr;
}

// expand a block list into the device-side selection (nothing for a contiguous list)
int group_build(gr_ctx *c, std::vector<grc::Block> blocks, Group *out) {
    uint64_t bad = 0;
    // no kernel bounds-checks a selection: a block outside [0, n) never becomes a group (the reference panics on first use)
    if (!grc::valid_for(blocks, c->n, &bad)) return fail(c, GR_E_OUT_OF_RANGE, "atom index out of range", bad);
    Group g;
    g.blocks = std::move(blocks);
    g.n = grc::n_atoms(g.blocks);
    g.contiguous = g.blocks.size() <= 1;
    g.start = g.blocks.empty() ? 0u : (uint32_t)g.blocks[0].first;
    if (!g.contiguous) {
        std::vector<uint64_t> e = grc::expand(g.blocks);
        std::vector<uint32_t> e32(e.begin(), e.end());
        HIPCHK(c, hipMalloc(&g.idx_dev, e32.size() * sizeof(uint32_t)));
        if (hipMemcpy(g.idx_dev, e32.data(), e32.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(g.idx_dev);
            return fail(c, GR_E_HIP, "copy of the selection's index list failed");
        }
        // a DENSE scattered selection also gets its bit mask: the streaming sums kernels then read its span coalesced instead of
        // gathering it atom by atom (tools/gather_bench.py: two blocks of 166 667 atoms cost 3x one block of 333 334 on the gather list)
        g.span = (uint32_t)(g.blocks.back().second - g.blocks.front().first + 1);
        g.masked = c->masked_sel && g.n >= 4096 && g.n * 8 >= (uint64_t)g.span;
        if (g.masked) {
            std::vector<uint32_t> bits(((size_t)c->n_pad + 31) / 32, 0u);
            for (const auto &b : g.blocks) for (uint64_t a = b.first; a <= b.second; ++a) bits[a >> 5] |= 1u << (a & 31u);
            // translate / wrap (read-modify-write of whole 4-atom groups) walk the span too when that moves fewer bytes than the list: a
            // group without a selected atom is skipped unread, so what counts is the number of groups the selection TOUCHES -- ~17 ps
            // per touched group against 8 (runs of neighbours) to 14 ps (lone atoms) per atom on the list (tools/wrap_bench.py)
            uint64_t touched = 0;
            for (uint32_t w : bits) for (int q = 0; q < 8; ++q) touched += ((w >> (4 * q)) & 15u) != 0u;
            g.walk_rw = touched * 17 <= g.n * 14;
            if (hipMalloc(&g.mask_dev, bits.size() * sizeof(uint32_t)) != hipSuccess ||
                hipMemcpy(g.mask_dev, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
                if (g.mask_dev) (void)hipFree(g.mask_dev);
                g.mask_dev = nullptr; g.masked = false;          // (the gather list serves every path)
            }
        }
    }
    *out = g;
    return GR_OK;
}

int install_group(gr_ctx *c, const char *name, std::vector<grc::Block> blocks) {
    if (!name) return fail(c, GR_E_INVALID_ARG, "group name is NULL");
    int st = busy_check(c); if (st) return st;
    Group g;
    st = group_build(c, std::move(blocks), &g); if (st) return st;   // on failure the context is unchanged (no group created, none lost)
    const bool existed = c->groups.count(name) != 0;
    if (existed) group_release(c, c->groups[name]);
    c->groups[name] = g;
    c->epoch++;
    return existed ? GR_E_GROUP_EXISTS : GR_OK;
}

const Group *find_group(const gr_ctx *c, const char *name) {
    if (!name) return nullptr;
    auto it = c->groups.find(name);
    return it == c->groups.end() ? nullptr : &it->second;
}

// simbox_check (simbox.rs:230-236) for one slot
int box_check(gr_ctx *c, uint32_t slot) {
    const int st = c->box_status[slot];
    if (st == GR_E_NO_BOX) return fail(c, GR_E_NO_BOX, "simulation box does not exist");
    if (c->strict && !c->boxes_host[slot].ortho) return fail(c, GR_E_NOT_ORTHOGONAL, "simulation box is not orthogonal");
    if (st != GR_OK) return fail(c, st, st == GR_E_ZERO_BOX ? "box length is not positive" : "box too skewed");
    return GR_OK;
}

// ingest = the call only feeds slots (uploads on the copy stream): allowed beside a batch in flight -- into OTHER slots (the
// batch's own frames may still be needed by gr_rmsd_batch_end, which redoes frames whose image proof failed); everything
// else is refused while a batch is in flight
int slot_check(gr_ctx *c, uint32_t slot, uint32_t n = 1, bool ingest = false) {
    if (!c) return GR_E_INVALID_ARG;
    if ((uint64_t)slot + n > c->n_slots || n == 0) return fail(c, GR_E_INVALID_ARG, "slot out of range");
    if (!ingest) return busy_check(c);
    if (c->in_flight && slot < c->in_flight_s0 + c->in_flight_n && c->in_flight_s0 < slot + n)
        return fail(c, GR_E_INVALID_ARG, "upload into a slot of the batch in flight: call gr_rmsd_batch_end first (or use the other half of the double buffer)");
    return GR_OK;
}

// Every API call that reads or writes frame slots on the compute stream brackets itself with a SlotUse:
// entry  -> the compute stream waits for uploads still in flight into those slots
// exit   -> a "compute done" event is recorded and remembered per slot, so a later upload into the same slot
//           (the other half of a double buffer) starts only after the kernels that read it have finished.
struct SlotUse {
    gr_ctx *c; uint32_t first, n;
    bool quiet = false;   // the caller has already WAITED for the kernels that read the slots (gr_small.h: the result came back): no event needed
    SlotUse(gr_ctx *ctx, uint32_t first_slot, uint32_t count = 1) : c(ctx), first(first_slot), n(count) {
        for (uint32_t s = first; s < first + n && s < c->n_slots; ++s)
            if (c->upload_pending[s]) { (void)hipStreamWaitEvent(c->stream, c->ev_ready[s], 0); c->upload_pending[s] = 0; }
    }
    ~SlotUse() {
        if (quiet) return;
        const uint64_t gen = ++c->done_gen;
        (void)hipEventRecord(c->ev_done_ring[gen % 64], c->stream);
        for (uint32_t s = first; s < first + n && s < c->n_slots; ++s) c->slot_gen[s] = gen;
    }
};

// both ingest streams idle: the H2D copy stream and the xtc unpack stream
static hipError_t sync_ingest(gr_ctx *c) {
    hipError_t e = hipStreamSynchronize(c->copy_stream);
    if (e == hipSuccess && c->unpack_stream) e = hipStreamSynchronize(c->unpack_stream);
    return e;
}
// host half of set_box: boxes_host[slot] + the slot's box status; the caller copies boxes_host to boxes_dev
void box_fill(gr_ctx *c, uint32_t slot, const float *box9) {
    GrBox &b = c->boxes_host[slot];
    if (!box9) {
        gr_box_setup(nullptr, &b);
        c->box_status[slot] = GR_E_NO_BOX; c->box9_set[slot] = 0;
    } else {
        // count feasible candidates first: more than the table holds = unsupported skew
        const int ok = gr_box_setup(box9, &b);
        memcpy(&c->box9_host[9 * (size_t)slot], box9, 9 * sizeof(float));
        c->box9_set[slot] = 1;
        if (!ok) c->box_status[slot] = GR_E_ZERO_BOX;
        else if (b.ncand > GR_MAX_CAND) c->box_status[slot] = GR_E_UNSUPPORTED_BOX;
        else c->box_status[slot] = GR_OK;
    }
}
int set_box(gr_ctx *c, uint32_t slot, const float *box9, hipStream_t on = nullptr) {
    box_fill(c, slot, box9);
    HIPCHK(c, hipMemcpyAsync(c->boxes_dev + slot, &c->boxes_host[slot], sizeof(GrBox), hipMemcpyHostToDevice, on ? on : c->stream));
    return GR_OK;
}

// the same box for `n` consecutive slots: one gr_box_setup, n host copies, ONE contiguous H2D copy
int set_boxes_same(gr_ctx *c, uint32_t first_slot, uint32_t n, const float *box9, hipStream_t on = nullptr) {
    if (n == 0) return GR_OK;
    float keep[9];
    if (box9) memcpy(keep, box9, sizeof keep);                 // (box9 may point into box9_host, which the loop below rewrites)
    box_fill(c, first_slot, box9 ? keep : nullptr);
    for (uint32_t f = 1; f < n; ++f) {
        const uint32_t s = first_slot + f;
        c->boxes_host[s] = c->boxes_host[first_slot];
        c->box_status[s] = c->box_status[first_slot]; c->box9_set[s] = c->box9_set[first_slot];
        if (box9) memcpy(&c->box9_host[9 * (size_t)s], keep, sizeof keep);
    }
    HIPCHK(c, hipMemcpyAsync(c->boxes_dev + first_slot, c->boxes_host + first_slot, (size_t)n * sizeof(GrBox), hipMemcpyHostToDevice, on ? on : c->stream));
    return GR_OK;
}

// packed-record staging buffers (the C ABI speaks rvec[n], the slots are pair-tiled): allocated on first use, pad atoms zero
int ensure_staging(gr_ctx *c, float **buf) {
    if (*buf) return GR_OK;
    HIPCHK(c, hipMalloc(buf, c->frame_stride * sizeof(float)));
    // (hipMemset on device memory may return before it has run, and the context's streams are non-blocking: clear on the
    // context's stream and wait, so the clear can never land on top of the first frame that passes through the buffer)
    HIPCHK(c, hipMemsetAsync(*buf, 0, c->frame_stride * sizeof(float), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
}
// slot -> packed records in aos_dl, on the compute stream (the caller copies aos_dl out on the same stream)
int untile_slot(gr_ctx *c, uint32_t slot) {
    int st = ensure_staging(c, &c->aos_dl); if (st) return st;
    const uint32_t ng = (uint32_t)(c->n_pad >> 2);
    k_untile<<<dim3((ng + 255) / 256), dim3(256), 0, c->stream>>>(c->frames + (size_t)slot * c->frame_stride, c->aos_dl, ng);
    HIPCHK(c, hipGetLastError());
    return GR_OK;
}

// events of the batched paths are created when first needed
int ensure_event(gr_ctx *c, hipEvent_t *ev, bool timing) {
    if (*ev) return GR_OK;
    HIPCHK(c, timing ? hipEventCreate(ev) : hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return GR_OK;
}
#define EVREC(ctx, ev, timing, stream) do { int s_ = ensure_event((ctx), &(ev), (timing)); if (s_) return s_; HIPCHK((ctx), hipEventRecord((ev), (stream))); } while (0)

int state_reset(gr_ctx *c, uint32_t n) {
    k_state_reset<<<dim3((n + 63) / 64), dim3(64), 0, c->stream>>>(c->state_dev, n);
    HIPCHK(c, hipGetLastError());
    return GR_OK;
}

int fetch_states(gr_ctx *c, uint32_t n) {
    HIPCHK(c, hipMemcpyAsync(c->state_host, c->state_dev, n * sizeof(GrFrameState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
}

int frame_status(gr_ctx *c, const GrFrameState &st) {
    if (st.status == GR_OK) return GR_OK;
    if (st.status == GR_E_NO_POSITION) return fail(c, GR_E_NO_POSITION, "atom has no position", st.err_index);
    if (st.status == GR_E_NO_MASS) return fail(c, GR_E_NO_MASS, "atom has no mass", st.err_index);
    return fail(c, st.status, "frame analysis failed");
}

// Per-frame calls on small selections (gr_small.h).  small_wait: the kernel's last act is the store of `seq` into the mapped word; spin
// on it for up to ~20 ms of wall clock (a healthy call answers in microseconds), then let the stream synchronisation decide.
static bool small_ok(const gr_ctx *c, const GrSel &sel) { return c->small_max != 0 && c->small_state != nullptr && sel.n <= c->small_max && !c->profile; }
static int small_wait(gr_ctx *c, uint32_t seq, uint32_t words = 1) {
    volatile uint32_t *w = c->small_flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; w[0] != seq || w[words - 1] != seq; ++spins) {
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
            c->small_sync_fallbacks++;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (w[0] != seq || w[words - 1] != seq) return fail(c, GR_E_HIP, "a single-wave kernel ended without publishing its result");
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    c->small_calls++;
    return GR_OK;
}

// launch one centre stage (sums + finalize) for `nf` frames starting at first_slot
int center_stage(gr_ctx *c, uint32_t first_slot, uint32_t nf, const GrSel &sel, int kind, int weighted,
                 int mass_first, int target, int only_status = 0) {
    // batches: a multiple of 8 chunks per frame, so that chunk c of every frame runs on XCD c % 8 and that XCD's L2 keeps its
    // eighth of the masses.  (The RMSD kernels' few, long chunks -- batch_chunks -- are slower here: 3.9 vs 3.1 us per 1e6-atom
    // frame for the naive centre; this kernel's per-atom fp64 chains want many workgroups.)
    if (small_ok(c, sel)) {   // a small selection: one wave per frame, sums and closing step in one launch (gr_small.h)
        k_center_small_stage<<<dim3(nf), dim3(64), 0, c->stream>>>(c->frames, c->frame_stride, first_slot, c->masses, sel, c->boxes_dev, kind, weighted, mass_first, target, c->state_dev, only_status);
        HIPCHK(c, hipGetLastError());
        return GR_OK;
    }
    uint32_t nch = chunks_for(sel);
    if (nf > 1 && nch >= 8) nch &= ~7u;
    const dim3 grid(nch, nf), wg(GR_WG);
    if (kind == 0) k_center_sums<0><<<grid, wg, 0, c->stream>>>(c->frames, c->frame_stride, first_slot, c->masses, sel, c->boxes_dev, c->state_dev, weighted, c->cen_partials, only_status);
    else if (kind == 1) k_center_sums<1><<<grid, wg, 0, c->stream>>>(c->frames, c->frame_stride, first_slot, c->masses, sel, c->boxes_dev, c->state_dev, weighted, c->cen_partials, only_status);
    else k_center_sums<2><<<grid, wg, 0, c->stream>>>(c->frames, c->frame_stride, first_slot, c->masses, sel, c->boxes_dev, c->state_dev, weighted, c->cen_partials, only_status);
    k_center_finalize<<<dim3(nf), dim3(GR_WG), 0, c->stream>>>(c->cen_partials, nch, c->boxes_dev, first_slot, kind, weighted, mass_first, target, sel.n, c->state_dev, only_status);
    HIPCHK(c, hipGetLastError());
    return GR_OK;
}

// get_center / get_com: unweighted Bai-Breen estimate, then the (weighted) unwrapped mean
int pbc_center_stages(gr_ctx *c, uint32_t first_slot, uint32_t nf, const GrSel &sel, int weighted, int only_status = 0) {
    if (small_ok(c, sel) && only_status == 0) {   // a small selection: both stages of every frame in one launch, one wave per frame (gr_small.h)
        k_center_small_pbc<<<dim3(nf), dim3(64), 0, c->stream>>>(c->frames, c->frame_stride, first_slot, c->masses, sel, c->boxes_dev, weighted, c->state_dev);
        HIPCHK(c, hipGetLastError());
        return GR_OK;
    }
    int st = center_stage(c, first_slot, nf, sel, 1, 0, 0, 0, only_status);
    if (st != GR_OK) return st;
    return center_stage(c, first_slot, nf, sel, 2, weighted, 0, 1, only_status);
}

// get_center / get_com of a large contiguous group in ONE pass over the frame (the sums pass of the RMSD path without a
// reference: images about the group's first atom, image proof, centre placed in the reference's periodic copy) instead of
// estimate + unwrapped mean.  Frames whose proof fails come back GR_ST_FALLBACK in state_dev; center_redo_fallbacks reruns
// those on the two-pass path.  The states must have been initialised (status 0 or the frame's host-side error).
static bool center_onepass_ok(const gr_ctx *c, const GrSel &sel) {
    return c->com_onepass_min != 0 && (sel.contiguous || sel.masked) && sel.n >= c->com_onepass_min && c->two_pass;
}
static int pbc_center_onepass(gr_ctx *c, uint32_t s0, uint32_t nb, const GrSel &sel, int weighted) {
    GrPlanDev plan; memset(&plan, 0, sizeof plan);
    plan.w_is_mass = weighted ? 1u : 0u;   // NOREF: "weighted"
    plan.n = sel.n; plan.sw = 1.0;
    const uint32_t nch = batch_chunks(c, sel, nb);
    if (sel.masked) {   // a dense scattered group: its span, coalesced, with its bit mask
        k_sums_pk<true, false, true><<<dim3(nch, nb), dim3(GR_WG), 0, c->stream>>>(c->frames, c->frame_stride, s0, c->masses, sel, c->boxes_dev, plan, c->acc_partials,
                                                                                   c->fuse ? c->fuse_cnt : nullptr, c->fuse ? c->state_dev : nullptr);
        if (!c->fuse) k_rmsd_finalize_lite<true><<<dim3(nb), dim3(64), 0, c->stream>>>(c->acc_partials, nch, c->frames, c->frame_stride, s0, sel, c->boxes_dev, plan, c->state_dev);
    } else if (c->fuse) {
        k_sums_pk<true><<<dim3(nch, nb), dim3(GR_WG), 0, c->stream>>>(c->frames, c->frame_stride, s0, c->masses, sel, c->boxes_dev, plan, c->acc_partials, c->fuse_cnt, c->state_dev);
    } else {
        k_sums_pk<true><<<dim3(nch, nb), dim3(GR_WG), 0, c->stream>>>(c->frames, c->frame_stride, s0, c->masses, sel, c->boxes_dev, plan, c->acc_partials, nullptr, nullptr);
        k_rmsd_finalize_lite<true><<<dim3(nb), dim3(64), 0, c->stream>>>(c->acc_partials, nch, c->frames, c->frame_stride, s0, sel, c->boxes_dev, plan, c->state_dev);
    }
    HIPCHK(c, hipGetLastError());
    return GR_OK;
}
// state_host[0..nb) has been fetched: when some frames are flagged GR_ST_FALLBACK the two dependent passes run over the
// whole batch, masked to those frames (the other frames' workgroups leave at once), and the states are fetched again
static int center_redo_fallbacks(gr_ctx *c, uint32_t s0, uint32_t nb, const GrSel &sel, int weighted, std::vector<GrFrameState> &res) {
    uint32_t n_fb = 0, n_amb = 0;
    for (uint32_t f = 0; f < nb; ++f) { n_fb += c->state_host[f].status == GR_ST_FALLBACK; n_amb += c->state_host[f].status == GR_ST_AMBIG; }
    if (n_fb || n_amb) {
        c->center_fallbacks += n_fb + n_amb;
        int st = GR_OK;
        // images proven, centre near a cell face: only the (unweighted) estimate is needed, it selects the periodic copy
        if (n_amb) { st = center_stage(c, s0, nb, sel, 1, 0, 0, 0, GR_ST_AMBIG); if (st) return st; }
        if (n_fb) { st = pbc_center_stages(c, s0, nb, sel, weighted, GR_ST_FALLBACK); if (st) return st; }
        st = fetch_states(c, nb); if (st) return st;
    }
    res.assign(c->state_host, c->state_host + nb);
    return GR_OK;
}

}  // namespace

// Nothing may unwind through the C ABI: every multi-line entry point below is a function-try-block.  What can throw in here
// is memory exhaustion in the host containers (std::bad_alloc) and std::system_error from the worker threads.
static int gr_abi_guard() { return GR_E_HIP; }

extern "C" {

const char *gr_version(void) { return "groan_hip 0.1.0 (gfx950)"; }

const char *gr_status_string(int s) try {
    switch (s) {
    case GR_OK: return "ok";
    case GR_E_NO_BOX: return "simulation box does not exist";
    case GR_E_NOT_ORTHOGONAL: return "simulation box is not orthogonal";
    case GR_E_ZERO_BOX: return "box length is zero or negative";
    case GR_E_EMPTY_GROUP: return "group is empty";
    case GR_E_INCONSISTENT_GROUP: return "group has a different number of atoms in reference and target";
    case GR_E_NO_POSITION: return "atom has undefined position";
    case GR_E_NO_MASS: return "atom has undefined mass";
    case GR_E_GROUP_NOT_FOUND: return "group does not exist";
    case GR_E_OUT_OF_RANGE: return "atom index out of range";
    case GR_E_INVALID_ARG: return "invalid argument";
    case GR_E_GROUP_EXISTS: return "group already existed and was overwritten";
    case GR_E_HIP: return "HIP runtime error";
    case GR_E_NO_DEVICE: return "no usable HIP device";
    case GR_E_UNSUPPORTED_BOX: return "box too skewed for the minimum-image table";
    case GR_E_IO: return "file could not be opened or read";
    case GR_E_FORMAT: return "not a valid xtc file";
    case GR_E_INVALID_NAME: return "invalid group name";
    default: return "unknown status";
    }
} catch (...) { return nullptr; }

int gr_device_count(int *count) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { if (count) *count = 0; return GR_E_NO_DEVICE; }
    if (count) *count = n;
    return n > 0 ? GR_OK : GR_E_NO_DEVICE;
} catch (...) { return gr_abi_guard(); }

gr_ctx *gr_ctx_create(int device, uint64_t n_atoms, uint32_t n_slots, int *status) try {
    int dummy; if (!status) status = &dummy;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) { *status = GR_E_NO_DEVICE; return nullptr; }
    if (n_atoms == 0 || n_atoms > 0xFFFFFFF0ull || n_slots == 0) { *status = GR_E_INVALID_ARG; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { *status = GR_E_NO_DEVICE; return nullptr; }
    gr_ctx *c = new gr_ctx();
    c->device = device; c->n = n_atoms; c->n_pad = (n_atoms + 255) & ~255ull; c->n_slots = n_slots;   // whole 256-atom wave tiles
    c->frame_stride = (size_t)c->n_pad * 3;
    bool ok = true;
    ok = ok && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 64; ++k) ok = ok && hipEventCreateWithFlags(&c->ev_done_ring[k], hipEventDisableTiming) == hipSuccess;
    c->ev_ready.assign(n_slots, nullptr); c->upload_pending.assign(n_slots, 0); c->slot_gen.assign(n_slots, 0);
    // pev (profiling events) are created on first use (ensure_event): thousands of events per context otherwise
    ok = ok && hipMalloc(&c->fuse_cnt, 2 * GR_MAX_BATCH * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemset(c->fuse_cnt, 0, 2 * GR_MAX_BATCH * sizeof(uint32_t)) == hipSuccess;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cus = (uint32_t)prop.multiProcessorCount; }
    ok = ok && hipMalloc(&c->frames, (size_t)n_slots * c->frame_stride * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(&c->masses, c->n_pad * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(&c->boxes_dev, n_slots * sizeof(GrBox)) == hipSuccess;
    ok = ok && hipHostMalloc(&c->boxes_host, n_slots * sizeof(GrBox), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc(&c->cen_partials, (size_t)GR_MAX_BATCH * GR_MAX_CHUNKS * sizeof(GrCenPartial)) == hipSuccess;
    ok = ok && hipMalloc(&c->acc_partials, (size_t)GR_MAX_BATCH * GR_MAX_CHUNKS * sizeof(GrAccPartial)) == hipSuccess;
    ok = ok && hipMalloc(&c->state_dev, GR_MAX_BATCH * sizeof(GrFrameState)) == hipSuccess;
    ok = ok && hipHostMalloc(&c->state_host, GR_MAX_BATCH * sizeof(GrFrameState), hipHostMallocDefault) == hipSuccess;
    {   // (when the mapped record cannot be had the single-wave kernels are simply not used)
        void *hp = nullptr, *dp = nullptr;
        if (ok && hipHostMalloc(&hp, 256, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) {
                memset(hp, 0, 256);
                c->small_state = reinterpret_cast<GrFrameState *>(hp); c->small_state_dev = reinterpret_cast<GrFrameState *>(dp);
                c->small_flag = reinterpret_cast<uint32_t *>(static_cast<char *>(hp) + 192); c->small_flag_dev = reinterpret_cast<uint32_t *>(static_cast<char *>(dp) + 192);
            } else { (void)hipHostFree(hp); (void)hipGetLastError(); }
        } else if (ok) (void)hipGetLastError();
        static_assert(2 * sizeof(GrFrameState) <= 192, "the mapped block holds two frame states + their sequence words");
    }
    ok = ok && hipMalloc(&c->bad_dev, 4 * GR_MAX_BATCH * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipHostMalloc(&c->bad_host, 4 * GR_MAX_BATCH * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess;
    ok = ok && hipMalloc(&c->res_abort, 96 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipHostMalloc(&c->res_words_host, 16 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMemset(c->res_abort, 0, 96 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc(&c->res_rec, (size_t)GR_MAX_BATCH * 16 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMemset(c->res_rec, 0, (size_t)GR_MAX_BATCH * 16 * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMalloc(&c->res_progress, (size_t)GR_MAX_CHUNKS * 8 * sizeof(uint32_t)) == hipSuccess;
    if (ok) {   // the clock the resident pass bounds its waits with
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) c->wall_khz = khz;
        (void)hipGetLastError();
    }
    if (ok) {   // can the resident pass run here?  (160 KiB of LDS per workgroup, one workgroup per CU)
        int per_cu = 0;
        int per_cu_x = 0;
        if (resident_prepare() &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fit_resident<false, false, true, true>, GrResShape::LANES, GrResShape::LDS_BYTES) == hipSuccess && per_cu >= 1 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_x, k_fit_resident<false, false, false, true>, GrResShape::LANES, GrResShape::LDS_BYTES) == hipSuccess && per_cu_x >= 1)
            c->res_max_wgs = c->n_cus * (uint32_t)std::min(per_cu, per_cu_x);
        (void)hipGetLastError();
    }

    if (!ok) { *status = GR_E_HIP; gr_ctx_destroy(c); return nullptr; }
    // masses undefined (None) until gr_set_masses; padding and frames zero
    std::vector<float> nanv(c->n_pad, NAN);
    c->masses_host.assign(c->n, NAN);
    (void)hipMemcpy(c->masses, nanv.data(), c->n_pad * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemset(c->frames, 0, (size_t)n_slots * c->frame_stride * sizeof(float));
    c->box_status.assign(n_slots, GR_E_NO_BOX);
    c->box9_set.assign(n_slots, 0);
    c->box9_host.assign((size_t)n_slots * 9, 0.0f);
    for (uint32_t s = 0; s < n_slots; ++s) gr_box_setup(nullptr, &c->boxes_host[s]);
    (void)hipMemcpy(c->boxes_dev, c->boxes_host, n_slots * sizeof(GrBox), hipMemcpyHostToDevice);
    // System::new creates the group "all" (src/system/mod.rs)
    std::vector<grc::Block> all(1, grc::Block(0, n_atoms - 1));
    install_group(c, "all", all);
    (void)hipDeviceSynchronize();   // the clears above run on the null stream; the context's own streams do not wait for it
    *status = GR_OK;
    return c;
} catch (...) { return nullptr; }

void gr_ctx_destroy(gr_ctx *c) try {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    resident_done(c);
    for (auto &kv : c->groups) { if (kv.second.idx_dev) (void)hipFree(kv.second.idx_dev); if (kv.second.mask_dev) (void)hipFree(kv.second.mask_dev); }
    if (c->frames) (void)hipFree(c->frames);
    if (c->aos_up) (void)hipFree(c->aos_up);
    if (c->aos_dl) (void)hipFree(c->aos_dl);
    if (c->masses) (void)hipFree(c->masses);
    if (c->boxes_dev) (void)hipFree(c->boxes_dev);
    if (c->boxes_host) (void)hipHostFree(c->boxes_host);
    if (c->cen_partials) (void)hipFree(c->cen_partials);
    if (c->acc_partials) (void)hipFree(c->acc_partials);
    if (c->fit_partials) (void)hipFree(c->fit_partials);
    if (c->fuse_cnt) (void)hipFree(c->fuse_cnt);
    if (c->res_abort) (void)hipFree(c->res_abort);
    if (c->res_words_host) (void)hipHostFree(c->res_words_host);
    if (c->shape_set_dev) (void)hipFree(c->shape_set_dev);
    if (c->shape_mask_dev) (void)hipFree(c->shape_mask_dev);
    if (c->shape_mask_host) (void)hipHostFree(c->shape_mask_host);
    if (c->res_wgrec) (void)hipFree(c->res_wgrec);
    if (c->res_rec) (void)hipFree(c->res_rec);
    if (c->res_progress) (void)hipFree(c->res_progress);
    if (c->state_dev) (void)hipFree(c->state_dev);
    if (c->state_host) (void)hipHostFree(c->state_host);
    if (c->small_state) (void)hipHostFree(c->small_state);
    if (c->bad_dev) (void)hipFree(c->bad_dev);
    if (c->bad_host) (void)hipHostFree(c->bad_host);
    if (c->pd_out) (void)hipFree(c->pd_out);
    for (int k = 0; k < 6; ++k) if (c->xe_dev[k]) (void)hipFree(c->xe_dev[k]);
    for (int k = 0; k < 2; ++k) if (c->xe_host[k]) (void)hipHostFree(c->xe_host[k]);
    if (c->unpack_stream) { (void)hipStreamSynchronize(c->unpack_stream); (void)hipStreamDestroy(c->unpack_stream); }
    for (int k = 0; k < 2; ++k) {
        if (c->xtc_host[k]) (void)hipHostFree(c->xtc_host[k]);
        if (c->xtc_dev[k]) (void)hipFree(c->xtc_dev[k]);
        if (c->xtc_ev[k]) (void)hipEventDestroy(c->xtc_ev[k]);
        if (c->xtc_unpacked[k]) (void)hipEventDestroy(c->xtc_unpacked[k]);
    }
    if (c->wr_host) (void)hipHostFree(c->wr_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (int k = 0; k < 6 * GR_MAX_BATCH; ++k) if (c->pev[k]) (void)hipEventDestroy(c->pev[k]);
    for (hipEvent_t e : c->ev_ready) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 64; ++k) if (c->ev_done_ring[k]) (void)hipEventDestroy(c->ev_done_ring[k]);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
} catch (...) { }

const char *gr_last_error(const gr_ctx *c) { return c ? c->err.c_str() : "null context"; }
uint64_t gr_last_error_index(const gr_ctx *c) { return c ? c->err_index : 0; }
void gr_last_error_counts(const gr_ctx *c, uint64_t counts[2]) { if (c && counts) { counts[0] = c->counts[0]; counts[1] = c->counts[1]; } }
int gr_ctx_set_strict_orthogonal(gr_ctx *c, int on) { if (!c) return GR_E_INVALID_ARG; c->strict = on ? 1 : 0; return GR_OK; }
uint64_t gr_n_atoms(const gr_ctx *c) { return c ? c->n : 0; }
uint32_t gr_n_slots(const gr_ctx *c) { return c ? c->n_slots : 0; }

int gr_sync(gr_ctx *c) try {
    if (!c) return GR_E_INVALID_ARG;
    HIPCHK(c, sync_ingest(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_set_masses(gr_ctx *c, const float *masses, uint64_t n) try {
    if (!c || !masses || n != c->n) return c ? fail(c, GR_E_INVALID_ARG, "masses: size mismatch") : GR_E_INVALID_ARG;
    { int st = busy_check(c); if (st) return st; }
    c->masses_host.assign(masses, masses + n);
    c->epoch++;
    HIPCHK(c, hipMemcpyAsync(c->masses, masses, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ containers */
size_t gr_container_from_indices(const uint64_t *indices, size_t n, uint64_t n_atoms, uint64_t *os, uint64_t *oe) try {
    return grc::store(grc::from_indices(std::vector<uint64_t>(indices, indices + n), n_atoms), os, oe);
} catch (...) { return 0; }
size_t gr_container_from_ranges(const uint64_t *s, const uint64_t *e, size_t n, uint64_t n_atoms, uint64_t *os, uint64_t *oe) try {
    return grc::store(grc::from_ranges(s, e, n, n_atoms), os, oe);
} catch (...) { return 0; }
size_t gr_container_union(const uint64_t *s1, const uint64_t *e1, size_t n1, const uint64_t *s2, const uint64_t *e2, size_t n2,
                          uint64_t *os, uint64_t *oe) try {
    return grc::store(grc::set_union(grc::make(s1, e1, n1), grc::make(s2, e2, n2)), os, oe);
} catch (...) { return 0; }
size_t gr_container_intersection(const uint64_t *s1, const uint64_t *e1, size_t n1, const uint64_t *s2, const uint64_t *e2, size_t n2,
                                 uint64_t *os, uint64_t *oe) try {
    return grc::store(grc::set_intersection(grc::make(s1, e1, n1), grc::make(s2, e2, n2)), os, oe);
} catch (...) { return 0; }
uint64_t gr_container_n_atoms(const uint64_t *s, const uint64_t *e, size_t n) { return grc::n_atoms(grc::make(s, e, n)); }
size_t gr_container_expand(const uint64_t *s, const uint64_t *e, size_t n, uint64_t *out) try {
    std::vector<uint64_t> v = grc::expand(grc::make(s, e, n));
    for (size_t k = 0; k < v.size(); ++k) out[k] = v[k];
    return v.size();
} catch (...) { return 0; }
int gr_container_isin(const uint64_t *s, const uint64_t *e, size_t n, uint64_t index) { return grc::isin(grc::make(s, e, n), index) ? 1 : 0; }
int gr_container_validate(const uint64_t *s, const uint64_t *e, size_t n, uint64_t n_atoms, uint64_t *bad_index) try {
    if (n && (!s || !e)) return GR_E_INVALID_ARG;
    return grc::valid_for(grc::make(s, e, n), n_atoms, bad_index) ? GR_OK : GR_E_OUT_OF_RANGE;
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ groups */
int gr_group_create_from_ranges(gr_ctx *c, const char *name, const uint64_t *s, const uint64_t *e, size_t n) try {
    if (!c) return GR_E_INVALID_ARG;
    (void)hipSetDevice(c->device);
    return install_group(c, name, grc::from_ranges(s, e, n, c->n));
} catch (...) { return gr_abi_guard(); }
int gr_group_create_from_indices(gr_ctx *c, const char *name, const uint64_t *indices, size_t n) try {
    if (!c) return GR_E_INVALID_ARG;
    (void)hipSetDevice(c->device);
    return install_group(c, name, grc::from_indices(std::vector<uint64_t>(indices, indices + n), c->n));
} catch (...) { return gr_abi_guard(); }
int gr_group_remove(gr_ctx *c, const char *name) try {
    if (!c || !name) return GR_E_INVALID_ARG;
    { int st = busy_check(c); if (st) return st; }
    auto it = c->groups.find(name);
    if (it == c->groups.end()) return fail(c, GR_E_GROUP_NOT_FOUND, name);
    group_release(c, it->second);
    c->groups.erase(it);
    c->epoch++;
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_group_exists(const gr_ctx *c, const char *name) { return (c && find_group(c, name)) ? 1 : 0; }
uint64_t gr_group_count(const gr_ctx *c) { return c ? (uint64_t)c->groups.size() : 0; }
int gr_group_name(const gr_ctx *c, uint64_t i, char *name, size_t cap) try {   // groups in name order (Groups::names_iter)
    if (!c || !name || cap == 0) return GR_E_INVALID_ARG;
    if (i >= c->groups.size()) return GR_E_OUT_OF_RANGE;
    auto it = c->groups.begin();
    std::advance(it, (long)i);
    const size_t n = std::min(cap - 1, it->first.size());
    memcpy(name, it->first.data(), n); name[n] = 0;
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_group_n_atoms(const gr_ctx *c, const char *name, uint64_t *n) try {
    const Group *g = c ? find_group(c, name) : nullptr;
    if (!g) return GR_E_GROUP_NOT_FOUND;
    if (n) *n = g->n;
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_group_n_blocks(const gr_ctx *c, const char *name, size_t *nb) try {
    const Group *g = c ? find_group(c, name) : nullptr;
    if (!g) return GR_E_GROUP_NOT_FOUND;
    if (nb) *nb = g->blocks.size();
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_group_blocks(const gr_ctx *c, const char *name, uint64_t *os, uint64_t *oe) try {
    const Group *g = c ? find_group(c, name) : nullptr;
    if (!g) return GR_E_GROUP_NOT_FOUND;
    grc::store(g->blocks, os, oe);
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ frames */
int gr_frame_upload(gr_ctx *c, uint32_t slot, const float *xyz, const float *box9) try {
    int st = slot_check(c, slot, 1, true); if (st) return st;
    if (!xyz) return fail(c, GR_E_INVALID_ARG, "xyz is NULL");
    (void)hipSetDevice(c->device);
    // boxes_host[slot] is the pinned SOURCE of the slot's previous box copy, which sits on the copy stream behind that
    // upload's 12 MB frame copy: it may only be rewritten once that copy has run -- whether or not a compute call has
    // meanwhile consumed the "upload pending" flag (upload -> batch_begin -> upload again into the same slot)
    if (c->ev_ready[slot]) HIPCHK(c, hipEventSynchronize(c->ev_ready[slot]));
    else HIPCHK(c, hipEventCreateWithFlags(&c->ev_ready[slot], hipEventDisableTiming));
    // the slot may still be read by kernels issued earlier: order the copy behind the last compute call that used it
    // (events of one stream complete in order, so a recycled ring entry only makes the wait conservative)
    if (c->slot_gen[slot]) HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_done_ring[c->slot_gen[slot] % 64], 0));
    // packed rvec[n] -> landing buffer -> pair-tiled slot, all on the copy stream (stream order keeps the one landing buffer safe)
    st = ensure_staging(c, &c->aos_up); if (st) return st;
    HIPCHK(c, hipMemcpyAsync(c->aos_up, xyz, c->n * 3 * sizeof(float), hipMemcpyHostToDevice, c->copy_stream));
    {
        const uint32_t ng = (uint32_t)(c->n_pad >> 2);
        k_tile<<<dim3((ng + 255) / 256), dim3(256), 0, c->copy_stream>>>(c->aos_up, c->frames + (size_t)slot * c->frame_stride, ng);
        HIPCHK(c, hipGetLastError());
    }
    st = set_box(c, slot, box9, c->copy_stream); if (st) return st;
    HIPCHK(c, hipEventRecord(c->ev_ready[slot], c->copy_stream));
    c->upload_pending[slot] = 1;
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_frame_upload_wait(gr_ctx *c, uint32_t slot) try {
    int st = slot_check(c, slot, 1, true); if (st) return st;
    if (c->ev_ready[slot]) HIPCHK(c, hipEventSynchronize(c->ev_ready[slot]));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_frame_download(gr_ctx *c, uint32_t slot, float *xyz) try {
    int st = slot_check(c, slot); if (st) return st;
    if (!xyz) return fail(c, GR_E_INVALID_ARG, "xyz is NULL");
    (void)hipSetDevice(c->device);
    SlotUse use(c, slot);
    st = untile_slot(c, slot); if (st) return st;
    HIPCHK(c, hipMemcpyAsync(xyz, c->aos_dl, c->n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_frame_set_box(gr_ctx *c, uint32_t slot, const float *box9) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    HIPCHK(c, sync_ingest(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // boxes_host[slot] may still feed an earlier async copy
    return set_box(c, slot, box9);
} catch (...) { return gr_abi_guard(); }
int gr_frame_get_box(const gr_ctx *c, uint32_t slot, float box9[9]) try {
    if (!c || slot >= c->n_slots) return GR_E_INVALID_ARG;
    if (!c->box9_set[slot]) return GR_E_NO_BOX;
    memcpy(box9, &c->box9_host[9 * (size_t)slot], 9 * sizeof(float));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_frame_copy(gr_ctx *c, uint32_t dst, uint32_t src) try {
    int st = slot_check(c, dst); if (st) return st;
    st = slot_check(c, src); if (st) return st;
    (void)hipSetDevice(c->device);
    SlotUse use_src(c, src), use_dst(c, dst);
    HIPCHK(c, hipMemcpyAsync(c->frames + (size_t)dst * c->frame_stride, c->frames + (size_t)src * c->frame_stride,
                             c->frame_stride * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, sync_ingest(c));                    // boxes_host[dst] may still feed an earlier upload's box copy
    return set_box(c, dst, c->box9_set[src] ? &c->box9_host[9 * (size_t)src] : nullptr);
} catch (...) { return gr_abi_guard(); }
void *gr_host_alloc(size_t bytes) { void *p = nullptr; return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
void gr_host_free(void *p) { if (p) (void)hipHostFree(p); }

/* ------------------------------------------------------------ centres */
// centre of one selection of one frame: the device work shared by the named-group calls (System::group_get_center ...,
// analysis.rs:52-320) and the anonymous-selection calls (the iterator-level AtomIteratorWithBox::get_center ..., iterators.rs:886-1438)
static int center_core(gr_ctx *c, uint32_t slot, const Group &g, int kind, int weighted, float out[3]) {
    int st;
    SlotUse use(c, slot);
    const GrSel sel = make_sel(g);
    if (small_ok(c, sel) && !(kind == GR_CENTER_PBC && center_onepass_ok(c, sel))) {
        // a small selection: one single-wave dispatch, the result read out of host-mapped memory (gr_small.h)
        const uint32_t seq = ++c->small_seq;
        k_center_small<<<dim3(1), dim3(64), 0, c->stream>>>(c->frames, c->frame_stride, slot, c->masses, sel, sel, c->boxes_dev,
                                                           kind == GR_CENTER_NAIVE ? 0 : (kind == GR_CENTER_ESTIMATE ? 1 : 2), weighted,
                                                           c->state_dev, c->small_state_dev, c->small_flag_dev, seq);
        HIPCHK(c, hipGetLastError());
        st = small_wait(c, seq); if (st) return st;
        use.quiet = true;                 // (the kernel has read everything it reads: nothing for a later upload into the slot to wait for)
        c->state_host[0] = *c->small_state;
        st = frame_status(c, c->state_host[0]); if (st) return st;
        if (out) { out[0] = c->state_host[0].com[0]; out[1] = c->state_host[0].com[1]; out[2] = c->state_host[0].com[2]; }
        return GR_OK;
    }
    st = state_reset(c, 1); if (st) return st;
    if (kind == GR_CENTER_NAIVE) st = center_stage(c, slot, 1, sel, 0, weighted, 0, 1);          // position first (:946-958)
    else if (kind == GR_CENTER_ESTIMATE) st = center_stage(c, slot, 1, sel, 1, weighted, 1, 1);  // mass first (:1324-1339)
    else st = center_onepass_ok(c, sel) ? pbc_center_onepass(c, slot, 1, sel, weighted) : pbc_center_stages(c, slot, 1, sel, weighted);
    if (st) return st;
    st = fetch_states(c, 1); if (st) return st;
    if (c->state_host[0].status == GR_ST_FALLBACK || c->state_host[0].status == GR_ST_AMBIG) {
        std::vector<GrFrameState> res;
        st = center_redo_fallbacks(c, slot, 1, sel, weighted, res); if (st) return st;
    }
    st = frame_status(c, c->state_host[0]); if (st) return st;
    if (out) { out[0] = c->state_host[0].com[0]; out[1] = c->state_host[0].com[1]; out[2] = c->state_host[0].com[2]; }
    return GR_OK;
}

int gr_group_center(gr_ctx *c, uint32_t slot, const char *group, int kind, int weighted, float out[3]) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    const Group *g = find_group(c, group);
    if (!g) return fail(c, GR_E_GROUP_NOT_FOUND, group ? group : "(null)");
    if (g->n == 0) return fail(c, GR_E_EMPTY_GROUP, group);               // analysis.rs:52-55
    if (kind != GR_CENTER_NAIVE && kind != GR_CENTER_ESTIMATE && kind != GR_CENTER_PBC) return fail(c, GR_E_INVALID_ARG, "unknown centre kind");
    if (kind != GR_CENTER_NAIVE) { st = box_check(c, slot); if (st) return st; }
    return center_core(c, slot, *g, kind, weighted, out);
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ distances */
int gr_group_distance(gr_ctx *c, uint32_t slot, const char *g1, const char *g2, int dim, float *out) try {
    float c1[3], c2[3];
    {   // both groups small (a protein and a ligand, two domains): their centres in ONE dispatch of two waves (gr_small.h); errors in the
        // reference's order -- everything about the first group before anything about the second -- whichever way the call goes
        const Group *a = (c && slot < c->n_slots) ? find_group(c, g1) : nullptr, *b = a ? find_group(c, g2) : nullptr;
        if (a && b && a->n && b->n && dim >= 0 && dim <= 7 && box_check(c, slot) == GR_OK && !c->in_flight) {
            const GrSel sa = make_sel(*a), sb = make_sel(*b);
            if (small_ok(c, sa) && small_ok(c, sb) && !center_onepass_ok(c, sa) && !center_onepass_ok(c, sb)) {
                (void)hipSetDevice(c->device);
                SlotUse use(c, slot);
                const uint32_t seq = ++c->small_seq;
                k_center_small<<<dim3(2), dim3(64), 0, c->stream>>>(c->frames, c->frame_stride, slot, c->masses, sa, sb, c->boxes_dev, 2, 0,
                                                                   c->state_dev, c->small_state_dev, c->small_flag_dev, seq);
                HIPCHK(c, hipGetLastError());
                int st = small_wait(c, seq, 2); if (st) return st;
                use.quiet = true;
                const GrFrameState r1 = c->small_state[0], r2 = c->small_state[1];
                st = frame_status(c, r1); if (st) return st;
                st = frame_status(c, r2); if (st) return st;
                if (out) *out = gr_distance(r1.com[0], r1.com[1], r1.com[2], r2.com[0], r2.com[1], r2.com[2], dim, c->boxes_host[slot]);
                return GR_OK;
            }
        }
    }
    int st = gr_group_center(c, slot, g1, GR_CENTER_PBC, 0, c1); if (st) return st;   // analysis.rs:354-355
    st = gr_group_center(c, slot, g2, GR_CENTER_PBC, 0, c2); if (st) return st;
    st = box_check(c, slot); if (st) return st;
    if (dim < 0 || dim > 7) return fail(c, GR_E_INVALID_ARG, "bad dimension");
    if (out) *out = gr_distance(c1[0], c1[1], c1[2], c2[0], c2[1], c2[2], dim, c->boxes_host[slot]);
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

static void batch_prechecks(gr_ctx *c, uint32_t s0, uint32_t nb, bool need_box, std::vector<int> &pre, std::vector<std::string> &msg);
// pair distances of `nb` consecutive slots in one launch; matrices `out_stride` floats apart; -> bad_host[4 f + 0 / 1]
static int pairdist_launch(gr_ctx *c, uint32_t s0, uint32_t nb, const GrSel &s1, const GrSel &s2, int dim, float *out_dev, size_t out_stride, const GrPdRed *red = nullptr) {
    SlotUse use(c, s0, nb);
    HIPCHK(c, hipMemsetAsync(c->bad_dev, 0xFF, 4 * (size_t)nb * sizeof(uint32_t), c->stream));
    if (s1.n && s2.n) {
        dim3 grid((s2.n + GR_WG * 4 - 1) / (GR_WG * 4), (s1.n + GR_PD_TI - 1) / GR_PD_TI, nb);
        const float *fr = c->frames + (size_t)s0 * c->frame_stride;
        // unrolled length of the minimum-image table: the smallest of 4 / 8 / 16 that holds every frame's entries
        int ncand = 0;
        for (uint32_t f = 0; f < nb; ++f) ncand = std::max(ncand, c->boxes_host[s0 + f].ncand);
        // a selection with itself: the symmetric kernel computes the tiles on and above the diagonal and writes every one twice
        // -- on non-orthogonal cells, where the kernel is bound by the VALU (measured, 1e4 x 1e4: XYZ 168 -> 111 us, XY 235 -> 163); the
        // orthorhombic loops are bound by the stores, and the mirror image's 256-byte runs cost more than half the arithmetic saves
        bool skewed = true;
        for (uint32_t f = 0; f < nb; ++f) skewed = skewed && !c->boxes_host[s0 + f].ortho;
        const GrPdRed none = { 0, 0, 0.0f, 0u, nullptr, 0 };
        const bool self = !red && c->pd_sym && skewed && s1.n == s2.n && s1.contiguous == s2.contiguous && s1.start == s2.start && s1.idx == s2.idx && s1.n >= 4 * GR_PDS_T;
        if (self) {
            const uint32_t nbk = (s1.n + GR_PDS_T - 1) / GR_PDS_T;
            dim3 tiles(nbk, nbk, nb);
            if (ncand <= 4) k_pairdist_sym<4><<<tiles, dim3(GR_WG), 0, c->stream>>>(fr, c->frame_stride, s1, c->boxes_dev + s0, dim, out_dev, out_stride, c->bad_dev);
            else if (ncand <= 8) k_pairdist_sym<8><<<tiles, dim3(GR_WG), 0, c->stream>>>(fr, c->frame_stride, s1, c->boxes_dev + s0, dim, out_dev, out_stride, c->bad_dev);
            else k_pairdist_sym<16><<<tiles, dim3(GR_WG), 0, c->stream>>>(fr, c->frame_stride, s1, c->boxes_dev + s0, dim, out_dev, out_stride, c->bad_dev);
        }
        else if (red) {     // the fused reducers: the same tiles, nothing of the matrix stored; a workgroup walks GR_PDR_ROW_TILES row tiles
            grid.y = (grid.y + GR_PDR_ROW_TILES - 1u) / GR_PDR_ROW_TILES;
            if (ncand <= 4) k_pairdist<4, true><<<grid, dim3(GR_WG), 0, c->stream>>>(fr, c->frame_stride, s1, s2, c->boxes_dev + s0, dim, nullptr, 0, c->bad_dev, *red);
            else if (ncand <= 8) k_pairdist<8, true><<<grid, dim3(GR_WG), 0, c->stream>>>(fr, c->frame_stride, s1, s2, c->boxes_dev + s0, dim, nullptr, 0, c->bad_dev, *red);
            else k_pairdist<16, true><<<grid, dim3(GR_WG), 0, c->stream>>>(fr, c->frame_stride, s1, s2, c->boxes_dev + s0, dim, nullptr, 0, c->bad_dev, *red);
        }
        else if (ncand <= 4) k_pairdist<4><<<grid, dim3(GR_WG), 0, c->stream>>>(fr, c->frame_stride, s1, s2, c->boxes_dev + s0, dim, out_dev, out_stride, c->bad_dev, none);
        else if (ncand <= 8) k_pairdist<8><<<grid, dim3(GR_WG), 0, c->stream>>>(fr, c->frame_stride, s1, s2, c->boxes_dev + s0, dim, out_dev, out_stride, c->bad_dev, none);
        else k_pairdist<16><<<grid, dim3(GR_WG), 0, c->stream>>>(fr, c->frame_stride, s1, s2, c->boxes_dev + s0, dim, out_dev, out_stride, c->bad_dev, none);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(c->bad_host, c->bad_dev, 4 * (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
}
// loop order of analysis.rs:414-424 with atom.rs:780-790: row atom first, then the column atoms of that row
static int pairdist_status(gr_ctx *c, uint32_t f, const GrSel &s1) {
    const uint32_t b1 = c->bad_host[4 * f], b2 = c->bad_host[4 * f + 1];
    if (b1 == GR_NOIDX && b2 == GR_NOIDX) return GR_OK;
    uint32_t idx;
    if (b1 == s1.start) idx = b1; else if (b2 != GR_NOIDX) idx = b2; else idx = b1;
    return fail(c, GR_E_NO_POSITION, "atom has no position", idx);
}
static int pairdist_run(gr_ctx *c, uint32_t slot, const GrSel &s1, const GrSel &s2, int dim, float *out_dev) {
    int st = pairdist_launch(c, slot, 1, s1, s2, dim, out_dev, 0); if (st) return st;
    return pairdist_status(c, 0, s1);
}

static int pairdist_reserve(gr_ctx *c, size_t need) {
    if (need > c->pd_cap) {
        if (c->pd_out) (void)hipFree(c->pd_out);
        c->pd_out = nullptr; c->pd_cap = 0;
        HIPCHK(c, hipMalloc(&c->pd_out, (need ? need : 1) * sizeof(float)));
        c->pd_cap = need;
    }
    return GR_OK;
}

int gr_group_all_distances_batch_device(gr_ctx *c, uint32_t first_slot, uint32_t n_frames, const char *g1, const char *g2, int dim,
                                        float **out_dev, uint64_t *n1, uint64_t *n2, int *status_out) try {
    int st = slot_check(c, first_slot, n_frames); if (st) return st;
    (void)hipSetDevice(c->device);
    const Group *a = find_group(c, g1); if (!a) return fail(c, GR_E_GROUP_NOT_FOUND, g1 ? g1 : "(null)");
    const Group *b = find_group(c, g2); if (!b) return fail(c, GR_E_GROUP_NOT_FOUND, g2 ? g2 : "(null)");
    if (dim < 0 || dim > 7) return fail(c, GR_E_INVALID_ARG, "bad dimension");
    const size_t per = (size_t)a->n * b->n;
    st = pairdist_reserve(c, per * n_frames); if (st) return st;
    const GrSel s1 = make_sel(*a), s2 = make_sel(*b);
    int first_err = GR_OK; std::string first_msg; uint64_t first_idx = 0;
    for (uint32_t b0 = 0; b0 < n_frames; b0 += GR_MAX_BATCH) {
        const uint32_t nb = std::min<uint32_t>(GR_MAX_BATCH, n_frames - b0), s0 = first_slot + b0;
        std::vector<int> pre; std::vector<std::string> msg;
        batch_prechecks(c, s0, nb, true, pre, msg);
        st = pairdist_launch(c, s0, nb, s1, s2, dim, c->pd_out + (size_t)b0 * per, per); if (st) return st;
        for (uint32_t f = 0; f < nb; ++f) {
            int s = pre[f];
            if (s != GR_OK) c->err = msg[f];
            else s = pairdist_status(c, f, s1);
            if (s != GR_OK && first_err == GR_OK) { first_err = s; first_msg = c->err; first_idx = c->err_index; }
            if (status_out) status_out[b0 + f] = s;
        }
    }
    if (out_dev) *out_dev = c->pd_out;
    if (n1) *n1 = a->n;
    if (n2) *n2 = b->n;
    if (first_err != GR_OK) { c->err = first_msg; c->err_index = first_idx; }
    return first_err;
} catch (...) { return gr_abi_guard(); }

int gr_group_all_distances_device(gr_ctx *c, uint32_t slot, const char *g1, const char *g2, int dim,
                                  float **out_dev, uint64_t *n1, uint64_t *n2) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    const Group *a = find_group(c, g1); if (!a) return fail(c, GR_E_GROUP_NOT_FOUND, g1 ? g1 : "(null)");
    const Group *b = find_group(c, g2); if (!b) return fail(c, GR_E_GROUP_NOT_FOUND, g2 ? g2 : "(null)");
    st = box_check(c, slot); if (st) return st;
    if (dim < 0 || dim > 7) return fail(c, GR_E_INVALID_ARG, "bad dimension");
    st = pairdist_reserve(c, (size_t)a->n * b->n); if (st) return st;
    st = pairdist_run(c, slot, make_sel(*a), make_sel(*b), dim, c->pd_out); if (st) return st;
    if (out_dev) *out_dev = c->pd_out;
    if (n1) *n1 = a->n;
    if (n2) *n2 = b->n;
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_device_read(gr_ctx *c, const void *dev, void *host, size_t bytes) try {
    if (!c) return GR_E_INVALID_ARG;
    if (!dev || !host) return fail(c, GR_E_INVALID_ARG, "NULL pointer");
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_group_all_distances(gr_ctx *c, uint32_t slot, const char *g1, const char *g2, int dim, float *out_host, size_t cap) try {
    float *dev = nullptr; uint64_t n1 = 0, n2 = 0;
    int st = gr_group_all_distances_device(c, slot, g1, g2, dim, &dev, &n1, &n2); if (st) return st;
    if ((size_t)(n1 * n2) > cap || !out_host) return fail(c, GR_E_INVALID_ARG, "output buffer too small");
    if (n1 * n2) {
        HIPCHK(c, hipMemcpyAsync(out_host, dev, (size_t)(n1 * n2) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

// group_all_distances followed by a reduction (analysis.rs:401-427 and what its callers do with the matrix, :1420-1451) WITHOUT the matrix:
// see k_pairdist<NC, true>.  `out`: GR_PD_MIN / GR_PD_MAX float[n_frames][per_row ? n1 : 1]; GR_PD_COUNT_BELOW uint64_t[n_frames][per_row ? n1 : 1];
// GR_PD_HIST uint64_t[n_frames][nbins].
int gr_group_all_distances_reduce_batch(gr_ctx *c, uint32_t first_slot, uint32_t n_frames, const char *g1, const char *g2, int dim, int op, int per_row,
                                        float param, uint32_t nbins, void *out, size_t out_capacity_bytes, int *status_out) try {
    int st = slot_check(c, first_slot, n_frames); if (st) return st;
    (void)hipSetDevice(c->device);
    const Group *a = find_group(c, g1); if (!a) return fail(c, GR_E_GROUP_NOT_FOUND, g1 ? g1 : "(null)");
    const Group *b = find_group(c, g2); if (!b) return fail(c, GR_E_GROUP_NOT_FOUND, g2 ? g2 : "(null)");
    if (dim < 0 || dim > 7) return fail(c, GR_E_INVALID_ARG, "bad dimension");
    if (op < GR_PD_MIN || op > GR_PD_HIST) return fail(c, GR_E_INVALID_ARG, "unknown reduction");
    if (op == GR_PD_HIST && (per_row || nbins == 0 || nbins > GR_PDR_MAX_BINS || !(param > 0.0f))) return fail(c, GR_E_INVALID_ARG, "histogram: 1 .. 4096 bins over (0, rmax), never per row");
    if (a->n == 0) return fail(c, GR_E_EMPTY_GROUP, g1);
    if (b->n == 0) return fail(c, GR_E_EMPTY_GROUP, g2);
    const bool wide = op == GR_PD_COUNT_BELOW || op == GR_PD_HIST;          // results the caller receives as 64-bit counts
    const size_t len = op == GR_PD_HIST ? nbins : (per_row ? a->n : 1);
    if (!out || out_capacity_bytes < (size_t)n_frames * len * (wide ? 8 : 4)) return fail(c, GR_E_INVALID_ARG, "output buffer too small");
    const bool sharded = op != GR_PD_HIST && !per_row;                         // whole-matrix values arrive in GR_PDR_SHARDS slots per frame
    const size_t words = sharded ? GR_PDR_SHARDS : len;
    const GrSel s1 = make_sel(*a), s2 = make_sel(*b);
    int first_err = GR_OK; std::string first_msg; uint64_t first_idx = 0;
    std::vector<uint32_t> host;
    for (uint32_t b0 = 0; b0 < n_frames; b0 += GR_MAX_BATCH) {
        const uint32_t nb = std::min<uint32_t>(GR_MAX_BATCH, n_frames - b0), s0 = first_slot + b0;
        st = pairdist_reserve(c, ((size_t)nb * words + 1) & ~(size_t)1); if (st) return st;
        uint32_t *acc = reinterpret_cast<uint32_t *>(c->pd_out);
        HIPCHK(c, hipMemsetAsync(acc, op == GR_PD_MIN ? 0xFF : 0x00, (size_t)nb * words * sizeof(uint32_t), c->stream));
        std::vector<int> pre; std::vector<std::string> msg;
        batch_prechecks(c, s0, nb, true, pre, msg);
        // per row of a group of some size: the groups change places and every lane keeps the values of its own four atoms (k_pairdist,
        // "transposed"); a handful of rows against many columns stays as it is (the lanes are the columns)
        const bool transposed = per_row && a->n >= 512;
        const GrPdRed red = { op, per_row ? (transposed ? 2 : 1) : 0, op == GR_PD_HIST ? (float)nbins / param : param, nbins, acc, words };
        st = transposed ? pairdist_launch(c, s0, nb, s2, s1, dim, nullptr, 0, &red) : pairdist_launch(c, s0, nb, s1, s2, dim, nullptr, 0, &red); if (st) return st;
        if (transposed) for (uint32_t f = 0; f < nb; ++f) std::swap(c->bad_host[4 * f], c->bad_host[4 * f + 1]);      // (first bad atom among the rows / the columns)
        host.resize((size_t)nb * words);
        HIPCHK(c, hipMemcpyAsync(host.data(), acc, host.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (uint32_t f = 0; f < nb; ++f) {
            int s = pre[f];
            if (s != GR_OK) c->err = msg[f];
            else s = pairdist_status(c, f, s1);
            if (s != GR_OK && first_err == GR_OK) { first_err = s; first_msg = c->err; first_idx = c->err_index; }
            if (status_out) status_out[b0 + f] = s;
            const uint32_t *h = host.data() + (size_t)f * words;
            if (wide) {
                uint64_t *o = static_cast<uint64_t *>(out) + (size_t)(b0 + f) * len;
                if (sharded) { uint64_t v = 0; for (size_t k = 0; k < words; ++k) v += h[k]; o[0] = v; }
                else for (size_t k = 0; k < len; ++k) o[k] = h[k];
            } else {
                float *o = static_cast<float *>(out) + (size_t)(b0 + f) * len;
                if (sharded) { uint32_t v = h[0]; for (size_t k = 1; k < words; ++k) v = op == GR_PD_MIN ? std::min(v, h[k]) : std::max(v, h[k]); o[0] = gr_key_f32(v); }
                else for (size_t k = 0; k < len; ++k) o[k] = gr_key_f32(h[k]);
            }
        }
    }
    if (first_err != GR_OK) { c->err = first_msg; c->err_index = first_idx; }
    return first_err;
} catch (...) { return gr_abi_guard(); }
int gr_group_all_distances_reduce(gr_ctx *c, uint32_t slot, const char *g1, const char *g2, int dim, int op, int per_row, float param, uint32_t nbins,
                                  void *out, size_t out_capacity_bytes) try {
    int st = slot_check(c, slot); if (st) return st;
    st = box_check(c, slot); if (st) return st;
    return gr_group_all_distances_reduce_batch(c, slot, 1, g1, g2, dim, op, per_row, param, nbins, out, out_capacity_bytes, nullptr);
} catch (...) { return gr_abi_guard(); }

int gr_atoms_distance(gr_ctx *c, uint32_t slot, uint64_t i1, uint64_t i2, int dim, float *out) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    if (i1 >= c->n) return fail(c, GR_E_OUT_OF_RANGE, "atom index out of range", i1);   // analysis.rs:465-466
    if (i2 >= c->n) return fail(c, GR_E_OUT_OF_RANGE, "atom index out of range", i2);
    st = box_check(c, slot); if (st) return st;
    if (dim < 0 || dim > 7) return fail(c, GR_E_INVALID_ARG, "bad dimension");
    GrSel s1 = { 1u, 1u, (uint32_t)i1, (uint32_t)i1 >> 8, nullptr }, s2 = { 1u, 1u, (uint32_t)i2, (uint32_t)i2 >> 8, nullptr };
    if (!c->pd_out) { HIPCHK(c, hipMalloc(&c->pd_out, 16 * sizeof(float))); c->pd_cap = 16; }
    st = pairdist_run(c, slot, s1, s2, dim, c->pd_out); if (st) return st;
    float v = 0.f;
    HIPCHK(c, hipMemcpy(&v, c->pd_out, sizeof(float), hipMemcpyDeviceToHost));
    if (out) *out = v;
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ geometry selection */
static bool name_is_valid(const char *name) {   // auxiliary.rs:37-51
    if (!name) return false;
    bool blank = true;
    for (const char *p = name; *p; ++p) {
        if (strchr("'\"&|!@()<>=", *p)) return false;
        if (!isspace((unsigned char)*p)) blank = false;
    }
    return !blank;
}
int gr_shape_sphere(gr_shape *s, const float pos[3], float radius) try {
    if (!s || !pos) return GR_E_INVALID_ARG;
    memset(s, 0, sizeof *s); s->kind = GR_SHAPE_SPHERE; memcpy(s->position, pos, 12); s->size[0] = radius; return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_shape_rectangular(gr_shape *s, const float pos[3], float x, float y, float z) try {
    if (!s || !pos) return GR_E_INVALID_ARG;
    memset(s, 0, sizeof *s); s->kind = GR_SHAPE_RECTANGULAR; memcpy(s->position, pos, 12); s->size[0] = x; s->size[1] = y; s->size[2] = z; return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_shape_cylinder(gr_shape *s, const float pos[3], float radius, float height, int orientation) try {
    if (!s || !pos || orientation < GR_DIM_X || orientation > GR_DIM_Z) return GR_E_INVALID_ARG;
    memset(s, 0, sizeof *s); s->kind = GR_SHAPE_CYLINDER; memcpy(s->position, pos, 12); s->size[0] = radius; s->size[1] = height;
    s->orientation = orientation;
    s->plane = orientation == GR_DIM_X ? GR_DIM_YZ : (orientation == GR_DIM_Y ? GR_DIM_XZ : GR_DIM_XY);
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_shape_triangular_prism(gr_shape *s, const float b1[3], const float b2[3], const float b3[3], float height) try {
    if (!s || !b1 || !b2 || !b3) return GR_E_INVALID_ARG;
    static const int orient[3] = { GR_DIM_X, GR_DIM_Y, GR_DIM_Z }, plane[3] = { GR_DIM_YZ, GR_DIM_XZ, GR_DIM_XY };
    int found = -1;
    for (int a = 0; a < 3; ++a)
        if (b1[a] == b2[a] && b2[a] == b3[a]) { if (found >= 0) return GR_E_INVALID_ARG; found = a; }
    if (found < 0) return GR_E_INVALID_ARG;
    memset(s, 0, sizeof *s); s->kind = GR_SHAPE_TRIANGULAR_PRISM;
    memcpy(s->position, b1, 12); memcpy(s->base2, b2, 12); memcpy(s->base3, b3, 12); s->size[0] = height;
    s->orientation = orient[found]; s->plane = plane[found];
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
static bool shape_to_dev(const gr_shape &h, int naive, GrShapeDev *d) {
    if (h.kind < GR_SHAPE_SPHERE || h.kind > GR_SHAPE_TRIANGULAR_PRISM) return false;
    if (naive && h.kind == GR_SHAPE_TRIANGULAR_PRISM) return false;                 // no NaiveShape for the prism (shape.rs:466-505)
    if (h.kind >= GR_SHAPE_CYLINDER && (h.orientation < GR_DIM_X || h.orientation > GR_DIM_Z || h.plane < GR_DIM_XY || h.plane > GR_DIM_YZ)) return false;
    d->kind = h.kind; d->px = h.position[0]; d->py = h.position[1]; d->pz = h.position[2];
    d->a = h.size[0]; d->b = h.size[1]; d->c = h.size[2];
    d->b2x = h.base2[0]; d->b2y = h.base2[1]; d->b2z = h.base2[2]; d->b3x = h.base3[0]; d->b3y = h.base3[1]; d->b3z = h.base3[2];
    d->orientation = h.orientation; d->plane = h.plane;
    return true;
}
int gr_shape_inside(const gr_shape *s, const float point[3], const float box9[9], int naive, int *inside) try {
    if (!s || !point || !inside || (!naive && !box9)) return GR_E_INVALID_ARG;
    GrShapeDev d;
    if (!shape_to_dev(*s, naive, &d)) return GR_E_INVALID_ARG;
    if (naive) { *inside = gr_shape_inside_naive(d, point[0], point[1], point[2]) ? 1 : 0; return GR_OK; }
    GrBox b;
    if (!gr_box_setup(box9, &b)) return GR_E_ZERO_BOX;
    *inside = gr_shape_inside_pbc(d, point[0], point[1], point[2], b) ? 1 : 0;   // (non-orthogonal boxes: the image-enumeration extension, gr_shape.h)
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
// atoms of `g` (in its iteration order) that have a position and lie inside every shape -> `picked`
static int geometry_filter(gr_ctx *c, uint32_t slot, const Group &g, const gr_shape *shapes, size_t ns, int naive, std::vector<uint64_t> &picked) {
    if (ns > GR_MAX_SHAPES || (ns && !shapes)) return fail(c, GR_E_INVALID_ARG, "too many shapes");
    GrShapeSet set; set.n = (int)ns; set.naive = naive ? 1 : 0;
    for (size_t q = 0; q < ns; ++q) if (!shape_to_dev(shapes[q], naive, &set.s[q])) return fail(c, GR_E_INVALID_ARG, "invalid shape");
    picked.clear();
    if (!naive)
        for (size_t q = 0; q < ns; ++q)
            if (gr_shape_tric_walk(set.s[q], c->boxes_host[slot]) > GR_SHAPE_WALK_MAX)
                return fail(c, GR_E_UNSUPPORTED_BOX, "the shape reaches more lattice images of this cell than the geometry selection enumerates");
    if (!g.n) return GR_OK;
    const GrSel sel = make_sel(g);
    const size_t words = ((size_t)g.n + 63) / 64;
    // (the mask buffers live with the context: a hipMalloc / hipFree pair per call cost more than the kernel and the copy together, and the
    //  copy lands in pinned memory)
    if (words > c->shape_mask_cap) {
        if (c->shape_mask_dev) (void)hipFree(c->shape_mask_dev);
        if (c->shape_mask_host) (void)hipHostFree(c->shape_mask_host);
        c->shape_mask_dev = nullptr; c->shape_mask_host = nullptr; c->shape_mask_cap = 0;
        const size_t cap = words + words / 8 + 64;
        HIPCHK(c, hipMalloc(&c->shape_mask_dev, cap * sizeof(unsigned long long)));
        HIPCHK(c, hipHostMalloc(&c->shape_mask_host, cap * sizeof(unsigned long long), hipHostMallocDefault));
        c->shape_mask_cap = cap;
    }
    unsigned long long *mask_dev = c->shape_mask_dev;
    if (!c->shape_set_dev) HIPCHK(c, hipMalloc(&c->shape_set_dev, sizeof(GrShapeSet)));
    HIPCHK(c, hipMemcpyAsync(c->shape_set_dev, &set, sizeof set, hipMemcpyHostToDevice, c->stream));     // (pageable source: staged by the runtime before the call returns)
    {
        SlotUse use(c, slot);
        k_shape_mask<<<dim3((unsigned)((g.n + 255) / 256)), dim3(256), 0, c->stream>>>(c->frames + (size_t)slot * c->frame_stride, sel, c->boxes_dev + slot, c->shape_set_dev, mask_dev);
    }
    const unsigned long long *mask = c->shape_mask_host;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(c->shape_mask_host, mask_dev, words * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { c->err = std::string("geometry selection: ") + hipGetErrorString(e); return GR_E_HIP; }
    // ordinal -> atom index in the source group's iteration order (AtomContainer::iter, container.rs:381-411): the set bits of each word
    // (count-trailing-zeros walk: the cost follows the atoms picked, not the atoms tested), block by block
    size_t picked_n = 0;
    for (size_t w = 0; w < words; ++w) picked_n += (size_t)__builtin_popcountll(mask[w]);
    picked.reserve(picked_n);
    size_t j0 = 0;                                            // ordinal of the block's first atom
    for (const auto &blk : g.blocks) {
        const size_t len = (size_t)(blk.second - blk.first + 1), j1 = j0 + len;
        for (size_t w = j0 >> 6; w <= ((j1 - 1) >> 6); ++w) {
            unsigned long long bits = mask[w];
            if (!bits) continue;
            const size_t base = w << 6;
            if (base < j0) bits &= ~0ull << (j0 - base);                       // ordinals before the block
            if (base + 64 > j1) bits &= ~0ull >> (base + 64 - j1);             // ... and behind it
            while (bits) {
                const size_t j = base + (size_t)__builtin_ctzll(bits);
                bits &= bits - 1;
                picked.push_back(blk.first + (uint64_t)(j - j0));
            }
        }
        j0 = j1;
    }
    return GR_OK;
}
// the box gate of Shape::inside with PBC (groups.rs:104-110 / the iterator's get_simbox_unwrap): a usable box; in strict mode orthogonal
static int geometry_box_check(gr_ctx *c, uint32_t slot) {
    if (c->box_status[slot] == GR_E_NO_BOX) return fail(c, GR_E_NO_BOX, "simulation box does not exist");   // :104-106
    if (c->box_status[slot] != GR_OK) return fail(c, c->box_status[slot], "invalid simulation box");
    if (c->strict && !c->boxes_host[slot].ortho) return fail(c, GR_E_NOT_ORTHOGONAL, "simulation box is not orthogonal"); // :108-110
    return GR_OK;
}
int gr_group_create_from_geometries(gr_ctx *c, uint32_t slot, const char *name, const char *source, const gr_shape *shapes, size_t ns, int naive) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    if (!name_is_valid(name)) return fail(c, GR_E_INVALID_NAME, name ? name : "(null)");                 // groups.rs:100-102
    st = geometry_box_check(c, slot); if (st) return st;
    const Group *g = find_group(c, source);
    if (!g) return fail(c, GR_E_GROUP_NOT_FOUND, source ? source : "(null)");                               // InvalidQuery(GroupNotFound)
    std::vector<uint64_t> picked;
    st = geometry_filter(c, slot, *g, shapes, ns, naive, picked); if (st) return st;
    return install_group(c, name, grc::from_indices(picked, c->n));
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ translate / wrap / centre */
// (check_state: the frame's state -- the centre estimate that precedes the translation on the stream -- is fetched with the same
//  synchronisation, and its error, if any, comes first: the kernel has left such a frame alone)
static int translate_core(gr_ctx *c, uint32_t slot, const Group &g, const float *v, int use_state, int dim_mask, bool check_state = false, bool bad_is_set = false) {
    int st = box_check(c, slot); if (st) return st;
    if (g.n == 0 && !check_state) return GR_OK;
    const GrSel sel = make_sel(g);
    SlotUse use(c, slot);
    if (!bad_is_set) HIPCHK(c, hipMemsetAsync(c->bad_dev, 0xFF, 4 * sizeof(uint32_t), c->stream));
    const uint64_t units = sel.contiguous ? ((uint64_t)sel.n + 3) / 4 + 128 : (sel.masked & 2u) ? ((uint64_t)sel.span + 3) / 4 + 128 : sel.n;   // (4-atom groups of the block / of a masked selection's span; list entries)
    uint32_t nwg = (uint32_t)std::min<uint64_t>((units + GR_WG - 1) / GR_WG, 4096);
    if (g.n != 0 && sel.contiguous && c->translate_rows && c->boxes_host[slot].ortho)
        k_translate_wrap_rows<<<dim3((((sel.start + sel.n - 1u) >> 8) - (sel.start >> 8) + 1u) * 3u), dim3(64), 0, c->stream>>>(c->frames + (size_t)slot * c->frame_stride, c->frame_stride, sel, c->boxes_dev + slot, c->state_dev, use_state, dim_mask,
                                                                                                                       v ? v[0] : 0.f, v ? v[1] : 0.f, v ? v[2] : 0.f, c->bad_dev);
    else
    k_translate_wrap<<<dim3(nwg), dim3(GR_WG), 0, c->stream>>>(c->frames + (size_t)slot * c->frame_stride, c->frame_stride, sel, c->boxes_dev + slot, c->state_dev, use_state, dim_mask, v ? v[0] : 0.f, v ? v[1] : 0.f, v ? v[2] : 0.f, c->bad_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->bad_host, c->bad_dev, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (check_state) HIPCHK(c, hipMemcpyAsync(c->state_host, c->state_dev, sizeof(GrFrameState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (check_state) { st = frame_status(c, c->state_host[0]); if (st) return st; }
    if (c->bad_host[0] != GR_NOIDX) return fail(c, GR_E_NO_POSITION, "atom has no position", c->bad_host[0]);
    return GR_OK;
}
static int translate_impl(gr_ctx *c, uint32_t slot, const char *group, const float *v, int use_state, int dim_mask, bool check_state = false, bool bad_is_set = false) {
    const Group *g = find_group(c, group ? group : "all");
    if (!g) return fail(c, GR_E_GROUP_NOT_FOUND, group);
    return translate_core(c, slot, *g, v, use_state, dim_mask, check_state, bad_is_set);
}

int gr_group_translate(gr_ctx *c, uint32_t slot, const char *group, const float v[3]) try {
    int st = slot_check(c, slot); if (st) return st;
    if (!v) return fail(c, GR_E_INVALID_ARG, "vector is NULL");
    (void)hipSetDevice(c->device);
    return translate_impl(c, slot, group, v, 0, 7);
} catch (...) { return gr_abi_guard(); }
int gr_group_wrap(gr_ctx *c, uint32_t slot, const char *group) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    const float z[3] = { 0.f, 0.f, 0.f };
    return translate_impl(c, slot, group, z, 0, 7);
} catch (...) { return gr_abi_guard(); }
int gr_atoms_center(gr_ctx *c, uint32_t slot, const char *ref_group, int dim, int weighted) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    const Group *g = find_group(c, ref_group);
    if (!g) return fail(c, GR_E_GROUP_NOT_FOUND, ref_group ? ref_group : "(null)");
    if (g->n == 0) return fail(c, GR_E_EMPTY_GROUP, ref_group);
    st = box_check(c, slot); if (st) return st;
    if (dim < 0 || dim > 7) return fail(c, GR_E_INVALID_ARG, "bad dimension");
    static const int mask[8] = { 0, 1, 2, 4, 3, 5, 6, 7 };
    { SlotUse use(c, slot); }   // the estimate below and translate_impl each bracket themselves; this orders a pending upload first
    const GrSel csel = make_sel(*g);
    if (small_ok(c, csel)) {
        // a small reference group: its single-wave estimate starts the frame's state itself and sets the translate kernel's words -- two
        // dispatches (estimate, translation) instead of four
        k_center_small_stage<<<dim3(1), dim3(64), 0, c->stream>>>(c->frames, c->frame_stride, slot, c->masses, csel, c->boxes_dev, 1, weighted, 1, 0, c->state_dev, 0, c->bad_dev);
        HIPCHK(c, hipGetLastError());
        return translate_impl(c, slot, "all", nullptr, 1, mask[dim], true, true);
    }
    st = state_reset(c, 1); if (st) return st;
    st = center_stage(c, slot, 1, csel, 1, weighted, 1, 0); if (st) return st;   // group_estimate_center / _com
    // (the translation follows the estimate on the stream and leaves the frame alone when the estimate failed: ONE synchronisation for
    //  both, the estimate's error reported first)
    return translate_impl(c, slot, "all", nullptr, 1, mask[dim], true);
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ anonymous selections (the iterator-level surface) */
// a selection handed over as AtomContainer blocks (container.rs:23-31) instead of a group name: validated, expanded for the
// device when it is not one block, used for one call, released
struct TempSel {
    gr_ctx *c; Group g; bool ok = false;
    TempSel(gr_ctx *ctx, const uint64_t *s, const uint64_t *e, size_t n, int *st) : c(ctx) {
        if (n && (!s || !e)) { *st = fail(c, GR_E_INVALID_ARG, "selection blocks are NULL"); return; }
        *st = group_build(c, grc::make(s, e, n), &g);
        ok = *st == GR_OK;
    }
    ~TempSel() { if (ok && (g.idx_dev || g.mask_dev)) { (void)hipStreamSynchronize(c->stream); if (g.idx_dev) (void)hipFree(g.idx_dev); if (g.mask_dev) (void)hipFree(g.mask_dev); } }
};

int gr_sel_center(gr_ctx *c, uint32_t slot, const uint64_t *start, const uint64_t *end, size_t n_blocks, int kind, int weighted, float out[3]) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    if (kind != GR_CENTER_NAIVE && kind != GR_CENTER_ESTIMATE && kind != GR_CENTER_PBC) return fail(c, GR_E_INVALID_ARG, "unknown centre kind");
    if (kind != GR_CENTER_NAIVE) { st = box_check(c, slot); if (st) return st; }      // simbox_check first (iterators.rs:1153,1238,1315,1405)
    TempSel sel(c, start, end, n_blocks, &st); if (st) return st;
    if (sel.g.n == 0) { if (out) out[0] = out[1] = out[2] = NAN; return GR_OK; }      // an empty iterator yields NaN, not an error (:1186-1188)
    return center_core(c, slot, sel.g, kind, weighted, out);
} catch (...) { return gr_abi_guard(); }

int gr_sel_translate(gr_ctx *c, uint32_t slot, const uint64_t *start, const uint64_t *end, size_t n_blocks, const float v[3]) try {
    int st = slot_check(c, slot); if (st) return st;
    if (!v) return fail(c, GR_E_INVALID_ARG, "vector is NULL");
    (void)hipSetDevice(c->device);
    st = box_check(c, slot); if (st) return st;                                          // :1521-1522
    TempSel sel(c, start, end, n_blocks, &st); if (st) return st;
    return translate_core(c, slot, sel.g, v, 0, 7);
} catch (...) { return gr_abi_guard(); }

int gr_sel_wrap(gr_ctx *c, uint32_t slot, const uint64_t *start, const uint64_t *end, size_t n_blocks) try {
    const float z[3] = { 0.f, 0.f, 0.f };
    return gr_sel_translate(c, slot, start, end, n_blocks, z);                           // :1548-1553
} catch (...) { return gr_abi_guard(); }

int gr_sel_all_distances(gr_ctx *c, uint32_t slot, const uint64_t *s1, const uint64_t *e1, size_t n1, const uint64_t *s2, const uint64_t *e2, size_t n2,
                         int dim, float *out_host, size_t cap) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    st = box_check(c, slot); if (st) return st;
    if (dim < 0 || dim > 7) return fail(c, GR_E_INVALID_ARG, "bad dimension");
    TempSel a(c, s1, e1, n1, &st); if (st) return st;
    TempSel b(c, s2, e2, n2, &st); if (st) return st;
    const size_t total = (size_t)a.g.n * b.g.n;
    if (total > cap || (total && !out_host)) return fail(c, GR_E_INVALID_ARG, "output buffer too small");
    st = pairdist_reserve(c, total); if (st) return st;
    st = pairdist_run(c, slot, make_sel(a.g), make_sel(b.g), dim, c->pd_out); if (st) return st;
    if (total) {
        HIPCHK(c, hipMemcpyAsync(out_host, c->pd_out, total * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_sel_filter_geometry(gr_ctx *c, uint32_t slot, const uint64_t *start, const uint64_t *end, size_t n_blocks, const gr_shape *shapes, size_t n_shapes, int naive,
                           uint64_t *out_start, uint64_t *out_end, size_t cap_blocks, size_t *n_out_blocks, uint64_t *n_out_atoms) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    if (!naive) { st = geometry_box_check(c, slot); if (st) return st; }                 // filter_geometry panics without a box (:1098); an error here
    TempSel sel(c, start, end, n_blocks, &st); if (st) return st;
    std::vector<uint64_t> picked;
    st = geometry_filter(c, slot, sel.g, shapes, n_shapes, naive, picked); if (st) return st;
    const std::vector<grc::Block> blocks = grc::from_indices(picked, c->n);
    if (n_out_blocks) *n_out_blocks = blocks.size();
    if (n_out_atoms) *n_out_atoms = picked.size();
    if (blocks.size() > cap_blocks) return (out_start || out_end) ? fail(c, GR_E_INVALID_ARG, "block buffers too small") : GR_OK;   // NULL buffers: count only
    if (out_start && out_end) grc::store(blocks, out_start, out_end);
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ cut-off pair search */
int gr_group_pairs_within(gr_ctx *c, uint32_t slot, const char *g1, const char *g2, float cutoff, uint64_t max_pairs,
                          uint32_t *i_out, uint32_t *j_out, float *d_out, uint64_t *n_pairs) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    const Group *a = find_group(c, g1); if (!a) return fail(c, GR_E_GROUP_NOT_FOUND, g1 ? g1 : "(null)");
    const Group *b = find_group(c, g2); if (!b) return fail(c, GR_E_GROUP_NOT_FOUND, g2 ? g2 : "(null)");
    if (!(cutoff > 0.0f)) return fail(c, GR_E_INVALID_ARG, "cell size (cut-off) must be positive");              // cellgrid.rs:323-325
    if (c->box_status[slot] == GR_E_NO_BOX) return fail(c, GR_E_NO_BOX, "simulation box does not exist");            // check_box :411-430
    if (c->box_status[slot] != GR_OK) return fail(c, c->box_status[slot], "invalid simulation box");
    if (c->strict && !c->boxes_host[slot].ortho) return fail(c, GR_E_NOT_ORTHOGONAL, "simulation box is not orthogonal");   // the reference's gate (cellgrid.rs:423)
    if (n_pairs) *n_pairs = 0;
    if (a->n == 0 || b->n == 0) return GR_OK;
    const GrBox &box = c->boxes_host[slot];
    const GrCellGrid grid = gr_cellgrid_make(box, cutoff);
    const GrSel s1 = make_sel(*a), s2 = make_sel(*b);
    const uint32_t n1 = (uint32_t)a->n, n2 = (uint32_t)b->n;
    const float *xyz = c->frames + (size_t)slot * c->frame_stride;
    // one device allocation carved up: keys/vals in + out, cell starts, counts + offsets, 2 error words, sort / scan scratch
    size_t tmp_sort = 0, tmp_scan = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp_sort, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, n2, 0, 32, c->stream);
    (void)rocprim::exclusive_scan(nullptr, tmp_scan, (unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull, n1, rocprim::plus<unsigned long long>(), c->stream);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_keys = 0, o_vals = o_keys + al(4 * (size_t)n2), o_skeys = o_vals + al(4 * (size_t)n2), o_svals = o_skeys + al(4 * (size_t)n2),
                 o_starts = o_svals + al(4 * (size_t)n2), o_counts = o_starts + al(4 * ((size_t)grid.ncells + 1)), o_offs = o_counts + al(8 * (size_t)n1),
                 o_bad = o_offs + al(8 * (size_t)n1), o_tmp = o_bad + 256, total_bytes = o_tmp + al(std::max(tmp_sort, tmp_scan));
    unsigned char *ws = nullptr;
    HIPCHK(c, hipMalloc(&ws, total_bytes));
    uint32_t *keys = (uint32_t *)(ws + o_keys), *vals = (uint32_t *)(ws + o_vals), *skeys = (uint32_t *)(ws + o_skeys), *svals = (uint32_t *)(ws + o_svals),
             *starts = (uint32_t *)(ws + o_starts), *bad = (uint32_t *)(ws + o_bad);
    unsigned long long *counts = (unsigned long long *)(ws + o_counts), *offs = (unsigned long long *)(ws + o_offs);
    uint32_t *oi = nullptr, *oj = nullptr; float *od = nullptr;
    auto cleanup = [&]() { (void)hipFree(ws); if (oi) (void)hipFree(oi); if (oj) (void)hipFree(oj); if (od) (void)hipFree(od); };
    auto herr = [&](hipError_t e, const char *what) { c->err = std::string(what) + ": " + hipGetErrorString(e); cleanup(); return GR_E_HIP; };
    SlotUse use(c, slot);
    hipError_t e = hipMemsetAsync(bad, 0xFF, 8, c->stream);
    if (e != hipSuccess) return herr(e, "memset");
    k_cg_assign<<<dim3((n2 + 255) / 256), dim3(256), 0, c->stream>>>(xyz, s2, box, grid, keys, vals, bad);
    size_t tsz = std::max(tmp_sort, tmp_scan);
    e = rocprim::radix_sort_pairs(ws + o_tmp, tsz, keys, skeys, vals, svals, n2, 0, 32, c->stream);
    if (e != hipSuccess) return herr(e, "radix_sort_pairs");
    k_cg_starts<<<dim3((grid.ncells + 1 + 255) / 256), dim3(256), 0, c->stream>>>(skeys, n2, grid.ncells, starts);
    k_cg_pairs<false><<<dim3((n1 + 255) / 256), dim3(256), 0, c->stream>>>(xyz, s1, box, grid, cutoff, svals, starts, counts, nullptr, 0ull, nullptr, nullptr, nullptr, bad);
    tsz = std::max(tmp_sort, tmp_scan);
    e = rocprim::exclusive_scan(ws + o_tmp, tsz, counts, offs, 0ull, (size_t)n1, rocprim::plus<unsigned long long>(), c->stream);
    if (e != hipSuccess) return herr(e, "exclusive_scan");
    unsigned long long last[2] = { 0, 0 }; uint32_t badh[2] = { GR_NOIDX, GR_NOIDX };
    e = hipMemcpyAsync(&last[0], offs + (n1 - 1), 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&last[1], counts + (n1 - 1), 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(badh, bad, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return herr(e, "pair count read-back");
    if (badh[0] != GR_NOIDX || badh[1] != GR_NOIDX) {   // ordinals -> atom indices, group2 (grid construction) first
        const bool second = badh[0] != GR_NOIDX;
        const uint64_t ord = second ? badh[0] : badh[1];
        uint64_t atom = 0, k = 0;
        for (const auto &blk : (second ? b : a)->blocks) { const uint64_t len = blk.second - blk.first + 1; if (ord < k + len) { atom = blk.first + (ord - k); break; } k += len; }
        cleanup();
        return fail(c, GR_E_NO_POSITION, "atom has no position", atom);
    }
    const unsigned long long total = last[0] + last[1];
    if (n_pairs) *n_pairs = total;
    const unsigned long long cap = std::min<unsigned long long>(total, max_pairs);
    if (cap > 0 && i_out && j_out && d_out) {
        e = hipMalloc(&oi, cap * 4); if (e == hipSuccess) e = hipMalloc(&oj, cap * 4); if (e == hipSuccess) e = hipMalloc(&od, cap * 4);
        if (e != hipSuccess) return herr(e, "pair buffers");
        k_cg_pairs<true><<<dim3((n1 + 255) / 256), dim3(256), 0, c->stream>>>(xyz, s1, box, grid, cutoff, svals, starts, counts, offs, cap, oi, oj, od, bad);
        e = hipMemcpyAsync(i_out, oi, cap * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(j_out, oj, cap * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_out, od, cap * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return herr(e, "pair read-back");
    }
    cleanup();
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ the same per-frame calls over a batch of slots */
// host checks of one batch in the reference's order; pre[f] = GR_OK or the frame's error (messages kept for the first)
static void batch_prechecks(gr_ctx *c, uint32_t s0, uint32_t nb, bool need_box, std::vector<int> &pre, std::vector<std::string> &msg) {
    pre.assign(nb, GR_OK); msg.assign(nb, std::string());
    for (uint32_t f = 0; f < nb; ++f) {
        if (!need_box) continue;
        const int s = box_check(c, s0 + f);
        pre[f] = s;
        if (s != GR_OK) msg[f] = c->err;
    }
}
// the frames' states of a batched call start from the host-side checks: zeroed by a kernel when every frame passed them (the usual
// case), copied out of pinned memory otherwise -- that copy costs ~30 us of host time before the call's first kernel can be queued
static int states_from_prechecks(gr_ctx *c, uint32_t nb, const std::vector<int> &pre) {
    bool all_ok = true;
    for (uint32_t f = 0; f < nb; ++f) all_ok = all_ok && pre[f] == GR_OK;
    if (all_ok) return state_reset(c, nb);
    for (uint32_t f = 0; f < nb; ++f) { GrFrameState z = {}; z.err_index = GR_NOIDX; z.status = pre[f]; c->state_host[f] = z; }
    HIPCHK(c, hipMemcpyAsync(c->state_dev, c->state_host, nb * sizeof(GrFrameState), hipMemcpyHostToDevice, c->stream));
    return GR_OK;
}
int gr_group_center_batch(gr_ctx *c, uint32_t first_slot, uint32_t n_frames, const char *group, int kind, int weighted, float *out, int *status_out) try {
    int st = slot_check(c, first_slot, n_frames); if (st) return st;
    (void)hipSetDevice(c->device);
    const Group *g = find_group(c, group);
    if (!g) return fail(c, GR_E_GROUP_NOT_FOUND, group ? group : "(null)");
    if (g->n == 0) return fail(c, GR_E_EMPTY_GROUP, group);
    if (kind != GR_CENTER_NAIVE && kind != GR_CENTER_ESTIMATE && kind != GR_CENTER_PBC) return fail(c, GR_E_INVALID_ARG, "unknown centre kind");
    const GrSel sel = make_sel(*g);
    int first_err = GR_OK; std::string first_msg; uint64_t first_idx = 0;
    for (uint32_t b0 = 0; b0 < n_frames; b0 += GR_MAX_BATCH) {
        const uint32_t nb = std::min<uint32_t>(GR_MAX_BATCH, n_frames - b0), s0 = first_slot + b0;
        std::vector<int> pre; std::vector<std::string> msg;
        batch_prechecks(c, s0, nb, kind != GR_CENTER_NAIVE, pre, msg);
        SlotUse use(c, s0, nb);
        st = states_from_prechecks(c, nb, pre); if (st) return st;
        if (kind == GR_CENTER_NAIVE) st = center_stage(c, s0, nb, sel, 0, weighted, 0, 1);
        else if (kind == GR_CENTER_ESTIMATE) st = center_stage(c, s0, nb, sel, 1, weighted, 1, 1);
        else st = center_onepass_ok(c, sel) ? pbc_center_onepass(c, s0, nb, sel, weighted) : pbc_center_stages(c, s0, nb, sel, weighted);
        if (st) return st;
        st = fetch_states(c, nb); if (st) return st;
        std::vector<GrFrameState> res;
        st = center_redo_fallbacks(c, s0, nb, sel, weighted, res); if (st) return st;   // (nothing to do unless the one-pass proof failed somewhere)
        for (uint32_t f = 0; f < nb; ++f) {
            int s = pre[f];
            if (s != GR_OK) c->err = msg[f];
            else s = frame_status(c, res[f]);
            if (s != GR_OK && first_err == GR_OK) { first_err = s; first_msg = c->err; first_idx = c->err_index; }
            if (status_out) status_out[b0 + f] = s;
            if (out) for (int k = 0; k < 3; ++k) out[3 * (size_t)(b0 + f) + k] = (s == GR_OK) ? res[f].com[k] : NAN;
        }
    }
    if (first_err != GR_OK) { c->err = first_msg; c->err_index = first_idx; }
    return first_err;
} catch (...) { return gr_abi_guard(); }
// atoms_center of the WHOLE system about a large contiguous reference group as ONE pass over HBM: the resident pass in its MODE 1 (gr_resident.h).
// `done[f]` = 1: frame f is finished (its state is in c->state_host[f]); 0: the caller runs the two passes on it (the launch was not taken, never
// started, was aborted before the frame, or handed the frame back -- an atom without position or mass, sums that are not finite);
// `torn[f]` = 1: an aborted launch left the frame half-moved (reported as GR_E_HIP, as the RMSD-fit form does).
static int center_resident(gr_ctx *c, uint32_t s0, uint32_t nb, const GrSel &all, const GrSel &csel, int dim_mask, int weighted,
                           const std::vector<int> &pre, std::vector<uint8_t> &done, std::vector<uint8_t> &torn) {
    done.assign(nb, 0); torn.assign(nb, 0);
    if (!c->center_resident || !csel.contiguous || csel.masked || !all.contiguous || all.start != 0 || all.n != c->n) return GR_OK;
    // which groups: the launch costs the same whatever the group -- 4.25 us per 1e6-atom frame in an orthorhombic cell, 4.35-4.4 in others -- while
    // the two passes shrink with it: orthorhombic (the one-float4-per-lane translate) 4.0 / 4.18 / 4.37 for a hundredth / a tenth / a fifth of the
    // system, other cells 4.33 / 4.53 / 4.71 (tools/center_small_groups.py, profiles/r05_center_bench.json): taken from 15 % of the system in
    // orthorhombic cells, from 3 % in others
    {
        bool ortho = true;
        for (uint32_t f = 0; f < nb; ++f) if (pre[f] == GR_OK) { ortho = c->boxes_host[s0 + f].ortho != 0; break; }
        if (c->resident == 1 && (uint64_t)csel.n * 100 < c->n * (ortho ? 15u : 3u)) return GR_OK;
    }
    if (small_ok(c, csel)) return GR_OK;      // (a group the single-wave estimate takes: that kernel adds in another order -- the same bits only against k_center_sums)
    uint32_t streams = 1, gwg = GR_RES_GROUPS;
    const uint32_t wgs = resident_wgs(c, true, nb, all, &streams, &gwg);
    if (!wgs) return GR_OK;
    if (!resident_acquire(c->device)) return GR_OK;
    struct Release { gr_ctx *c; ~Release() { resident_release(c->device); } } release{ c };
    hipStream_t S = c->stream;
    const uint32_t res_stream = wgs * streams;
    const uint32_t n_fin = std::min<uint32_t>(streams > 8 ? GR_RES_MAX_FIN : 8, c->res_max_wgs - res_stream);
    const size_t rec_words = (size_t)nb * ((wgs + GR_RES_REC_PAD - 1u) & ~(uint32_t)(GR_RES_REC_PAD - 1u)) * GR_RES_REC_WORDS;
    if (rec_words > c->res_wgrec_cap) {
        if (c->res_wgrec) (void)hipFree(c->res_wgrec);
        c->res_wgrec = nullptr; c->res_wgrec_cap = 0;
        HIPCHK(c, hipMalloc(&c->res_wgrec, rec_words * sizeof(unsigned long long)));
        HIPCHK(c, hipMemsetAsync(c->res_wgrec, 0, rec_words * sizeof(unsigned long long), S));   // tag 0 = no launch
        c->res_wgrec_cap = rec_words;
    }
    GrResCtl ctl;
    memset(&ctl, 0, sizeof ctl);
    ctl.wgrec = c->res_wgrec; ctl.rec = c->res_rec; ctl.abort = c->res_abort; ctl.progress = c->res_progress; ctl.epoch = ++c->res_epoch; ctl.n_stream = res_stream; ctl.n_fin = n_fin;
    ctl.wgs_frame = wgs; ctl.streams = streams; ctl.groups_wg = gwg;
    ctl.team_waves = resident_team_waves(wgs, streams);
    ctl.patience_ticks = (unsigned long long)c->wall_khz * 3000ull; ctl.start_ticks = (unsigned long long)c->wall_khz * 200ull;   // 3 s, 0.2 s
    ctl.test_abort_frame = c->res_test_abort_at; c->res_test_abort_at = 0xFFFFFFFFu;
    // (the waves run free unless GR_TUNE_RESIDENT_METRO_NS names a period: the controller of the RMSD-fit form is not applied here)
    ctl.metro_t16 = c->res_metro_ns >= 100 ? (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (uint64_t)c->res_metro_ns * (uint64_t)c->wall_khz * 16ull / 1000000ull) : 0u;
    ctl.metro_lead = (uint32_t)((uint64_t)c->wall_khz * 2000ull / 1000000ull);   // 2 us
    ctl.cen_weighted = weighted ? 1u : 0u; ctl.cen_dim_mask = (uint32_t)dim_mask;
    float *frames = c->frames; size_t stride = c->frame_stride; uint32_t slot0 = s0, nfr = nb, natoms = (uint32_t)c->n;
    const float *masses = c->masses; GrSel sel_arg = csel; const GrBox *boxes = c->boxes_dev; GrPlanDev plan; memset(&plan, 0, sizeof plan);
    GrFrameState *states = c->state_dev; double *fparts = nullptr;
    void *args[] = { &frames, &stride, &slot0, &nfr, &natoms, &masses, &sel_arg, &boxes, &plan, &states, &fparts, &ctl };
    bool ubox = true;
    for (uint32_t f = 1; f < nb && ubox; ++f) ubox = memcmp(&c->boxes_host[s0 + f], &c->boxes_host[s0], sizeof(GrBox)) == 0;
    const bool fit_last = c->res_fit_last ? c->res_fit_last == 2 : (uint64_t)res_stream * 10u >= (uint64_t)c->res_max_wgs * 9u;
    k_res_prepare<<<dim3((res_stream * 8 + 255) / 256), dim3(256), 0, S>>>(c->res_abort + 1, c->res_progress, res_stream * 8);
    HIPCHK(c, hipGetLastError());
    if (c->res_test_no_start) { const uint32_t two = 2u; c->res_test_no_start = 0; HIPCHK(c, hipMemcpyAsync(c->res_abort + 2, &two, sizeof two, hipMemcpyHostToDevice, S)); HIPCHK(c, hipStreamSynchronize(S)); }
    if (hipLaunchKernel(resident_center_fn(ubox, fit_last), dim3(res_stream + n_fin), dim3(GrResShape::LANES), args, GrResShape::LDS_BYTES, S) != hipSuccess) {
        (void)hipGetLastError();          // nothing ran: the two passes take the batch
        c->res_max_wgs = 0;
        return GR_OK;
    }
    HIPCHK(c, hipMemcpyAsync(c->res_words_host, c->res_abort, 12 * sizeof(uint32_t), hipMemcpyDeviceToHost, S));
    HIPCHK(c, hipMemcpyAsync(c->state_host, c->state_dev, nb * sizeof(GrFrameState), hipMemcpyDeviceToHost, S));
    HIPCHK(c, hipStreamSynchronize(S));
    c->res_last_streams = streams;
    if (c->res_words_host[2] != 1u) {      // the launch never started (the device is shared): nothing was touched; sit out the next batches
        c->res_handshake_misses++;
        c->res_backoff = c->res_backoff ? std::min<uint32_t>(c->res_backoff * 2u, 1024u) : 4u;
        c->res_skip = c->res_backoff;
        return GR_OK;
    }
    c->res_backoff = 0;
    c->cen_res_launches++;
#ifdef GR_EXP_STEPTIME
    if (getenv("GR_STEPTIME")) {
        unsigned long long st[32];
        if (hipMemcpy(st, c->res_abort + 16, sizeof st, hipMemcpyDeviceToHost) == hipSuccess) {
            const double turns_ = (double)((nb + streams - 1) / streams + GrResShape::K);
            const char *who[4] = { "wg 0 wave 0", "wg 0 wave 5", "wg mid wave 0", "wg mid wave 5" };
            for (int w = 0; w < 4; ++w) {
                fprintf(stderr, "center steptime %-14s ticks/turn:", who[w]);
                double tot = 0; for (int k = 0; k < 7; ++k) tot += (double)st[w * 8 + k];
                const char *nm[7] = { "gate", "request", "rows+sums", "reduce+handover", "record wait", "fit+stores", "park" };
                for (int k = 0; k < 7; ++k) fprintf(stderr, " %s %.0f", nm[k], (double)st[w * 8 + k] / turns_);
                fprintf(stderr, " | total %.0f | polled %.0f %% of fits\n", tot / turns_, 100.0 * (double)st[w * 8 + 7] / turns_);
            }
        }
    }
#endif
    std::vector<uint32_t> lo(streams, nb), hi(streams, nb);   // per stream: turns every wave has been through / some wave has
    if (c->res_words_host[0]) {
        // aborted (see segment_end): frames below the smallest count of their stream are complete, frames at or above the largest are untouched
        (void)hipMemset(c->res_abort, 0, sizeof(uint32_t));
        c->res_aborts++;
        std::vector<uint32_t> prog((size_t)res_stream * 8);
        HIPCHK(c, hipMemcpy(prog.data(), c->res_progress, prog.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t per = wgs * 8u;
        for (uint32_t s = 0; s < streams; ++s) {
            lo[s] = nb; hi[s] = 0u;
            for (uint32_t k = 0; k < per; ++k) {
                const uint32_t v = prog[(size_t)s * per + k];
                if (v == GR_RES_IDLE_WAVE) continue;
                lo[s] = std::min(lo[s], v); hi[s] = std::max(hi[s], v);
            }
        }
    }
    for (uint32_t f = 0; f < nb; ++f) {
        const int st = c->state_host[f].status;
        const uint32_t s = f % streams, turn = f / streams;
        if (pre[f] != GR_OK) { done[f] = 1; continue; }                       // (failed before the launch: nobody touched it)
        if (st == GR_ST_FALLBACK || st == GR_ST_ABORTED || turn >= hi[s]) { c->cen_res_redone++; continue; }
        if (turn >= lo[s]) { torn[f] = 1; done[f] = 1; continue; }
        done[f] = 1;
    }
    return GR_OK;
}
// translate / wrap / centre a batch of frames: pre[] carries the host checks, frames that failed are left untouched
static int translate_batch(gr_ctx *c, uint32_t s0, uint32_t nb, const Group *g, const float *v, int mode, int dim_mask,
                           const std::vector<int> &pre, const std::vector<std::string> &msg, int *status_out, int &first_err, std::string &first_msg, uint64_t &first_idx) {
    const GrSel sel = make_sel(*g);
    HIPCHK(c, hipMemsetAsync(c->bad_dev, 0xFF, 4 * (size_t)nb * sizeof(uint32_t), c->stream));
    // orthorhombic cells (every frame of the batch that will be touched): one float4 per lane, a 256-atom tile per workgroup (k_translate_wrap_rows)
    bool rows = g->n != 0 && sel.contiguous && c->translate_rows;
    for (uint32_t f = 0; f < nb && rows; ++f) rows = pre[f] != GR_OK || c->boxes_host[s0 + f].ortho != 0;
    if (rows) {
        const uint32_t tiles = ((sel.start + sel.n - 1u) >> 8) - (sel.start >> 8) + 1u;
        k_translate_wrap_rows<<<dim3(tiles * 3u, nb), dim3(64), 0, c->stream>>>(c->frames + (size_t)s0 * c->frame_stride, c->frame_stride, sel, c->boxes_dev + s0, c->state_dev, mode, dim_mask,
                                                                            v ? v[0] : 0.f, v ? v[1] : 0.f, v ? v[2] : 0.f, c->bad_dev);
        HIPCHK(c, hipGetLastError());
    } else if (g->n) {
        const uint64_t units = sel.contiguous ? ((uint64_t)sel.n + 3) / 4 + 128 : (sel.masked & 2u) ? ((uint64_t)sel.span + 3) / 4 + 128 : sel.n;   // (4-atom groups of the block / of a masked selection's span; list entries)
        const uint32_t nwg = (uint32_t)std::min<uint64_t>((units + GR_WG - 1) / GR_WG, 4096);
        k_translate_wrap<<<dim3(nwg, nb), dim3(GR_WG), stream_lds(c, GR_STREAM_WGS_CU_DEFAULT), c->stream>>>(c->frames + (size_t)s0 * c->frame_stride, c->frame_stride, sel, c->boxes_dev + s0, c->state_dev, mode, dim_mask,
                                                                       v ? v[0] : 0.f, v ? v[1] : 0.f, v ? v[2] : 0.f, c->bad_dev);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(c->bad_host, c->bad_dev, 4 * (size_t)nb * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->state_host, c->state_dev, nb * sizeof(GrFrameState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t f = 0; f < nb; ++f) {
        int s = pre[f];
        if (s != GR_OK) c->err = msg[f];
        else if (mode == 1 && c->state_host[f].status != GR_OK) s = frame_status(c, c->state_host[f]);
        else if (c->bad_host[4 * f] != GR_NOIDX) s = fail(c, GR_E_NO_POSITION, "atom has no position", c->bad_host[4 * f]);
        if (s != GR_OK && first_err == GR_OK) { first_err = s; first_msg = c->err; first_idx = c->err_index; }
        if (status_out) status_out[f] = s;
    }
    return GR_OK;
}
static int translate_batch_api(gr_ctx *c, uint32_t first_slot, uint32_t n_frames, const char *group, const float *v, const char *center_group, int dim, int weighted, int *status_out) {
    int st = slot_check(c, first_slot, n_frames); if (st) return st;
    (void)hipSetDevice(c->device);
    const Group *g = find_group(c, group ? group : "all");
    if (!g) return fail(c, GR_E_GROUP_NOT_FOUND, group);
    const Group *cg = nullptr;
    if (center_group) {
        cg = find_group(c, center_group);
        if (!cg) return fail(c, GR_E_GROUP_NOT_FOUND, center_group);
        if (cg->n == 0) return fail(c, GR_E_EMPTY_GROUP, center_group);
        if (dim < 0 || dim > 7) return fail(c, GR_E_INVALID_ARG, "bad dimension");
    }
    static const int mask[8] = { 0, 1, 2, 4, 3, 5, 6, 7 };
    int first_err = GR_OK; std::string first_msg; uint64_t first_idx = 0;
    for (uint32_t b0 = 0; b0 < n_frames; b0 += GR_MAX_BATCH) {
        const uint32_t nb = std::min<uint32_t>(GR_MAX_BATCH, n_frames - b0), s0 = first_slot + b0;
        std::vector<int> pre; std::vector<std::string> msg;
        batch_prechecks(c, s0, nb, true, pre, msg);
        SlotUse use(c, s0, nb);
        st = states_from_prechecks(c, nb, pre); if (st) return st;
        // atoms_center of the whole system about a large group: one pass over HBM where the resident pass takes it (center_resident); the frames
        // it did not finish -- all of them when it was not taken -- go through the two passes below, one run of consecutive frames at a time
        std::vector<uint8_t> done(nb, 0), torn(nb, 0);
        if (cg) { st = center_resident(c, s0, nb, make_sel(*g), make_sel(*cg), mask[dim], weighted, pre, done, torn); if (st) return st; }
        bool any_done = false;
        for (uint32_t f = 0; f < nb; ++f) any_done = any_done || done[f];
        if (any_done) {
            for (uint32_t f = 0; f < nb; ++f) {
                if (!done[f]) continue;
                int sf = pre[f];
                if (sf != GR_OK) c->err = msg[f];
                else if (torn[f]) sf = fail(c, GR_E_HIP, "the resident atoms_center pass stalled while this frame was being moved: some of its atoms carry the new coordinates, others the original ones (frame index in gr_last_error_index)", b0 + f);
                else if (c->state_host[f].status != GR_OK) sf = frame_status(c, c->state_host[f]);
                if (sf != GR_OK && first_err == GR_OK) { first_err = sf; first_msg = c->err; first_idx = c->err_index; }
                if (status_out) status_out[b0 + f] = sf;
            }
            for (uint32_t a = 0; a < nb; ) {
                if (done[a]) { ++a; continue; }
                uint32_t b = a;
                while (b < nb && !done[b]) ++b;
                const std::vector<int> pre_run(pre.begin() + a, pre.begin() + b);
                const std::vector<std::string> msg_run(msg.begin() + a, msg.begin() + b);
                st = states_from_prechecks(c, b - a, pre_run); if (st) return st;
                st = center_stage(c, s0 + a, b - a, make_sel(*cg), 1, weighted, 1, 0); if (st) return st;
                st = translate_batch(c, s0 + a, b - a, g, v, 1, mask[dim], pre_run, msg_run, status_out ? status_out + b0 + a : nullptr, first_err, first_msg, first_idx);
                if (st) return st;
                a = b;
            }
            continue;
        }
        if (cg) { st = center_stage(c, s0, nb, make_sel(*cg), 1, weighted, 1, 0); if (st) return st; }   // group_estimate_center / _com per frame
        st = translate_batch(c, s0, nb, g, v, cg ? 1 : 2, cg ? mask[dim] : 7, pre, msg, status_out ? status_out + b0 : nullptr, first_err, first_msg, first_idx);
        if (st) return st;
    }
    if (first_err != GR_OK) { c->err = first_msg; c->err_index = first_idx; }
    return first_err;
}
int gr_group_translate_batch(gr_ctx *c, uint32_t first_slot, uint32_t n_frames, const char *group, const float v[3], int *status_out) try {
    if (!c) return GR_E_INVALID_ARG;
    if (!v) return fail(c, GR_E_INVALID_ARG, "vector is NULL");
    return translate_batch_api(c, first_slot, n_frames, group, v, nullptr, 7, 0, status_out);
} catch (...) { return gr_abi_guard(); }
int gr_group_wrap_batch(gr_ctx *c, uint32_t first_slot, uint32_t n_frames, const char *group, int *status_out) try {
    if (!c) return GR_E_INVALID_ARG;
    const float z[3] = { 0.f, 0.f, 0.f };
    return translate_batch_api(c, first_slot, n_frames, group, z, nullptr, 7, 0, status_out);
} catch (...) { return gr_abi_guard(); }
int gr_atoms_center_batch(gr_ctx *c, uint32_t first_slot, uint32_t n_frames, const char *ref_group, int dim, int weighted, int *status_out) try {
    if (!c) return GR_E_INVALID_ARG;
    if (!ref_group) return fail(c, GR_E_GROUP_NOT_FOUND, "(null)");
    return translate_batch_api(c, first_slot, n_frames, "all", nullptr, ref_group, dim, weighted, status_out);
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ RMSD */
void gr_rmsd_plan_destroy(gr_rmsd_plan *p) try {
    if (!p) return;
    if (p->target) (void)hipSetDevice(p->target->device);
    if (p->target && (p->target->in_flight == p || p->pend.active)) { (void)hipStreamSynchronize(p->target->stream); if (p->target->in_flight == p) p->target->in_flight = nullptr; if (p->pend.resident) resident_done(p->target); }
    // a one-frame fit of a small selection returns as soon as its single-wave kernel has flagged the result; the fit kernel queued behind it reads
    // the plan's reference rows and weights: they are freed only when the stream has run dry (hipFree happens to synchronise the device
    // today; a pooled or asynchronous free would not)
    if (p->target && p->fit_behind_small) (void)hipStreamSynchronize(p->target->stream);
    if (p->p_dev) (void)hipFree(p->p_dev);
    if (p->w_dev) (void)hipFree(p->w_dev);
    if (p->p_span_dev) (void)hipFree(p->p_span_dev);
    delete p;
} catch (...) { }

gr_rmsd_plan *gr_rmsd_plan_create(gr_ctx *ref, uint32_t ref_slot, gr_ctx *target, const char *group, int *status) try {
    int dummy; if (!status) status = &dummy;
    if (!ref || !target || !group || ref_slot >= ref->n_slots) { *status = GR_E_INVALID_ARG; return nullptr; }
    if (ref->device != target->device) { *status = fail(ref, GR_E_INVALID_ARG, "reference and target live on different devices"); return nullptr; }
    (void)hipSetDevice(ref->device);
    int st = busy_check(ref);
    if (st) { *status = st; return nullptr; }
    // extract_data_from_system(reference): box first (rmsd.rs:430), then group_get_com (:433)
    st = box_check(ref, ref_slot);
    if (st) { *status = st; return nullptr; }
    const Group *g = find_group(ref, group);
    if (!g) { *status = fail(ref, GR_E_GROUP_NOT_FOUND, group); return nullptr; }
    if (g->n == 0) { *status = fail(ref, GR_E_EMPTY_GROUP, group); return nullptr; }
    const GrSel sel = make_sel(*g);
    SlotUse use(ref, ref_slot);
    st = state_reset(ref, 1);
    if (!st) st = pbc_center_stages(ref, ref_slot, 1, sel, 1);
    if (!st) st = fetch_states(ref, 1);
    if (!st) st = frame_status(ref, ref->state_host[0]);
    if (st) { *status = st; return nullptr; }
    gr_rmsd_plan *p = new gr_rmsd_plan();
    p->target = target; p->group = group; p->n_ref = g->n;
    const uint32_t pofs = sel.contiguous ? (sel.start & 255u) : 0u;   // tiles of p line up with the frame's tiles
    const size_t s_pad = ((size_t)g->n + pofs + 511) & ~(size_t)255;
    bool ok = hipMalloc(&p->p_dev, s_pad * 3 * sizeof(float)) == hipSuccess && hipMalloc(&p->w_dev, s_pad * sizeof(float)) == hipSuccess;
    if (ok) ok = hipMemsetAsync(p->p_dev, 0, s_pad * 3 * sizeof(float), ref->stream) == hipSuccess &&
                 hipMemsetAsync(p->w_dev, 0, s_pad * sizeof(float), ref->stream) == hipSuccess;
    if (!ok) { *status = fail(ref, GR_E_HIP, "plan allocation failed"); gr_rmsd_plan_destroy(p); return nullptr; }
    if (sel.masked) {   // the by-atom copy for k_sums_pk<.., MASK>: [tile of the first atom .. tile of the last], unselected atoms zero
        const size_t span_pad = (((size_t)(sel.start & 255u) + sel.span + 255) & ~(size_t)255) + 256;
        if (hipMalloc(&p->p_span_dev, span_pad * 3 * sizeof(float)) != hipSuccess || hipMemsetAsync(p->p_span_dev, 0, span_pad * 3 * sizeof(float), ref->stream) != hipSuccess) {
            if (p->p_span_dev) (void)hipFree(p->p_span_dev);
            p->p_span_dev = nullptr;                  // (the plan works without it: the gather paths)
        }
        p->ref_blocks = g->blocks;
    }
    const uint32_t nch = chunks_for(sel);
    k_plan_extract<<<dim3(nch), dim3(GR_WG), 0, ref->stream>>>(ref->frames + (size_t)ref_slot * ref->frame_stride, ref->masses, sel, ref->boxes_dev + ref_slot, ref->state_dev, pofs, p->p_dev, p->w_dev, ref->cen_partials, p->p_span_dev);
    std::vector<GrCenPartial> parts(nch);
    if (hipMemcpyAsync(parts.data(), ref->cen_partials, nch * sizeof(GrCenPartial), hipMemcpyDeviceToHost, ref->stream) != hipSuccess ||
        hipStreamSynchronize(ref->stream) != hipSuccess) {
        *status = fail(ref, GR_E_HIP, "plan extraction failed"); gr_rmsd_plan_destroy(p); return nullptr;
    }
    double s[GR_CEN_K] = { 0 };
    for (uint32_t k = 0; k < nch; ++k) for (int q = 0; q < GR_CEN_K; ++q) s[q] += parts[k].s[q];
    p->dev.p = p->p_dev; p->dev.w = p->w_dev;
    for (int a = 0; a < 3; ++a) { p->dev.sp[a] = s[a]; p->dev.swp[a] = s[3 + a]; p->dev.ref_com[a] = ref->state_host[0].com[a]; }
    p->dev.swpp = s[6]; p->dev.sw = s[7]; p->dev.n = (uint32_t)g->n; p->dev.w_is_mass = 0;
    // reference masses of the group in selection order (weights, rmsd.rs:154-155)
    p->w_host.reserve(g->n);
    for (uint64_t i : grc::expand(g->blocks)) p->w_host.push_back(ref->masses_host[i]);
    *status = GR_OK;
    return p;
} catch (...) { return nullptr; }

uint32_t gr_rmsd_plan_last_fallbacks(const gr_rmsd_plan *p) { return p ? p->last_fallbacks : 0; }
int gr_rmsd_plan_force_exact(gr_rmsd_plan *p, int on) { if (!p) return GR_E_INVALID_ARG; p->exact = on ? 1 : 0; return GR_OK; }
int gr_ctx_set_center_onepass_min(gr_ctx *c, uint32_t min_atoms) { if (!c) return GR_E_INVALID_ARG; c->com_onepass_min = min_atoms; return GR_OK; }
uint64_t gr_center_fallbacks(const gr_ctx *c) { return c ? c->center_fallbacks : 0; }
int gr_ctx_stat(const gr_ctx *c, int key, uint64_t *value) {
    if (!c || !value) return GR_E_INVALID_ARG;
    switch (key) {
    case GR_STAT_N_CUS: *value = c->n_cus; return GR_OK;
    case GR_STAT_RES_MAX_WGS: *value = c->res_max_wgs; return GR_OK;
    case GR_STAT_RES_LAST_STREAMS: *value = c->res_last_streams; return GR_OK;
    case GR_STAT_RES_LAUNCHES: *value = c->res_launches; return GR_OK;
    case GR_STAT_RES_HANDSHAKE_MISSES: *value = c->res_handshake_misses; return GR_OK;
    case GR_STAT_RES_ABORTS: *value = c->res_aborts; return GR_OK;
    case GR_STAT_RES_REDONE_FRAMES: *value = c->res_redone_frames; return GR_OK;
    case GR_STAT_RMSD_FAST_FRAMES: *value = c->rmsd_fast_frames; return GR_OK;
    case GR_STAT_RMSD_EXACT_REDOS: *value = c->rmsd_exact_redos; return GR_OK;
    case GR_STAT_XTC_DEVICE_FRAMES: *value = c->xtc_dev_frames; return GR_OK;
    case GR_STAT_SMALL_CALLS: *value = c->small_calls; return GR_OK;
    case GR_STAT_SMALL_SYNC_FALLBACKS: *value = c->small_sync_fallbacks; return GR_OK;
    case GR_STAT_RES_METRO_PERIOD_NS: *value = c->res_metro_period_ns; return GR_OK;
    case GR_STAT_RES_LAST_TURN_NS: *value = c->res_last_turn_ns; return GR_OK;
    case GR_STAT_RES_LATE_PERMILLE: *value = c->res_late_permille; return GR_OK;
    case GR_STAT_RES_SCLK_MHZ: *value = c->res_sclk_mhz; return GR_OK;
    case GR_STAT_CENTER_RES_LAUNCHES: *value = c->cen_res_launches; return GR_OK;
    case GR_STAT_CENTER_RES_REDONE: *value = c->cen_res_redone; return GR_OK;
    default: return GR_E_INVALID_ARG;
    }
}
int gr_ctx_set_tuning(gr_ctx *c, int key, int64_t value) try {
    if (!c) return GR_E_INVALID_ARG;
    { int st = busy_check(c); if (st) return st; }
    switch (key) {
    case GR_TUNE_SUB_BATCH: if (value < 1 || value > GR_MAX_BATCH) break; c->sub_batch = (uint32_t)value; return GR_OK;
    case GR_TUNE_CHUNKS: if (value < 0 || value > GR_MAX_CHUNKS) break; c->chunks = (uint32_t)value; return GR_OK;
    case GR_TUNE_FIT_WGS: if (value < 0 || value > 65535) break; c->fit_wgs = (uint32_t)value; return GR_OK;
    case GR_TUNE_FUSE: c->fuse = value ? 1 : 0; return GR_OK;
    case GR_TUNE_SMALL_CALLS: if (value < 0 || value > 65536) break; c->small_max = (uint32_t)value; return GR_OK;
    case GR_TUNE_TWO_PASS: c->two_pass = value ? 1 : 0; return GR_OK;
    case GR_TUNE_RESIDENT: if (value < 0 || value > 2) break; c->resident = value; return GR_OK;
    case GR_TUNE_MASKED_SELECTIONS: if (value != 0 && value != 1) break; c->masked_sel = (int)value; return GR_OK;
    case GR_TUNE_XTC_DEVICE_ENCODE: if (value != 0 && value != 1) break; c->xtc_dev_encode = (int)value; return GR_OK;
    case GR_TUNE_RMSD_FAST: if (value != 0 && value != 1) break; c->rmsd_fast = (int)value; return GR_OK;
    case GR_TUNE_RMSD_FAST_MIN: if (value < 0 || value > 0x7fffffff) break; c->rmsd_fast_min = (uint32_t)value; return GR_OK;
    case GR_TUNE_RMSD_FAST_SIGMAS: if (value < 0 || value > 1000) break; c->rmsd_fast_sigmas = (int)value; return GR_OK;
    case GR_TUNE_PAIRDIST_SYMMETRIC: if (value != 0 && value != 1) break; c->pd_sym = (int)value; return GR_OK;
    case GR_TUNE_RESIDENT_WG_GROUPS: if (value != 0 && (value < 64 || value > GR_RES_GROUPS || value % 64 != 0)) break; c->res_wg_groups = (int)value; return GR_OK;
    case GR_TUNE_RESIDENT_STREAMS: if (value < 0 || value > GR_RES_MAX_STREAMS) break; c->res_streams = (int)value; return GR_OK;
    case GR_TUNE_RESIDENT_METRO_NS: if (value < 0 || value > 1000000 || (value > 1 && value < 100)) break; c->res_metro_ns = (int)value; c->metro = gr_ctx::Metro(); return GR_OK;
    case GR_TUNE_RESIDENT_FIT_LAST: if (value < 0 || value > 2) break; c->res_fit_last = (int)value; return GR_OK;
    case GR_TUNE_STREAM_WGS_PER_CU: if (value < 0 || value > 8) break; c->stream_wgs_cu = (int)value; return GR_OK;
    case GR_TUNE_CENTER_RESIDENT: if (value < 0 || value > 1) break; c->center_resident = (int)value; return GR_OK;
    case GR_TUNE_TRANSLATE_ROWS: if (value < 0 || value > 1) break; c->translate_rows = (int)value; return GR_OK;
    case GR_TUNE_RESIDENT_FILL: if (value < 1 || value > 16) break; c->res_fill16 = (int)value; return GR_OK;
    case GR_TUNE_RESIDENT_GROUPS: if (value != 2) break; return GR_OK;   // (the one-group shape was removed: gr_resident.h)
    case GR_TUNE_TEST_RESIDENT_NO_START: c->res_test_no_start = value ? 1 : 0; return GR_OK;
    case GR_TUNE_TEST_RESIDENT_ABORT_AT: c->res_test_abort_at = value < 0 ? 0xFFFFFFFFu : (uint32_t)value; return GR_OK;
    default: break;
    }
    return fail(c, GR_E_INVALID_ARG, "unknown tuning key or value out of range");
} catch (...) { return gr_abi_guard(); }

// multi-pass exact path for `nf` frames starting at first_slot; states [0, nf) must be reset by the caller
static int rmsd_exact(gr_rmsd_plan *p, gr_ctx *c, const GrSel &sel, uint32_t first_slot, uint32_t nf, int fit) {
    int st = pbc_center_stages(c, first_slot, nf, sel, 1); if (st) return st;
    if (small_ok(c, sel)) {   // a small selection: sums and closing step of every frame by one wave (gr_small.h), as its centre stages above
        k_rmsd_small<1><<<dim3(nf), dim3(64), 0, c->stream>>>(c->frames, c->frame_stride, first_slot, c->masses, sel, c->boxes_dev, p->dev, c->state_dev, 1, nullptr, nullptr, 0u);
    } else {
        const uint32_t nch = chunks_for(sel);
        k_rmsd_accum<1><<<dim3(nch, nf), dim3(GR_WG), 0, c->stream>>>(c->frames, c->frame_stride, first_slot, c->masses, sel, c->boxes_dev, p->dev, c->state_dev, c->acc_partials);
        k_rmsd_finalize<1><<<dim3(nf), dim3(GR_WG), 0, c->stream>>>(c->acc_partials, nch, c->frames, c->frame_stride, first_slot, sel, c->boxes_dev, p->dev, c->state_dev);
    }
    if (fit) {
        const uint32_t gx = (uint32_t)std::min<uint64_t>(((c->n >> 2) + GR_WG * 4 - 1) / (GR_WG * 4) + 1, 1024);
        k_fit_pk<false><<<dim3(gx, nf), dim3(GR_WG), 0, c->stream>>>(c->frames, c->frame_stride, first_slot, (uint32_t)c->n, c->boxes_dev, p->dev, c->state_dev, c->masses, sel, nullptr);
    }
    HIPCHK(c, hipGetLastError());
    return GR_OK;
}

// ---- one segment (<= GR_MAX_BATCH frames) in two halves, so a caller can overlap the GPU work of one segment with
// uploads (gr_frame_upload on the copy stream) and host work for the next:
//   segment_begin : host checks in the reference's order, state upload, all launches, async state read-back
//   segment_end   : wait, per-kernel event read-out, exact-path redo of frames whose image proof failed, results
static int segment_begin(gr_rmsd_plan *p, uint32_t s0, uint32_t nb, int fit) {
    gr_ctx *c = p->target;
    Pending &q = p->pend;
    q = Pending();
    q.s0 = s0; q.nb = nb; q.fit = fit; q.active = true;
    SlotUse use(c, s0, nb);
    const Group *g = find_group(c, p->group.c_str());
    // host-side checks in the reference's order: box (rmsd.rs:430) -> group exists -> non-empty
    q.pre.assign(nb, GR_OK); q.pre_idx.assign(nb, 0); q.pre_msg.assign(nb, std::string());
    for (uint32_t f = 0; f < nb; ++f) {
        int s = box_check(c, s0 + f);
        if (s == GR_OK && !g) s = fail(c, GR_E_GROUP_NOT_FOUND, p->group);
        if (s == GR_OK && g->n == 0) s = fail(c, GR_E_EMPTY_GROUP, p->group);
        q.pre[f] = s;
        if (s != GR_OK) { q.pre_msg[f] = c->err; q.pre_idx[f] = c->err_index; }
        q.any_ok = q.any_ok || s == GR_OK;
    }
    if (!q.any_ok) return GR_OK;
    const GrSel sel = make_sel(*g);
    q.sel = sel; q.group_n = g->n; q.has_group = true;
    if (!p->resolved || p->resolved_epoch != c->epoch) {   // weights identical to the target's masses of the group -> one load serves both
        p->resolved_epoch = c->epoch;
        bool same = (g->n == p->n_ref);
        if (same) { size_t k = 0; for (uint64_t i : grc::expand(g->blocks)) { const float a = c->masses_host[i], b = p->w_host[k++]; if (!(a == b)) { same = false; break; } } }
        p->dev.w_is_mass = same ? 1u : 0u; p->resolved = true;
        p->span_ok = p->p_span_dev != nullptr && g->masked && g->mask_dev != nullptr && g->blocks == p->ref_blocks;   // the by-atom copy fits THIS group
    }
    q.consistent = (g->n == p->n_ref);
    // (a selection the caller has sent to the f32-chain RMSD pass -- GR_TUNE_RMSD_FAST_MIN lowered below GR_TUNE_SMALL_CALLS -- takes that pass)
    const bool fast_asked = c->two_pass && c->rmsd_fast && p->dev.w_is_mass != 0 && g->n >= c->rmsd_fast_min && ((!fit && sel.contiguous) || (sel.masked && p->span_ok));
    const bool small = q.consistent && !p->exact && small_ok(c, sel) && !fast_asked;
    if (small && nb == 1) {
        // ONE frame, RMSD (with or without fit) of a small selection -- the reference's per-frame calc_rmsd / calc_rmsd_and_fit on a
        // protein: one single-wave dispatch that accumulates, closes the frame (closed-form rmsd from exact products, rotation, shift) and
        // leaves its state in host-mapped memory (gr_small.h); a fit follows it on the stream as the plain transform of all atoms, and the
        // host does not wait for that one (the rmsd is known; whatever touches the slot next is ordered behind it).  Everything after
        // the wait -- the redo of a frame whose image proof failed included -- is segment_end's usual way.
        q.small = true; q.small_seq = ++c->small_seq;
        k_rmsd_small<0><<<dim3(1), dim3(64), 0, c->stream>>>(c->frames, c->frame_stride, s0, c->masses, sel, c->boxes_dev, p->dev,
                                                         c->state_dev, 0, c->small_state_dev, c->small_flag_dev, q.small_seq);
        if (fit) k_fit_pk<false><<<dim3(fit_grid(c, 1), 1), dim3(GR_WG), 0, c->stream>>>(c->frames, c->frame_stride, s0, (uint32_t)c->n, c->boxes_dev, p->dev, c->state_dev, c->masses, sel, nullptr);
        p->fit_behind_small = fit != 0;       // (the host returns when the small kernel's flag is seen: this fit may still be reading plan->p_dev / w_dev -- see gr_rmsd_plan_destroy)
        HIPCHK(c, hipGetLastError());
        return GR_OK;
    }
    {
        // the frames' states start from the host-side checks.  When every frame passed them -- the usual case -- the states are zeroed by a
        // kernel: the copy of nb records out of pinned memory costs ~30 us of host time BEFORE the first kernel of the segment can be
        // queued (rocprofv3 --hip-trace of bench.py: hipMemcpyAsync 29 us per call), i.e. with the device idle
        bool all_ok = true;
        for (uint32_t f = 0; f < nb; ++f) all_ok = all_ok && q.pre[f] == GR_OK;
        if (all_ok) { const int sr = state_reset(c, nb); if (sr) return sr; }
        else {
            for (uint32_t f = 0; f < nb; ++f) { GrFrameState z = {}; z.err_index = GR_NOIDX; z.status = q.pre[f]; c->state_host[f] = z; }
            HIPCHK(c, hipMemcpyAsync(c->state_dev, c->state_host, nb * sizeof(GrFrameState), hipMemcpyHostToDevice, c->stream));
        }
    }
    int st;
    if (small) {
        // ... and a batch of frames of such a selection: the same kernel, one wave per frame (a batch equals its per-frame calls bit for bit)
        k_rmsd_small<0><<<dim3(nb), dim3(64), 0, c->stream>>>(c->frames, c->frame_stride, s0, c->masses, sel, c->boxes_dev, p->dev, c->state_dev, 1, nullptr, nullptr, 0u);
        if (fit) k_fit_pk<false><<<dim3(fit_grid(c, nb), nb), dim3(GR_WG), 0, c->stream>>>(c->frames, c->frame_stride, s0, (uint32_t)c->n, c->boxes_dev, p->dev, c->state_dev, c->masses, sel, nullptr);
        HIPCHK(c, hipGetLastError());
    } else if (!q.consistent) {
        // positions and masses of the target are still checked first (extract_data_from_system runs to
        // completion before number_of_positions_consistent, rmsd.rs:206-214)
        st = pbc_center_stages(c, s0, nb, sel, 1); if (st) return st;
    } else if (p->exact) {
        st = rmsd_exact(p, c, sel, s0, nb, fit); if (st) return st;
    } else {
        // groups of sub_batch frames: sums (closing its frames on its own tail) -> fit -> rmsd close back to back on the stream,
        // no host round trip in between; one state fetch for the whole segment afterwards.  (Orders that put the fit of group g
        // beside the sums of group g + 1 on a second stream, or the small kernels on a high-priority stream, were measured
        // slower in round 1 and are gone: DESIGN.md "Batching".)
        const uint32_t sb = c->sub_batch;
        // RMSD-fit of a contiguous selection: the sums pass only steers R and the centre, the fit pass evaluates
        // sum w |R q - p|^2 on the way (k_fit_pk<true>) and k_rmsd_close turns it into the rmsd
        const bool lite = fit && sel.contiguous && c->two_pass;
        // RMSD WITHOUT fit of a contiguous mass-weighted selection: the sums pass of the fit path + the closed-form RMSD's sums as f32
        // chains widened to fp64 (k_sums_pk<false, true>); frames whose rmsd comes out too close to the rounding of its own sums
        // are handed back (GR_ST_REDO_EXACT) and redone below by the exact-product pass that every other selection takes
        // ... and, with or without fit, of a DENSE scattered one (GrSel::masked, the plan's by-atom copy of the reference fitting this
        // group): the same pass over the selection's span with its bit mask, then -- for a fit -- the plain transform of every atom
        const bool msk = sel.masked && p->span_ok;
        const bool fast = c->two_pass && c->rmsd_fast && p->dev.w_is_mass != 0 && g->n >= c->rmsd_fast_min && ((!fit && sel.contiguous) || msk);
        q.rmsd_fast = fast;
        p->dev.fast_sigmas = (float)c->rmsd_fast_sigmas;
        GrPlanDev plan_span = p->dev; plan_span.p = p->p_span_dev;      // (after every field of p->dev this call sets)
        const uint32_t n_groups = (nb + sb - 1) / sb;
        if (lite) {
            size_t need = 0;
            for (uint32_t f0 = 0; f0 < nb; f0 += sb) need = std::max(need, (size_t)nb * fit_grid(c, std::min<uint32_t>(sb, nb - f0)) * (GR_WG / 64));   // (one word per wave)
            if (need > c->fit_partials_cap) {
                if (c->fit_partials) (void)hipFree(c->fit_partials);
                c->fit_partials = nullptr; c->fit_partials_cap = 0;
                HIPCHK(c, hipMalloc(&c->fit_partials, need * sizeof(double)));
                c->fit_partials_cap = need;
            }
        }
        const bool fused = (lite || fast) && c->fuse;   // the finalize rides on the tail of the sums kernel
        q.fused = fused;
        hipStream_t S = c->stream;
        uint32_t res_streams = 1, res_gwg = GR_RES_GROUPS;
        // (the pass also takes the RMSD-fit of a masked selection weighted by its masses: the lanes' membership flags carry the bits)
        const bool res_msk = fit && msk && c->two_pass && p->dev.w_is_mass != 0;
        uint32_t res_wgs = resident_wgs(c, lite || res_msk, nb, sel, &res_streams, &res_gwg);   // workgroups per frame x frame streams
        if (res_wgs && !resident_acquire(c->device)) res_wgs = 0;
        if (res_wgs) c->res_in_use = true;           // (released in segment_end, or by the caller when this function fails)
        if (res_wgs) {
            // ONE launch for the segment: every frame is read once and written once (gr_resident.h)
            const uint32_t res_stream = res_wgs * res_streams;
            const uint32_t n_fin = std::min<uint32_t>(res_streams > 8 ? GR_RES_MAX_FIN : 8, c->res_max_wgs - res_stream);
            if ((size_t)nb * res_wgs * GrResShape::WAVES > c->fit_partials_cap) {      // (one word per streaming WAVE and frame)
                if (c->fit_partials) (void)hipFree(c->fit_partials);
                c->fit_partials = nullptr; c->fit_partials_cap = 0;
                HIPCHK(c, hipMalloc(&c->fit_partials, (size_t)nb * res_wgs * GrResShape::WAVES * sizeof(double)));
                c->fit_partials_cap = (size_t)nb * res_wgs * GrResShape::WAVES;
            }
            const size_t rec_words = (size_t)nb * ((res_wgs + GR_RES_REC_PAD - 1u) & ~(uint32_t)(GR_RES_REC_PAD - 1u)) * GR_RES_REC_WORDS;
            if (rec_words > c->res_wgrec_cap) {
                if (c->res_wgrec) (void)hipFree(c->res_wgrec);
                c->res_wgrec = nullptr; c->res_wgrec_cap = 0;
                HIPCHK(c, hipMalloc(&c->res_wgrec, rec_words * sizeof(unsigned long long)));
                HIPCHK(c, hipMemsetAsync(c->res_wgrec, 0, rec_words * sizeof(unsigned long long), S));   // tag 0 = no launch
                c->res_wgrec_cap = rec_words;
            }
            GrResCtl ctl;
            ctl.wgrec = c->res_wgrec; ctl.rec = c->res_rec; ctl.abort = c->res_abort; ctl.progress = c->res_progress; ctl.epoch = ++c->res_epoch; ctl.n_stream = res_stream; ctl.n_fin = n_fin;
            ctl.wgs_frame = res_wgs; ctl.streams = res_streams; ctl.groups_wg = res_gwg;
            ctl.team_waves = resident_team_waves(res_wgs, res_streams);
            ctl.patience_ticks = (unsigned long long)c->wall_khz * 3000ull; ctl.start_ticks = (unsigned long long)c->wall_khz * 200ull;   // 3 s, 0.2 s
            ctl.test_abort_frame = c->res_test_abort_at; c->res_test_abort_at = 0xFFFFFFFFu;
            {   // the metronome's period for this launch
                const uint64_t shape = ((uint64_t)res_wgs << 40) ^ ((uint64_t)res_streams << 32) ^ ((uint64_t)res_gwg << 16) ^ (uint64_t)(c->n & 0xFFFF) ^ ((uint64_t)(sel.start == 0 && sel.n == c->n) << 63);
                uint32_t period_ns = 0;
                q.res_metro_auto = false;
                if (c->res_metro_ns >= 100) period_ns = (uint32_t)c->res_metro_ns;
                else if (c->res_metro_ns == 0) {
                    gr_ctx::Metro &m = c->metro;
                    if (m.shape != shape) { m = gr_ctx::Metro(); m.shape = shape; }
                    if (m.off_for) m.off_for--;
                    else if (m.free_ns > 0) period_ns = (uint32_t)(m.T_ns + 0.5);
                    q.res_metro_auto = true;
                }
                q.res_metro_ns = period_ns;
                ctl.metro_t16 = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (uint64_t)period_ns * (uint64_t)c->wall_khz * 16ull / 1000000ull);
                ctl.metro_lead = (uint32_t)((uint64_t)c->wall_khz * 2000ull / 1000000ull);   // 2 us
            }
#ifdef GR_EXP_STEPTIME
            {
                static unsigned long long *dbg_dev = nullptr;
                if (!dbg_dev) (void)hipMalloc(&dbg_dev, (size_t)GR_MAX_CHUNKS * 8 * 4 * sizeof(unsigned long long));
                ctl.dbg = getenv("GR_STEPTIME") ? dbg_dev : nullptr;
                g_steptime_dbg = ctl.dbg;
            }
#endif
#ifdef GR_EXP_TIMELINE
            // experiment (tools/timeline_bench.sh): device-clock stamps of every frame's way through the launch
            {
                static unsigned long long *tl_dev = nullptr; static size_t tl_cap = 0;
                if (nb > tl_cap) { if (tl_dev) (void)hipFree(tl_dev); tl_dev = nullptr; tl_cap = 0; if (hipMalloc(&tl_dev, (size_t)nb * 8 * sizeof(unsigned long long)) == hipSuccess) tl_cap = nb; }
                ctl.tl = getenv("GR_TIMELINE") ? tl_dev : nullptr;
                if (ctl.tl) HIPCHK(c, hipMemsetAsync(ctl.tl, 0, (size_t)nb * 8 * sizeof(unsigned long long), S));
                q.tl_dev = ctl.tl;
            }
#endif
            float *frames = c->frames; size_t stride = c->frame_stride; uint32_t slot0 = s0, nfr = nb, natoms = (uint32_t)c->n;
            const float *masses = c->masses; GrSel sel_arg = sel; const GrBox *boxes = c->boxes_dev; GrPlanDev plan = res_msk ? plan_span : p->dev;
            GrFrameState *states = c->state_dev; double *fparts = c->fit_partials;
            void *args[] = { &frames, &stride, &slot0, &nfr, &natoms, &masses, &sel_arg, &boxes, &plan, &states, &fparts, &ctl };
            bool ubox = true;   // the same box in every frame of the segment (constant-volume runs): its constants are loaded once
            for (uint32_t f = 1; f < nb && ubox; ++f) ubox = memcmp(&c->boxes_host[s0 + f], &c->boxes_host[s0], sizeof(GrBox)) == 0;
#ifdef GR_EXP_NO_UBOX
            ubox = false;
#endif
            // the selection is the whole system: the kernel that parks image vectors (gr_resident.h, V)
            const bool whole = sel.start == 0 && sel.n == c->n;
            // sums before fit (gr_resident.h, FL) where the streaming workgroups fill the chip and memory paces the walk: measured (profiles/
            // r05_ab_sizes.txt, frames/s, fit first -> sums first) 1e6 atoms 230 -> 235 k, 3 x 330 k 656 -> 679 k, level at 4 x 250 k and below;
            // one stream of 520-800 k atoms (170-196 of 256 CUs) 261 -> 252 k, 251 -> 247 k, 237 -> 235 k: those keep the fit first
            const bool fit_last = c->res_fit_last ? c->res_fit_last == 2 : (uint64_t)res_stream * 10u >= (uint64_t)c->res_max_wgs * 9u;
            const void *fn = resident_fn(p->dev.w_is_mass != 0, ubox, whole, fit_last);
            const uint32_t lanes = GrResShape::LANES, lds = GrResShape::LDS_BYTES;
            // start handshake (count, verdict) and the waves' progress words back to zero: one small kernel instead of two memsets
            k_res_prepare<<<dim3((res_stream * 8 + 255) / 256), dim3(256), 0, S>>>(c->res_abort + 1, c->res_progress, res_stream * 8);
            HIPCHK(c, hipGetLastError());
            if (c->res_test_no_start) { const uint32_t two = 2u; c->res_test_no_start = 0; HIPCHK(c, hipMemcpyAsync(c->res_abort + 2, &two, sizeof two, hipMemcpyHostToDevice, S)); HIPCHK(c, hipStreamSynchronize(S)); }
            if (c->profile) EVREC(c, c->pev[0], true, S);
            // An ORDINARY launch: the grid fits the device with one workgroup per CU (resident_wgs checked), other kernels that hold
            // CUs when it starts finish on their own, and the guard above keeps a second resident launch of this process away.
            // (hipLaunchCooperativeKernel, which has the runtime check co-residency, was used at first: rocprofv3 --pmc faults on
            // it, and a process that issued it from two host threads crashed inside the runtime at exit -- ROCm 7.2.)
            const hipError_t le = hipLaunchKernel(fn, dim3(res_stream + n_fin), dim3(lanes), args, lds, S);
            if (le == hipSuccess) {
                if (c->profile) EVREC(c, c->pev[1], true, S);
                k_rmsd_close<<<dim3(nb), dim3(64), 0, S>>>(c->fit_partials, res_wgs * GrResShape::WAVES, p->dev.sw, c->state_dev);
                HIPCHK(c, hipGetLastError());
                // (the launch's control words follow it on the stream into pinned memory: segment_end reads them after its one synchronisation
                //  instead of fetching them with a blocking copy of their own -- 15-25 us per call)
                HIPCHK(c, hipMemcpyAsync(c->res_words_host, c->res_abort, 12 * sizeof(uint32_t), hipMemcpyDeviceToHost, S));
                q.resident = true; q.res_stream = res_stream; q.res_streams = res_streams;
                q.rmsd_fast = false;                    // (the pass closes its frames with the fit's own sum: nothing to hand back)
                c->res_last_streams = res_streams;
            } else {
                (void)hipGetLastError();          // nothing ran: the two-pass path takes the segment
                c->res_max_wgs = 0;
                resident_done(c);
            }
        }
        for (uint32_t g = 0; g < n_groups && !q.resident; ++g) {
            const uint32_t f0 = g * sb, nf = std::min<uint32_t>(sb, nb - f0), nch = batch_chunks(c, sel, nf), gx = fit_grid(c, nf);
            GrAccPartial *parts = c->acc_partials + (size_t)f0 * GR_MAX_CHUNKS;
            // each kernel is bracketed by its own pair of profiling events (gr_profile_*)
            if (c->profile) EVREC(c, c->pev[6 * g], true, S);
            if (lite) k_sums_pk<false><<<dim3(nch, nf), dim3(GR_WG), 0, S>>>(c->frames, c->frame_stride, s0 + f0, c->masses, sel, c->boxes_dev, p->dev, parts,
                                                                             fused ? c->fuse_cnt + f0 : nullptr, c->state_dev + f0);
            else if (fast && msk) k_sums_pk<false, true, true><<<dim3(nch, nf), dim3(GR_WG), 0, S>>>(c->frames, c->frame_stride, s0 + f0, c->masses, sel, c->boxes_dev, plan_span, parts,
                                                                                                      fused ? c->fuse_cnt + f0 : nullptr, c->state_dev + f0);
            else if (fast) k_sums_pk<false, true><<<dim3(nch, nf), dim3(GR_WG), 0, S>>>(c->frames, c->frame_stride, s0 + f0, c->masses, sel, c->boxes_dev, p->dev, parts,
                                                                                        fused ? c->fuse_cnt + f0 : nullptr, c->state_dev + f0);
            else k_rmsd_accum<0><<<dim3(nch, nf), dim3(GR_WG), 0, S>>>(c->frames, c->frame_stride, s0 + f0, c->masses, sel, c->boxes_dev, p->dev, c->state_dev + f0, parts);
            if (c->profile) EVREC(c, c->pev[6 * g + 1], true, S);
            if (!fused) {
                if (c->profile) EVREC(c, c->pev[6 * g + 2], true, S);
                if (lite) k_rmsd_finalize_lite<false><<<dim3(nf), dim3(64), 0, S>>>(parts, nch, c->frames, c->frame_stride, s0 + f0, sel, c->boxes_dev, p->dev, c->state_dev + f0);
                else if (fast) k_rmsd_finalize_lite<false, true><<<dim3(nf), dim3(64), 0, S>>>(parts, nch, c->frames, c->frame_stride, s0 + f0, sel, c->boxes_dev, msk ? plan_span : p->dev, c->state_dev + f0);
                else k_rmsd_finalize<0><<<dim3(nf), dim3(GR_WG), 0, S>>>(parts, nch, c->frames, c->frame_stride, s0 + f0, sel, c->boxes_dev, p->dev, c->state_dev + f0);
                if (c->profile) EVREC(c, c->pev[6 * g + 3], true, S);
            }
            if (fit) {
                if (c->profile) EVREC(c, c->pev[6 * g + 4], true, S);
                // (closing the rmsd on the tail of this kernel makes every one of its 62 k workgroups drain its stores: 3x slower)
                if (lite) k_fit_pk<true><<<dim3(gx, nf), dim3(GR_WG), stream_lds(c, GR_STREAM_WGS_CU_DEFAULT), S>>>(c->frames, c->frame_stride, s0 + f0, (uint32_t)c->n, c->boxes_dev, p->dev, c->state_dev + f0, c->masses, sel, c->fit_partials + (size_t)f0 * gx * (GR_WG / 64));
                else k_fit_pk<false><<<dim3(gx, nf), dim3(GR_WG), 0, S>>>(c->frames, c->frame_stride, s0 + f0, (uint32_t)c->n, c->boxes_dev, p->dev, c->state_dev + f0, c->masses, sel, nullptr);
                if (c->profile) EVREC(c, c->pev[6 * g + 5], true, S);
                if (lite) k_rmsd_close<<<dim3(nf), dim3(64), 0, S>>>(c->fit_partials + (size_t)f0 * gx * (GR_WG / 64), gx * (GR_WG / 64), p->dev.sw, c->state_dev + f0);
            }
        }
        HIPCHK(c, hipGetLastError());
        if (c->profile) q.n_prof_groups = q.resident ? 0 : n_groups;
    }
    HIPCHK(c, hipMemcpyAsync(c->state_host, c->state_dev, nb * sizeof(GrFrameState), hipMemcpyDeviceToHost, c->stream));
    return GR_OK;
}

// `final_states` (optional) receives the frames' closing records as the results below are read from them -- after the exact-path
// redo of frames whose image proof failed: what a caller that ran this segment on behalf of another one (the redo of an aborted
// resident launch) needs; c->state_host is scratch that the redo loops overwrite.
static int segment_end(gr_rmsd_plan *p, float *rmsd_out, int *status_out, float *R_out, std::vector<GrFrameState> *final_states = nullptr) {
    gr_ctx *c = p->target;
    Pending &q = p->pend;
    if (!q.active) return fail(c, GR_E_INVALID_ARG, "no batch in flight");
    q.active = false;
    const uint32_t nb = q.nb, s0 = q.s0;
    const int fit = q.fit;
    int first_err = GR_OK; uint64_t first_err_index = 0; std::string first_err_msg; uint64_t first_counts[2] = { 0, 0 };
    auto note = [&](int s) { if (s != GR_OK && first_err == GR_OK) { first_err = s; first_err_index = c->err_index; first_err_msg = c->err; first_counts[0] = c->counts[0]; first_counts[1] = c->counts[1]; } };
    if (!q.any_ok) {
        for (uint32_t f = 0; f < nb; ++f) {
            c->err = q.pre_msg[f]; c->err_index = q.pre_idx[f]; note(q.pre[f]);
            if (status_out) status_out[f] = q.pre[f];
            if (rmsd_out) rmsd_out[f] = NAN;
            if (R_out) for (int k = 0; k < 9; ++k) R_out[9 * (size_t)f + k] = NAN;
        }
        if (final_states) { final_states->assign(nb, GrFrameState{}); for (uint32_t f = 0; f < nb; ++f) { (*final_states)[f].status = q.pre[f]; (*final_states)[f].err_index = GR_NOIDX; } }
    } else {
        if (!q.has_group) return fail(c, GR_E_INVALID_ARG, "batch state lost");
        const GrSel sel = q.sel;
        if (q.small) { const int sw = small_wait(c, q.small_seq); if (sw) return sw; c->state_host[0] = *c->small_state; }
        else HIPCHK(c, hipStreamSynchronize(c->stream));
        std::vector<uint8_t> redo;      // frames of an aborted resident launch that nobody touched: redone on the two-pass path below
        std::vector<uint8_t> torn;      // ... and frames that SOME waves fitted and others did not
#ifdef GR_EXP_TIMELINE
        if (q.resident && q.tl_dev) {
            // per frame (ticks of the 100 MHz device clock): 0 workgroup 0 published its record, 1 the finalizer had all records, 2 before /
            // 3 after the load of the frame's state, 4 closing arithmetic done, 5 record published and state stored, 6 workgroup 0 wave 0
            // looked at the record (| polls << 48), 7 the finalizer's polls
            std::vector<unsigned long long> tl((size_t)nb * 8);
            if (hipMemcpy(tl.data(), q.tl_dev, tl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                double sum[8] = { 0 }; double period = 0; uint64_t n = 0, polled = 0, fpolls = 0; unsigned long long prev = 0;
                for (uint32_t f = 0; f < nb; ++f) {
                    const unsigned long long *t = &tl[(size_t)f * 8];
                    if (!t[0] || !t[1] || !t[5] || !(t[6] & 0xFFFFFFFFFFFFull)) continue;
                    const unsigned long long look = t[6] & 0xFFFFFFFFFFFFull;
                    sum[0] += (double)((long long)(t[1] - t[0])); sum[1] += (double)((long long)(t[2] - t[1])); sum[2] += (double)((long long)(t[3] - t[2]));
                    sum[3] += (double)((long long)(t[4] - t[3])); sum[4] += (double)((long long)(t[5] - t[4])); sum[5] += (double)((long long)(look - t[5]));
                    sum[6] += (double)((long long)(look - t[0]));
                    if (prev) period += (double)((long long)(t[0] - prev));
                    prev = t[0];
                    polled += (t[6] >> 48) != 0; fpolls += t[7]; n++;
                }
                if (n > 1) fprintf(stderr, "timeline (%llu frames, us): pub->all records %.2f | ->team totals %.2f | state load %.2f | closing %.2f | publish+store %.2f | ->wg0 looks %.2f | pub->look %.2f | period %.3f | wg0 polled for %.0f %% of the frames | finalizer polls per frame %.1f\n",
                                   (unsigned long long)n, sum[0] / n / 100.0, sum[1] / n / 100.0, sum[2] / n / 100.0, sum[3] / n / 100.0, sum[4] / n / 100.0, sum[5] / n / 100.0, sum[6] / n / 100.0,
                                   period / (n - 1) / 100.0, 100.0 * polled / n, (double)fpolls / n);
            }
        }
#endif
        if (q.resident) {
            resident_done(c);
            uint32_t words[3] = { 0, 0, 0 };
            words[0] = c->res_words_host[0]; words[1] = c->res_words_host[1]; words[2] = c->res_words_host[2];   // (copied behind the launch: segment_begin)
            if (words[2] != 1u) {
                // the launch never started (its workgroups did not all get onto the chip: the device is shared): no frame was
                // touched -- the segment runs on the two-pass path, and this context sits out the next segments (twice as many after
                // every miss in a row) before it tries the pass again
                c->res_handshake_misses++;
                c->res_backoff = c->res_backoff ? std::min<uint32_t>(c->res_backoff * 2u, 1024u) : 4u;
                c->res_skip = c->res_backoff + 1u;      // (+ 1: the re-run of this segment)
                const int fit_again = q.fit;
                const uint32_t s0_again = q.s0, nb_again = q.nb;
                int st2 = segment_begin(p, s0_again, nb_again, fit_again);
                if (st2) { p->pend.active = false; resident_done(c); return st2; }
                return segment_end(p, rmsd_out, status_out, R_out, final_states);
            }
            c->res_backoff = 0;
            const uint32_t aborted = words[0];
            if (aborted) {
                // A wait inside the launch ran out of patience (or a test asked for it) and the grid drained.  Every streaming wave left
                // word of how many frames it had been through (ctl.progress; waves only ever leave between two frames):
                //   frames below the smallest count are complete (their records, fitted coordinates and rmsd are final);
                //   frames at or above the largest count -- and frames whose finalizer gave up (GR_ST_ABORTED) -- are untouched: redone below;
                //   frames in between were fitted by some waves only (possible when a wave gives up in the instant its record
                //   arrives for the others): their coordinates are lost and they are reported as GR_E_HIP, frame by frame.
                // (Counts are per frame STREAM: frame f is turn f / streams of stream f % streams, whose workgroups are the
                // res_stream / streams consecutive ones from (f % streams) * that many.)
                (void)hipMemset(c->res_abort, 0, sizeof(uint32_t));
                c->res_aborts++;
                std::vector<uint32_t> prog((size_t)q.res_stream * 8);
                HIPCHK(c, hipMemcpy(prog.data(), c->res_progress, prog.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                const uint32_t ns = q.res_streams, per = q.res_stream / ns * 8u;
                std::vector<uint32_t> lo(ns, nb), hi(ns, 0u);
                // (a wave whose chunk lies behind the workgroup's last group fits nothing and says so with GR_RES_IDLE_WAVE: it must not
                // count -- as "all turns done" it used to lift hi[s] to the end of the stream, and a frame that the finalizer had
                // closed but no working wave had reached was then classed as torn instead of untouched)
                for (uint32_t s = 0; s < ns; ++s)
                    for (uint32_t k = 0; k < per; ++k) {
                        const uint32_t v = prog[(size_t)s * per + k];
                        if (v == GR_RES_IDLE_WAVE) continue;
                        lo[s] = std::min(lo[s], v); hi[s] = std::max(hi[s], v);
                    }
                redo.assign(nb, 0); torn.assign(nb, 0);
                for (uint32_t f = 0; f < nb; ++f) {
                    if (q.pre[f] != GR_OK) continue;
                    const bool finalizer_gave_up = c->state_host[f].status == GR_ST_ABORTED;
                    const uint32_t s = f % ns, turn = f / ns;
                    if (turn >= hi[s] || finalizer_gave_up) redo[f] = 1;
                    else if (turn >= lo[s]) torn[f] = 1;
                }
            } else {
                c->res_launches++;
#ifdef GR_EXP_STEPTIME
                if (getenv("GR_STEPTIME") && g_steptime_dbg) {
                    // every streaming wave: share of its walk spent waiting for records, by XCC / by wave number / the 12 waves that waited least
                    std::vector<unsigned long long> d((size_t)q.res_stream * 8 * 4);
                    if (hipMemcpy(d.data(), g_steptime_dbg, d.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                        double by_xcc[8] = { 0 }, by_wave[8] = { 0 }; int n_xcc[8] = { 0 }, n_wave[8] = { 0 };
                        std::vector<std::pair<double, uint32_t>> share;
                        for (uint32_t w = 0; w < q.res_stream * 8; ++w) {
                            if (!d[(size_t)w * 4 + 3]) continue;
                            const double sh = (double)d[(size_t)w * 4] / (double)d[(size_t)w * 4 + 3];
                            const uint32_t x = (uint32_t)(d[(size_t)w * 4 + 2] & 7);
                            by_xcc[x] += sh; n_xcc[x]++; by_wave[w & 7] += sh; n_wave[w & 7]++;
                            share.push_back({ sh, w });
                        }
                        {   // which wave of a workgroup completes the workgroup's record (and so runs the combine): the share of the wave that does it most often
                            double top = 0, top2 = 0; uint32_t nwg = 0; double by_w[8] = { 0 };
                            for (uint32_t g = 0; g < q.res_stream; ++g) {
                                double tot = 0, mx = 0, mx2 = 0;
                                for (uint32_t w = 0; w < 8; ++w) { const double cnt = (double)(d[(size_t)(g * 8 + w) * 4 + 1] >> 32); tot += cnt; by_w[w] += cnt; if (cnt > mx) { mx2 = mx; mx = cnt; } else if (cnt > mx2) mx2 = cnt; }
                                if (tot > 0) { top += mx / tot; top2 += mx2 / tot; nwg++; }
                            }
                            double all = 0; for (int w = 0; w < 8; ++w) all += by_w[w];
                            fprintf(stderr, "steptime combines: the busiest wave of a workgroup does %.1f %% of its combines, the second %.1f %% (even: 12.5); by wave number:", nwg ? 100.0 * top / nwg : 0.0, nwg ? 100.0 * top2 / nwg : 0.0);
                            for (int w = 0; w < 8; ++w) fprintf(stderr, " %.1f", all > 0 ? 100.0 * by_w[w] / all : 0.0);
                            fprintf(stderr, "\n");
                        }
                        std::sort(share.begin(), share.end());
                        fprintf(stderr, "steptime record-wait share by XCC:");
                        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.3f", n_xcc[x] ? by_xcc[x] / n_xcc[x] : 0.0);
                        fprintf(stderr, " | by wave:");
                        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.3f", n_wave[x] ? by_wave[x] / n_wave[x] : 0.0);
                        fprintf(stderr, " | least:");
                        for (size_t k = 0; k < 12 && k < share.size(); ++k) fprintf(stderr, " wg%u.w%u(x%u)=%.3f", share[k].second >> 3, share[k].second & 7, (uint32_t)(d[(size_t)share[k].second * 4 + 2] & 7), share[k].first);
                        fprintf(stderr, " | median %.3f max %.3f\n", share[share.size() / 2].first, share.back().first);
                    }
                }
                if (getenv("GR_STEPTIME")) {
                    unsigned long long st[32];
                    if (hipMemcpy(st, c->res_abort + 16, sizeof st, hipMemcpyDeviceToHost) == hipSuccess) {
                        const double turns_ = (double)((nb + q.res_streams - 1) / q.res_streams + GrResShape::K);
                        const char *who[4] = { "wg 0 wave 0", "wg 0 wave 5", "wg mid wave 0", "wg mid wave 5" };
                        for (int w = 0; w < 4; ++w) {
                            fprintf(stderr, "steptime %-14s ticks/turn:", who[w]);
                            double tot = 0; for (int k = 0; k < 7; ++k) tot += (double)st[w * 8 + k];
                            const char *nm[7] = { "gate", "request", "rows+sums", "reduce+handover", "record wait", "fit+stores", "park" };
                            for (int k = 0; k < 7; ++k) fprintf(stderr, " %s %.0f", nm[k], (double)st[w * 8 + k] / turns_);
                            fprintf(stderr, " | total %.0f | polled %.0f %% of fits\n", tot / turns_, 100.0 * (double)st[w * 8 + 7] / turns_);
                        }
                    }
                }
#endif
                // what the launch says about its own pace (device clock): first slot -> the last streaming wave's exit, over its turns
                // (+ the six turns a frame waits on chip), and how many of its metronome slots were reached late
                unsigned long long t0 = 0, t1 = 0;
                memcpy(&t0, c->res_words_host + 4, 8); memcpy(&t1, c->res_words_host + 6, 8);
                const uint32_t turns = (nb + q.res_streams - 1) / q.res_streams;
                const double turn_ns = (t1 > t0 && turns) ? (double)(t1 - t0) * 1.0e6 / (double)c->wall_khz / (double)(turns + GrResShape::K) : 0.0;
                const uint64_t slots = (uint64_t)q.res_stream * GrResShape::WAVES * turns;
                const uint64_t late_pm = slots ? (uint64_t)c->res_words_host[8] * 1000ull / slots : 0;
                c->res_metro_period_ns = q.res_metro_ns; c->res_last_turn_ns = (uint64_t)(turn_ns + 0.5); c->res_late_permille = late_pm;
                c->res_sclk_mhz = c->res_words_host[11] ? (uint64_t)((double)c->res_words_host[10] / (double)c->res_words_host[11] * (double)c->wall_khz / 1000.0 + 0.5) : 0;
                if (q.res_metro_auto && turns >= 64 && turn_ns > 0.0) {
                    gr_ctx::Metro &m = c->metro;
                    const double T = (double)q.res_metro_ns;
                    if (q.res_metro_ns == 0) {
                        if (!m.off_for) {                            // a free-running launch: the figure to beat (the first probe asks for 3 % less)
                            m.free_ns = m.free_ns > 0 ? std::min(m.free_ns, turn_ns) : turn_ns;
                            if (m.T_ns == 0) m.T_ns = m.free_ns * 0.97;
                        }
                    } else if (turn_ns <= T * 1.012 && late_pm < 50) {   // kept: remember it, probe on unless a period this short has failed
                        m.best_ns = m.best_ns > 0 ? std::min(m.best_ns, T) : T;
                        const double next = T * 0.985;
                        if (m.fail_ns == 0 || next > m.fail_ns * 1.004) m.T_ns = next;
                        else if (m.best_ns > m.free_ns * 0.98) {      // settled, and less than 2 % under the free-running turn: the waits cost what the order buys
                            m.off_for = 512; m.free_ns = 0; m.T_ns = 0; m.best_ns = 0; m.fail_ns = 0;
                        }
                        else if (++m.held >= 64) { m.held = 0; m.fail_ns *= 0.995; }   // (conditions drift: let an old failure fade)
                    } else {                                          // not kept: back to the best kept period, or up
                        m.fail_ns = std::max(m.fail_ns, T);
                        m.T_ns = m.best_ns > T ? m.best_ns : T * 1.02;
                        if (m.best_ns > 0 && m.best_ns <= T) m.best_ns = 0;      // (what was kept once is not kept any more)
                        if (m.T_ns >= m.free_ns * 0.995) {           // the clock buys nothing for this shape (it is not bound by memory): off for a while
                            m.off_for = 512; m.free_ns = 0; m.T_ns = 0; m.best_ns = 0; m.fail_ns = 0;
                        }
                    }
                }
            }
            if (c->profile) {
                float ms = 0.f;
                HIPCHK(c, hipEventElapsedTime(&ms, c->pev[0], c->pev[1]));
                c->prof_ms[3] += ms; c->prof_launches[3] += 1; c->prof_frames[3] += nb;
            }
        }
        for (uint32_t gi = 0; gi < q.n_prof_groups; ++gi) {   // the stream is idle here: read this segment's event pairs
            const uint32_t nf = std::min<uint32_t>(c->sub_batch, nb - gi * c->sub_batch);
            for (int k = 0; k < 3; ++k) {     // 0 sums, 1 finalize (separate launch only when not fused), 2 fit
                if ((k == 1 && q.fused) || (k == 2 && !fit)) continue;
                float ms = 0.f;
                HIPCHK(c, hipEventElapsedTime(&ms, c->pev[6 * gi + 2 * k], c->pev[6 * gi + 2 * k + 1]));
                c->prof_ms[k] += ms; c->prof_launches[k] += 1; c->prof_frames[k] += nf;
            }
        }
        std::vector<GrFrameState> res(c->state_host, c->state_host + nb);
        if (!redo.empty()) {
            // runs of untouched frames go through the two-pass path as segments of their own (the pass is off meanwhile)
            const int keep = c->resident;
            c->resident = 0;
            int st_redo = GR_OK;
            for (uint32_t f0 = 0; f0 < nb && st_redo == GR_OK; ) {
                if (!redo[f0]) { ++f0; continue; }
                uint32_t f1 = f0;
                while (f1 < nb && redo[f1]) ++f1;
                // The run's results are the nested call's FINAL records (after its own exact-path redo of frames whose image proof
                // failed), not c->state_host, which that redo overwrites frame by frame.  The nested return value is a per-frame data
                // error (reported again below, frame by frame) unless the records did not come back: then the device failed.
                std::vector<GrFrameState> fs2;
                Pending sub;
                std::swap(sub, p->pend);                      // (segment_begin / _end work on p->pend)
                st_redo = segment_begin(p, s0 + f0, f1 - f0, fit);
                if (st_redo == GR_OK) {
                    const int st_nested = segment_end(p, nullptr, nullptr, nullptr, &fs2);
                    if (fs2.size() != (size_t)(f1 - f0)) st_redo = st_nested != GR_OK ? st_nested : GR_E_HIP;
                    else for (uint32_t f = f0; f < f1; ++f) res[f] = fs2[f - f0];
                } else { p->pend.active = false; resident_done(c); }
                std::swap(sub, p->pend);
                c->res_redone_frames += f1 - f0;
                f0 = f1;
            }
            c->resident = keep;
            if (st_redo != GR_OK) return st_redo;
        }
        // frames the f32-chain RMSD pass handed back (rmsd too close to the rounding of its own sums: rigid copies of the reference)
        // take the exact-product pass, in runs of consecutive frames; that pass may in turn flag a frame GR_ST_FALLBACK (it cannot:
        // the image proof is the same one the first pass has already held -- but the loop below would pick it up)
        if (q.rmsd_fast) {
            for (uint32_t f0 = 0; f0 < nb; ) {
                if (res[f0].status != GR_ST_REDO_EXACT) { if (q.pre[f0] == GR_OK) c->rmsd_fast_frames++; ++f0; continue; }
                uint32_t f1 = f0;
                while (f1 < nb && res[f1].status == GR_ST_REDO_EXACT) ++f1;
                const uint32_t nf = f1 - f0, nch = batch_chunks(c, sel, nf);
                SlotUse use(c, s0 + f0, nf);
                int st = state_reset(c, nf); if (st) return st;
                GrAccPartial *parts = c->acc_partials;
                k_rmsd_accum<0><<<dim3(nch, nf), dim3(GR_WG), 0, c->stream>>>(c->frames, c->frame_stride, s0 + f0, c->masses, sel, c->boxes_dev, p->dev, c->state_dev, parts);
                k_rmsd_finalize<0><<<dim3(nf), dim3(GR_WG), 0, c->stream>>>(parts, nch, c->frames, c->frame_stride, s0 + f0, sel, c->boxes_dev, p->dev, c->state_dev);
                if (fit) {   // (a masked selection's fit: the frames handed back have not been transformed)
                    const uint32_t gx = fit_grid(c, nf);
                    k_fit_pk<false><<<dim3(gx, nf), dim3(GR_WG), 0, c->stream>>>(c->frames, c->frame_stride, s0 + f0, (uint32_t)c->n, c->boxes_dev, p->dev, c->state_dev, c->masses, sel, nullptr);
                }
                HIPCHK(c, hipGetLastError());
                st = fetch_states(c, nf); if (st) return st;
                for (uint32_t f = f0; f < f1; ++f) res[f] = c->state_host[f - f0];
                c->rmsd_exact_redos += nf;
                f0 = f1;
            }
        }
        // frames whose single-pass image proof failed are redone on the exact (literal multi-pass) path -- in RUNS of consecutive frames,
        // one set of launches and one read-back per run: a group that spans the cell in some direction (a membrane) fails the proof in
        // EVERY frame, and frame-by-frame redo (rounds 1-3) turned such a trajectory into a per-frame loop of six launches and a wait
        for (uint32_t f0 = 0; f0 < nb; ) {
            if (res[f0].status != GR_ST_FALLBACK || (!redo.empty() && redo[f0])) { ++f0; continue; }   // (redo[]: closed by the nested segment above, its own fallbacks included)
            uint32_t f1 = f0;
            while (f1 < nb && res[f1].status == GR_ST_FALLBACK && (redo.empty() || !redo[f1])) ++f1;
            const uint32_t nf = f1 - f0;
            p->last_fallbacks += nf;
            SlotUse use(c, s0 + f0, nf);
            int st = state_reset(c, nf); if (st) return st;
            st = rmsd_exact(p, c, sel, s0 + f0, nf, fit); if (st) return st;
            st = fetch_states(c, nf); if (st) return st;
            for (uint32_t f = f0; f < f1; ++f) res[f] = c->state_host[f - f0];
            f0 = f1;
        }
        for (uint32_t f = 0; f < nb; ++f) {
            int s = res[f].status;
            if (q.pre[f] != GR_OK) { s = q.pre[f]; c->err = q.pre_msg[f]; c->err_index = q.pre_idx[f]; }
            else if (!torn.empty() && torn[f]) s = fail(c, GR_E_HIP, "the resident RMSD-fit pass stalled while this frame was being fitted: some of its atoms carry the fitted coordinates, others the original ones (frame index in gr_last_error_index)", f);
            else if (s == GR_OK && !q.consistent) {
                c->counts[0] = p->n_ref; c->counts[1] = q.group_n;
                s = fail(c, GR_E_INCONSISTENT_GROUP, p->group);
            } else if (s != GR_OK) {
                s = frame_status(c, res[f]);
            }
            note(s);
            if (status_out) status_out[f] = s;
            if (rmsd_out) rmsd_out[f] = (s == GR_OK) ? res[f].rmsd : NAN;
            if (R_out) for (int k = 0; k < 9; ++k) R_out[9 * (size_t)f + k] = (s == GR_OK) ? res[f].R[k] : NAN;
        }
        if (final_states) *final_states = res;
    }
    if (first_err != GR_OK) { c->err = first_err_msg; c->err_index = first_err_index; c->counts[0] = first_counts[0]; c->counts[1] = first_counts[1]; }
    return first_err;
}

static int rmsd_batch_impl(gr_rmsd_plan *p, uint32_t first_slot, uint32_t n_frames, float *rmsd_out, int *status_out, float *R_out, int fit) {
    if (!p || !p->target) return GR_E_INVALID_ARG;
    gr_ctx *c = p->target;
    int st = slot_check(c, first_slot, n_frames); if (st) return st;
    (void)hipSetDevice(c->device);
    if (p->pend.active) return fail(c, GR_E_INVALID_ARG, "a batch begun with gr_rmsd_batch_begin is still in flight");
    p->last_fallbacks = 0;
    int first_err = GR_OK; uint64_t e_idx = 0; std::string e_msg; uint64_t e_cnt[2] = { 0, 0 };
    for (uint32_t b0 = 0; b0 < n_frames; b0 += GR_MAX_BATCH) {
        const uint32_t nb = std::min<uint32_t>(GR_MAX_BATCH, n_frames - b0);
        st = segment_begin(p, first_slot + b0, nb, fit);
        if (st) { p->pend.active = false; resident_done(c); return st; }
        st = segment_end(p, rmsd_out ? rmsd_out + b0 : nullptr, status_out ? status_out + b0 : nullptr, R_out ? R_out + 9 * (size_t)b0 : nullptr);
        if (st == GR_E_HIP) return st;
        if (st != GR_OK && first_err == GR_OK) { first_err = st; e_idx = c->err_index; e_msg = c->err; e_cnt[0] = c->counts[0]; e_cnt[1] = c->counts[1]; }
    }
    if (first_err != GR_OK) { c->err = e_msg; c->err_index = e_idx; c->counts[0] = e_cnt[0]; c->counts[1] = e_cnt[1]; }
    return first_err;
}

int gr_rmsd_batch_begin(gr_rmsd_plan *p, uint32_t first_slot, uint32_t n, int fit) try {
    if (!p || !p->target) return GR_E_INVALID_ARG;
    gr_ctx *c = p->target;
    int st = slot_check(c, first_slot, n); if (st) return st;
    if (n > GR_MAX_BATCH) return fail(c, GR_E_INVALID_ARG, "at most 1024 frames per asynchronous batch");
    if (p->pend.active) return fail(c, GR_E_INVALID_ARG, "a batch is already in flight on this plan");
    (void)hipSetDevice(c->device);
    p->last_fallbacks = 0;
    st = segment_begin(p, first_slot, n, fit ? 1 : 0);
    if (st) { p->pend.active = false; resident_done(c); } else { c->in_flight = p; c->in_flight_s0 = first_slot; c->in_flight_n = n; }
    return st;
} catch (...) { return gr_abi_guard(); }
int gr_rmsd_batch_end(gr_rmsd_plan *p, float *rmsd_out, int *status_out, float *R_out) try {
    if (!p || !p->target) return GR_E_INVALID_ARG;
    (void)hipSetDevice(p->target->device);
    if (p->target->in_flight == p) p->target->in_flight = nullptr;
    return segment_end(p, rmsd_out, status_out, R_out);
} catch (...) { return gr_abi_guard(); }

int gr_rmsd_batch(gr_rmsd_plan *p, uint32_t first_slot, uint32_t n, float *rmsd_out, int *status_out, float *R_out) try {
    return rmsd_batch_impl(p, first_slot, n, rmsd_out, status_out, R_out, 0);
} catch (...) { return gr_abi_guard(); }
int gr_rmsd_fit_batch(gr_rmsd_plan *p, uint32_t first_slot, uint32_t n, float *rmsd_out, int *status_out) try {
    return rmsd_batch_impl(p, first_slot, n, rmsd_out, status_out, nullptr, 1);
} catch (...) { return gr_abi_guard(); }

static int calc_rmsd_impl(gr_ctx *c, uint32_t slot, gr_ctx *ref, uint32_t ref_slot, const char *group, float *rmsd, float *R, int fit) {
    if (!c || !ref) return GR_E_INVALID_ARG;
    int st = slot_check(c, slot); if (st) return st;
    gr_rmsd_plan *p = gr_rmsd_plan_create(ref, ref_slot, c, group, &st);
    if (!p) {   // reference-side failure: report it on the calling context too
        if (ref != c) { c->err = ref->err; c->err_index = ref->err_index; }
        return st;
    }
    float r = NAN, Rm[9];
    st = rmsd_batch_impl(p, slot, 1, &r, nullptr, Rm, fit);
    gr_rmsd_plan_destroy(p);
    if (st) return st;
    if (rmsd) *rmsd = r;
    if (R) memcpy(R, Rm, sizeof(Rm));
    return GR_OK;
}
int gr_calc_rmsd(gr_ctx *c, uint32_t slot, gr_ctx *ref, uint32_t ref_slot, const char *group, float *rmsd, float *R) try {
    return calc_rmsd_impl(c, slot, ref, ref_slot, group, rmsd, R, 0);
} catch (...) { return gr_abi_guard(); }
int gr_calc_rmsd_and_fit(gr_ctx *c, uint32_t slot, gr_ctx *ref, uint32_t ref_slot, const char *group, float *rmsd) try {
    return calc_rmsd_impl(c, slot, ref, ref_slot, group, rmsd, nullptr, 1);
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ xtc reader (host; NEXT-1 of SURVEY.md section 8f) */
struct gr_xtc { grx::File f; };

static int xtc_status(int s) { return s == grx::XTC_OK ? GR_OK : (s == grx::XTC_E_IO ? GR_E_IO : (s == grx::XTC_E_BOX ? GR_E_UNSUPPORTED_BOX : GR_E_FORMAT)); }

gr_xtc *gr_xtc_open(const char *path, int *status) try {
    int dummy; if (!status) status = &dummy;
    if (!path) { *status = GR_E_INVALID_ARG; return nullptr; }
    gr_xtc *x = new gr_xtc();
    const int s = grx::open_file(x->f, path);
    if (s != grx::XTC_OK) { *status = xtc_status(s); if (x->f.fd >= 0) ::close(x->f.fd); delete x; return nullptr; }
    *status = GR_OK;
    return x;
} catch (...) { return nullptr; }
void gr_xtc_close(gr_xtc *x) { if (!x) return; if (x->f.fd >= 0) ::close(x->f.fd); delete x; }
uint64_t gr_xtc_n_atoms(const gr_xtc *x) { return x ? x->f.natoms : 0; }
uint64_t gr_xtc_n_frames(const gr_xtc *x) { return x ? x->f.frames.size() : 0; }

// box matrix (rows = box vectors) -> gro-order box9; rejects boxes GROMACS does not produce (src/io/xdrfile.rs:170-187)
static int xtc_box9(const float m[9], float box9[9]) {
    if (m[1] != 0.0f || m[2] != 0.0f || m[5] != 0.0f) return GR_E_UNSUPPORTED_BOX;
    box9[0] = m[0]; box9[1] = m[4]; box9[2] = m[8]; box9[3] = m[1]; box9[4] = m[2]; box9[5] = m[3]; box9[6] = m[5]; box9[7] = m[6]; box9[8] = m[7];
    return GR_OK;
}

int gr_xtc_frame_info(const gr_xtc *x, uint64_t frame, uint64_t *step, float *time, float box9[9], float *precision) try {
    if (!x) return GR_E_INVALID_ARG;
    if (frame >= x->f.frames.size()) return GR_E_OUT_OF_RANGE;
    const grx::FrameIndex &fi = x->f.frames[frame];
    if (step) *step = (uint64_t)(uint32_t)fi.step;   // i32 -> u32 -> u64 like src/io/xtc_io/xdrfile_xtc.rs:98-100
    if (time) *time = fi.time;
    if (precision) *precision = fi.precision;
    return box9 ? xtc_box9(fi.box, box9) : GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_xtc_read_frame(const gr_xtc *x, uint64_t frame, float *xyz, float box9[9], uint64_t *step, float *time, float *precision) try {
    if (!x || !xyz) return GR_E_INVALID_ARG;
    int st = gr_xtc_frame_info(x, frame, step, time, box9, precision);
    if (st != GR_OK) return st;
    static thread_local std::vector<unsigned char> scratch;   // one bit-stream buffer per decoding thread
    return xtc_status(grx::decode_frame(x->f, x->f.frames[frame], xyz, scratch));
} catch (...) { return gr_abi_guard(); }

// frames of an xtc file -> frame slots, unpacked on the device.  group == nullptr: every atom (XtcReader); else only the atoms of
// the group change (GroupXtcReader, molly_xtc.rs:440-470,585-587): the host reads + skims the bit stream only up to the group's
// last atom, only that prefix crosses PCIe, and the unpack kernel writes the group's atoms only.
static int xtc_read_frames_device_impl(const gr_xtc *x, uint64_t first_frame, uint32_t n_frames, uint64_t frame_step, gr_ctx *c,
                                       uint32_t first_slot, const char *group, int host_threads, uint64_t *steps, float *times) {
    if (!x || !c) return GR_E_INVALID_ARG;
    int st = slot_check(c, first_slot, n_frames, true); if (st) return st;
    if (frame_step == 0) frame_step = 1;
    if (first_frame + (uint64_t)(n_frames - 1) * frame_step >= x->f.frames.size()) return fail(c, GR_E_OUT_OF_RANGE, "xtc frame out of range", first_frame);
    if (x->f.natoms != c->n) return fail(c, GR_E_INVALID_ARG, "the trajectory's atom count differs from the context's");
    (void)hipSetDevice(c->device);
    const uint32_t n = x->f.natoms;
    Group *g = nullptr;
    uint32_t n_stop = n;
    if (group) {
        auto it = c->groups.find(group);
        if (it == c->groups.end()) return fail(c, GR_E_GROUP_NOT_FOUND, group);            // ReadTrajError::GroupNotFound
        g = &it->second;
        n_stop = g->blocks.empty() ? 0u : (uint32_t)(g->blocks.back().second + 1);       // everything up to the group's last atom
        if (!g->mask_dev && g->n) {
            std::vector<uint32_t> bits(((size_t)c->n_pad + 31) / 32, 0u);
            for (const auto &b : g->blocks) for (uint64_t a = b.first; a <= b.second; ++a) bits[a >> 5] |= 1u << (a & 31u);
            HIPCHK(c, hipMalloc(&g->mask_dev, bits.size() * sizeof(uint32_t)));
            HIPCHK(c, hipMemcpy(g->mask_dev, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    if (n <= 9) {   // uncompressed frames: nothing to unpack
        std::vector<float> xyz(3 * (size_t)n), old(3 * (size_t)n);
        for (uint32_t k = 0; k < n_frames; ++k) {
            float box9[9];
            st = gr_xtc_read_frame(x, first_frame + k * frame_step, xyz.data(), box9, steps ? steps + k : nullptr, times ? times + k : nullptr, nullptr); if (st) return st;
            if (g) {   // keep the atoms outside the group
                st = gr_frame_download(c, first_slot + k, old.data()); if (st) return st;
                for (uint32_t a = 0; a < n; ++a) if (!grc::isin(g->blocks, a)) memcpy(&xyz[3 * a], &old[3 * a], 12);
            }
            st = gr_frame_upload(c, first_slot + k, xyz.data(), box9); if (st) return st;
            st = gr_frame_upload_wait(c, first_slot + k); if (st) return st;
        }
        return GR_OK;
    }
    static const bool trace = getenv("GR_XTC_TRACE") != nullptr;   // host phase times to stderr (tracing only: no effect on results)
    const auto t_call = std::chrono::steady_clock::now();
    // ---- layout of the batch's staging buffer: [streams (16-byte aligned, 16 zero bytes behind each)] [descs] [slots] [checkpoints]
    const uint32_t ncp = (n_stop + GR_XTC_CP_ATOMS - 1) / GR_XTC_CP_ATOMS;
    std::vector<size_t> soff(n_frames), sread(n_frames);
    size_t bytes = 0;
    for (uint32_t k = 0; k < n_frames; ++k) {
        const grx::FrameIndex &fi = x->f.frames[first_frame + k * frame_step];
        soff[k] = bytes; bytes += ((size_t)fi.nbytes + 16 + 15) & ~(size_t)15;
        // a partial read starts with the share of the stream the group's prefix should need (mean bits per atom + 30 %)
        sread[k] = n_stop >= n ? (size_t)fi.nbytes : std::min<size_t>((size_t)fi.nbytes, (size_t)((double)fi.nbytes * ((double)n_stop / (double)n) * 1.3) + 256);
    }
    const size_t stream_bytes = bytes;
    const size_t off_desc = bytes;  bytes += (size_t)n_frames * sizeof(grx::FrameDesc);
    const size_t off_slot = bytes;  bytes += (((size_t)n_frames * sizeof(uint32_t)) + 15) & ~(size_t)15;
    const size_t off_cp = bytes;    bytes += (size_t)n_frames * (ncp ? ncp : 1) * sizeof(grx::Checkpoint);
    const uint32_t bank = c->xtc_bank; c->xtc_bank ^= 1u;
    if (c->xtc_ev[bank]) HIPCHK(c, hipEventSynchronize(c->xtc_ev[bank]));       // the batch before the previous one has left this bank
    else HIPCHK(c, hipEventCreateWithFlags(&c->xtc_ev[bank], hipEventDisableTiming));
    if (bytes > c->xtc_host_cap[bank]) {
        if (c->xtc_host[bank]) (void)hipHostFree(c->xtc_host[bank]);
        c->xtc_host[bank] = nullptr; c->xtc_host_cap[bank] = 0;
        const size_t cap = bytes + bytes / 4;
        HIPCHK(c, hipHostMalloc(&c->xtc_host[bank], cap, hipHostMallocDefault));
        c->xtc_host_cap[bank] = cap;
    }
    if (!c->unpack_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->unpack_stream, hipStreamNonBlocking));
    if (!c->xtc_unpacked[bank]) HIPCHK(c, hipEventCreateWithFlags(&c->xtc_unpacked[bank], hipEventDisableTiming));
    if (bytes > c->xtc_dev_cap[bank]) {
        HIPCHK(c, hipStreamSynchronize(c->unpack_stream));        // nobody reads the old bank any more
        if (c->xtc_dev[bank]) (void)hipFree(c->xtc_dev[bank]);
        c->xtc_dev[bank] = nullptr; c->xtc_dev_cap[bank] = 0;
        const size_t cap = bytes + bytes / 4;
        HIPCHK(c, hipMalloc(&c->xtc_dev[bank], cap));
        c->xtc_dev_cap[bank] = cap;
    }
    unsigned char *H = c->xtc_host[bank];
    grx::FrameDesc *descs = reinterpret_cast<grx::FrameDesc *>(H + off_desc);
    uint32_t *slots = reinterpret_cast<uint32_t *>(H + off_slot);
    grx::Checkpoint *cps = reinterpret_cast<grx::Checkpoint *>(H + off_cp);
    // ---- host: read + skim, one frame per worker at a time (walking two frames interleaved in one thread was measured:
    // slower per frame on the EPYC hosts -- the walk is bound by its instruction count, not by the latency of its loads)
    std::atomic<uint32_t> next(0);
    std::atomic<int> bad(grx::XTC_OK);
    std::atomic<uint64_t> ns_read(0), ns_skim(0);
    const auto t_begin = std::chrono::steady_clock::now();
    auto work = [&]() {
        std::vector<grx::Checkpoint> local;
        for (;;) {
            const uint32_t k = next.fetch_add(1);
            if (k >= n_frames) return;
            const grx::FrameIndex &fi = x->f.frames[first_frame + k * frame_step];
            unsigned char *dst = H + soff[k];
            const auto t0 = std::chrono::steady_clock::now();
            if (!grx::pread_all(x->f.fd, dst, sread[k], fi.data_offset)) { bad = grx::XTC_E_IO; return; }
            memset(dst + sread[k], 0, (((size_t)fi.nbytes + 16 + 15) & ~(size_t)15) - sread[k]);
            grx::FrameDesc d;
            memset(&d, 0, sizeof d);
            d.stream_off = soff[k]; d.cp_off = (uint64_t)k * ncp;
            const auto t1 = std::chrono::steady_clock::now();
            int r = grx::skim_frame(dst, fi, n, d, local, n_stop, sread[k] < (size_t)fi.nbytes ? (uint64_t)sread[k] : ~0ull);
            if (r == grx::XTC_E_RANGE) {   // the estimate of the prefix was too short: read the whole stream, walk again
                if (!grx::pread_all(x->f.fd, dst, (size_t)fi.nbytes, fi.data_offset)) { bad = grx::XTC_E_IO; return; }
                r = grx::skim_frame(dst, fi, n, d, local, n_stop);
            }
            if (r != grx::XTC_OK || local.size() != ncp) { bad = r != grx::XTC_OK ? r : (int)grx::XTC_E_FORMAT; return; }
            if (ncp) memcpy(cps + (size_t)k * ncp, local.data(), ncp * sizeof(grx::Checkpoint));
            descs[k] = d; slots[k] = first_slot + k;
            if (trace) {
                const auto t2 = std::chrono::steady_clock::now();
                ns_read += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
                ns_skim += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
            }
        }
    };
    uint32_t nt = host_threads > 0 ? (uint32_t)host_threads : std::min<uint32_t>(n_frames, 16u);
    nt = std::max<uint32_t>(1u, std::min<uint32_t>(nt, n_frames));
    if (nt == 1) work();
    else {
        // (a thread that cannot be started must not unwind through the C ABI: the frames it would have taken are picked up by
        // the workers that did start, or by this thread)
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < nt; ++t) { try { th.emplace_back(work); } catch (const std::system_error &) { break; } }
        if (th.empty()) work();
        for (auto &t : th) t.join();
    }
    if (bad.load() != grx::XTC_OK) return fail(c, xtc_status(bad.load()), "corrupt or unreadable xtc frame");
    const auto t_host = std::chrono::steady_clock::now();
    // ---- device.  copy stream: the bank's H2D, once the unpack kernel that last read the device bank is done.
    // unpack stream: behind that copy and behind the last kernels that still read these slots -- the boxes (one copy from
    // the pinned boxes_host range), the unpack kernel, then the slots' ready events.
    unsigned char *D = c->xtc_dev[bank];
    for (uint32_t k = 0; k < n_frames; ++k) {
        const uint32_t slot = first_slot + k;
        // boxes_host[slot] is about to be rewritten: the slot's previous box copy must have left it (see gr_frame_upload)
        if (c->ev_ready[slot]) HIPCHK(c, hipEventSynchronize(c->ev_ready[slot]));
        else HIPCHK(c, hipEventCreateWithFlags(&c->ev_ready[slot], hipEventDisableTiming));
    }
    for (uint32_t k = 0; k < n_frames; ++k) {
        const uint64_t fr = first_frame + k * frame_step;
        float box9[9];
        st = gr_xtc_frame_info(x, fr, steps ? steps + k : nullptr, times ? times + k : nullptr, box9, nullptr);
        if (st != GR_OK) return fail(c, st, "unsupported box in xtc frame", fr);
        box_fill(c, first_slot + k, box9);
    }
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->xtc_unpacked[bank], 0));
    if (n_stop >= n) {
        HIPCHK(c, hipMemcpyAsync(D, H, bytes, hipMemcpyHostToDevice, c->copy_stream));
    } else {
        // partial read: only the stream prefixes the skim said the group needs cross PCIe, then the tables
        for (uint32_t k = 0; k < n_frames; ++k)
            if (descs[k].nbytes) HIPCHK(c, hipMemcpyAsync(D + soff[k], H + soff[k], ((size_t)descs[k].nbytes + 16 + 15) & ~(size_t)15, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(c, hipMemcpyAsync(D + stream_bytes, H + stream_bytes, bytes - stream_bytes, hipMemcpyHostToDevice, c->copy_stream));
    }
    HIPCHK(c, hipEventRecord(c->xtc_ev[bank], c->copy_stream));
    hipStream_t U = c->unpack_stream;
    HIPCHK(c, hipStreamWaitEvent(U, c->xtc_ev[bank], 0));
    uint64_t waited = 0;
    for (uint32_t k = 0; k < n_frames; ++k) {
        const uint64_t gen = c->slot_gen[first_slot + k];
        if (gen && gen != waited) { HIPCHK(c, hipStreamWaitEvent(U, c->ev_done_ring[gen % 64], 0)); waited = gen; }
    }
    HIPCHK(c, hipMemcpyAsync(c->boxes_dev + first_slot, c->boxes_host + first_slot, (size_t)n_frames * sizeof(GrBox), hipMemcpyHostToDevice, U));
    if (ncp) {
        k_xtc_unpack<<<dim3((ncp + 255) / 256, n_frames), dim3(256), 0, U>>>(
            D, reinterpret_cast<const grx::FrameDesc *>(D + off_desc), reinterpret_cast<const grx::Checkpoint *>(D + off_cp),
            c->frames, c->frame_stride, reinterpret_cast<const uint32_t *>(D + off_slot), n, g ? g->mask_dev : nullptr);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->xtc_unpacked[bank], U));
    for (uint32_t k = 0; k < n_frames; ++k) {
        HIPCHK(c, hipEventRecord(c->ev_ready[first_slot + k], U));
        c->upload_pending[first_slot + k] = 1;
    }
    if (trace) {
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[xtc] %u frames, %u threads: waits + buffers %.2f ms, read + skim %.2f ms (per frame, thread time: read %.3f, skim %.3f), enqueue %.2f ms\n",
                n_frames, nt, ms(t_call, t_begin), ms(t_begin, t_host), ns_read.load() * 1e-6 / n_frames, ns_skim.load() * 1e-6 / n_frames, ms(t_host, t_end));
    }
    return GR_OK;
}

int gr_xtc_read_frames_device(const gr_xtc *x, uint64_t first_frame, uint32_t n_frames, uint64_t frame_step, gr_ctx *c,
                              uint32_t first_slot, int host_threads, uint64_t *steps, float *times) try {
    return xtc_read_frames_device_impl(x, first_frame, n_frames, frame_step, c, first_slot, nullptr, host_threads, steps, times);
} catch (...) { return gr_abi_guard(); }

int gr_xtc_read_frames_device_group(const gr_xtc *x, uint64_t first_frame, uint32_t n_frames, uint64_t frame_step, gr_ctx *c,
                                    uint32_t first_slot, const char *group, int host_threads, uint64_t *steps, float *times) try {
    if (!group) return c ? fail(c, GR_E_GROUP_NOT_FOUND, "(null)") : GR_E_INVALID_ARG;
    return xtc_read_frames_device_impl(x, first_frame, n_frames, frame_step, c, first_slot, group, host_threads, steps, times);
} catch (...) { return gr_abi_guard(); }

int gr_xtc_read_frame_prefix(const gr_xtc *x, uint64_t frame, uint64_t n_prefix, float *xyz, float box9[9], uint64_t *step, float *time, float *precision,
                             uint64_t *stream_bytes_read) try {
    if (!x || (!xyz && n_prefix)) return GR_E_INVALID_ARG;
    int st = gr_xtc_frame_info(x, frame, step, time, box9, precision);
    if (st != GR_OK) return st;
    static thread_local std::vector<unsigned char> scratch;
    size_t got = 0;
    const uint32_t want = (uint32_t)std::min<uint64_t>(n_prefix, x->f.natoms);
    st = xtc_status(grx::decode_frame_prefix(x->f, x->f.frames[frame], want, xyz, scratch, &got));
    if (stream_bytes_read) *stream_bytes_read = got;
    return st;
} catch (...) { return gr_abi_guard(); }

static void box9_rows(const float *b, float m[9]);
/* ------------------------------------------------------------ trr reader / writer */
struct gr_trr { grtr::File f; };
static int trr_status(int s) { return s == grtr::TRR_OK ? GR_OK : (s == grtr::TRR_E_IO ? GR_E_IO : GR_E_FORMAT); }

gr_trr *gr_trr_open(const char *path, int *status) try {
    int dummy; if (!status) status = &dummy;
    if (!path) { *status = GR_E_INVALID_ARG; return nullptr; }
    gr_trr *t = new gr_trr();
    const int s = grtr::open_file(t->f, path);
    if (s != grtr::TRR_OK) { *status = trr_status(s); if (t->f.fd >= 0) ::close(t->f.fd); delete t; return nullptr; }
    *status = GR_OK;
    return t;
} catch (...) { return nullptr; }
void gr_trr_close(gr_trr *t) { if (!t) return; if (t->f.fd >= 0) ::close(t->f.fd); delete t; }
uint64_t gr_trr_n_atoms(const gr_trr *t) { return t ? t->f.natoms : 0; }
uint64_t gr_trr_n_frames(const gr_trr *t) { return t ? t->f.frames.size() : 0; }

int gr_trr_frame_info(const gr_trr *t, uint64_t frame, uint64_t *step, float *time, float *lambda, float box9[9], int *sections, int *double_precision) try {
    if (!t) return GR_E_INVALID_ARG;
    if (frame >= t->f.frames.size()) return GR_E_OUT_OF_RANGE;
    const grtr::FrameIndex &fi = t->f.frames[frame];
    if (step) *step = (uint64_t)(uint32_t)fi.step;
    if (time) *time = fi.time;
    if (lambda) *lambda = fi.lambda;
    if (sections) *sections = (fi.x_off ? 1 : 0) | (fi.v_off ? 2 : 0) | (fi.f_off ? 4 : 0) | (fi.has_box ? 8 : 0);
    if (double_precision) *double_precision = fi.real_size == 8;
    return box9 ? xtc_box9(fi.box, box9) : GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_trr_read_frame(const gr_trr *t, uint64_t frame, float *xyz, float *velocities, float *forces, float box9[9], uint64_t *step, float *time, float *lambda) try {
    if (!t) return GR_E_INVALID_ARG;
    int st = gr_trr_frame_info(t, frame, step, time, lambda, box9, nullptr, nullptr);
    if (st != GR_OK) return st;
    static thread_local std::vector<unsigned char> scratch;
    const grtr::FrameIndex &fi = t->f.frames[frame];
    st = trr_status(grtr::read_section(t->f, fi, fi.x_off, xyz, scratch)); if (st) return st;
    st = trr_status(grtr::read_section(t->f, fi, fi.v_off, velocities, scratch)); if (st) return st;
    return trr_status(grtr::read_section(t->f, fi, fi.f_off, forces, scratch));
} catch (...) { return gr_abi_guard(); }

struct gr_trr_writer { FILE *fp = nullptr; };
gr_trr_writer *gr_trr_writer_open(const char *path, int *status) try {
    int dummy; if (!status) status = &dummy;
    if (!path) { *status = GR_E_INVALID_ARG; return nullptr; }
    FILE *fp = fopen(path, "wb");
    if (!fp) { *status = GR_E_IO; return nullptr; }
    gr_trr_writer *w = new gr_trr_writer(); w->fp = fp;
    *status = GR_OK;
    return w;
} catch (...) { return nullptr; }
int gr_trr_writer_close(gr_trr_writer *w) { if (!w) return GR_E_INVALID_ARG; const int r = w->fp ? fclose(w->fp) : 0; delete w; return r == 0 ? GR_OK : GR_E_IO; }
int gr_trr_write_frame(gr_trr_writer *w, uint64_t n, const float *xyz, const float *velocities, const float *forces, const float box9[9],
                       int64_t step, float time, float lambda) try {
    if (!w || !w->fp || n > 0x0fffffffull) return GR_E_INVALID_ARG;
    float m[9]; box9_rows(box9, m);
    std::vector<unsigned char> out;
    grtr::serialise_frame(out, (uint32_t)n, (int32_t)step, time, lambda, m, xyz, velocities, forces);
    return fwrite(out.data(), 1, out.size(), w->fp) == out.size() ? GR_OK : GR_E_IO;
} catch (...) { return gr_abi_guard(); }

// raw big-endian position sections of a batch of frames -> frame slots (f32); an all-zero position is "no position"
// (TrrFrameData::update_system, trr_io.rs:108-112) and travels as NaN in x like everywhere else in this library
__global__ __launch_bounds__(256) void k_trr_unpack(const unsigned char *__restrict__ raw, const uint64_t *__restrict__ sec_off, const uint32_t *__restrict__ real_size,
                                                    float *__restrict__ frames, size_t frame_stride, const uint32_t *__restrict__ slots, uint32_t n_atoms) {
    const uint32_t k = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_atoms) return;
    const uint64_t so = sec_off[k];
    float v[3] = { 0.f, 0.f, 0.f };
    if (so != ~0ull) {
        const unsigned char *src = raw + so;
        if (real_size[k] == 4) {
            for (int a = 0; a < 3; ++a) { const uint32_t u = __builtin_bswap32(reinterpret_cast<const uint32_t *>(src)[3 * (size_t)i + a]); v[a] = __uint_as_float(u); }
        } else {
            for (int a = 0; a < 3; ++a) { const uint64_t u = __builtin_bswap64(reinterpret_cast<const uint64_t *>(src)[3 * (size_t)i + a]); v[a] = (float)__longlong_as_double((long long)u); }
        }
    }
    if (v[0] == 0.0f && v[1] == 0.0f && v[2] == 0.0f) v[0] = v[1] = v[2] = __uint_as_float(0x7fc00000u);      // (no position: NaN in x -- and, inside the library, in y and z: gr_layout.h::k_tile)
    gr_pos_store(frames + (size_t)slots[k] * frame_stride, i, v[0], v[1], v[2]);
}

int gr_trr_read_frames_device(const gr_trr *t, uint64_t first_frame, uint32_t n_frames, uint64_t frame_step, gr_ctx *c, uint32_t first_slot,
                              uint64_t *steps, float *times) try {
    if (!t || !c) return GR_E_INVALID_ARG;
    int st = slot_check(c, first_slot, n_frames, true); if (st) return st;
    if (frame_step == 0) frame_step = 1;
    if (n_frames == 0) return GR_OK;
    if (first_frame + (uint64_t)(n_frames - 1) * frame_step >= t->f.frames.size()) return fail(c, GR_E_OUT_OF_RANGE, "trr frame out of range", first_frame);
    if (t->f.natoms != c->n) return fail(c, GR_E_INVALID_ARG, "the trajectory's atom count differs from the context's");
    (void)hipSetDevice(c->device);
    // staging: [raw sections, 16-byte aligned] [section offsets u64] [real sizes u32] [slots u32]; the xtc banks are reused
    const size_t n3 = (size_t)t->f.natoms * 3;
    std::vector<size_t> soff(n_frames);
    size_t bytes = 0;
    for (uint32_t k = 0; k < n_frames; ++k) {
        const grtr::FrameIndex &fi = t->f.frames[first_frame + k * frame_step];
        soff[k] = bytes; bytes += ((fi.x_off ? n3 * fi.real_size : 0) + 15) & ~(size_t)15;
    }
    const size_t off_sec = bytes;  bytes += (size_t)n_frames * sizeof(uint64_t);
    const size_t off_rs = bytes;   bytes += (((size_t)n_frames * sizeof(uint32_t)) + 15) & ~(size_t)15;
    const size_t off_slot = bytes; bytes += (((size_t)n_frames * sizeof(uint32_t)) + 15) & ~(size_t)15;
    const uint32_t bank = c->xtc_bank; c->xtc_bank ^= 1u;
    if (c->xtc_ev[bank]) HIPCHK(c, hipEventSynchronize(c->xtc_ev[bank]));
    else HIPCHK(c, hipEventCreateWithFlags(&c->xtc_ev[bank], hipEventDisableTiming));
    if (bytes > c->xtc_host_cap[bank]) {
        if (c->xtc_host[bank]) (void)hipHostFree(c->xtc_host[bank]);
        c->xtc_host[bank] = nullptr; c->xtc_host_cap[bank] = 0;
        HIPCHK(c, hipHostMalloc(&c->xtc_host[bank], bytes + bytes / 4, hipHostMallocDefault));
        c->xtc_host_cap[bank] = bytes + bytes / 4;
    }
    if (!c->unpack_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->unpack_stream, hipStreamNonBlocking));
    if (!c->xtc_unpacked[bank]) HIPCHK(c, hipEventCreateWithFlags(&c->xtc_unpacked[bank], hipEventDisableTiming));
    if (bytes > c->xtc_dev_cap[bank]) {
        HIPCHK(c, hipStreamSynchronize(c->unpack_stream));
        if (c->xtc_dev[bank]) (void)hipFree(c->xtc_dev[bank]);
        c->xtc_dev[bank] = nullptr; c->xtc_dev_cap[bank] = 0;
        HIPCHK(c, hipMalloc(&c->xtc_dev[bank], bytes + bytes / 4));
        c->xtc_dev_cap[bank] = bytes + bytes / 4;
    }
    unsigned char *H = c->xtc_host[bank], *D = c->xtc_dev[bank];
    uint64_t *sec = reinterpret_cast<uint64_t *>(H + off_sec);
    uint32_t *rsz = reinterpret_cast<uint32_t *>(H + off_rs), *slots = reinterpret_cast<uint32_t *>(H + off_slot);
    for (uint32_t k = 0; k < n_frames; ++k) {
        const uint64_t fr = first_frame + k * frame_step;
        const grtr::FrameIndex &fi = t->f.frames[fr];
        const uint32_t slot = first_slot + k;
        if (fi.x_off) {
            if (!grtr::pread_all(t->f.fd, H + soff[k], n3 * fi.real_size, fi.x_off)) return fail(c, GR_E_IO, "short read in trr frame", fr);
            sec[k] = soff[k];
        } else sec[k] = ~0ull;
        rsz[k] = fi.real_size; slots[k] = slot;
        if (c->ev_ready[slot]) HIPCHK(c, hipEventSynchronize(c->ev_ready[slot]));
        else HIPCHK(c, hipEventCreateWithFlags(&c->ev_ready[slot], hipEventDisableTiming));
        float box9[9];
        st = gr_trr_frame_info(t, fr, steps ? steps + k : nullptr, times ? times + k : nullptr, nullptr, box9, nullptr, nullptr);
        if (st != GR_OK) return fail(c, st, "unsupported box in trr frame", fr);
        box_fill(c, slot, fi.has_box ? box9 : nullptr);
    }
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->xtc_unpacked[bank], 0));
    HIPCHK(c, hipMemcpyAsync(D, H, bytes, hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(c, hipEventRecord(c->xtc_ev[bank], c->copy_stream));
    hipStream_t U = c->unpack_stream;
    HIPCHK(c, hipStreamWaitEvent(U, c->xtc_ev[bank], 0));
    uint64_t waited = 0;
    for (uint32_t k = 0; k < n_frames; ++k) {
        const uint64_t gen = c->slot_gen[first_slot + k];
        if (gen && gen != waited) { HIPCHK(c, hipStreamWaitEvent(U, c->ev_done_ring[gen % 64], 0)); waited = gen; }
    }
    HIPCHK(c, hipMemcpyAsync(c->boxes_dev + first_slot, c->boxes_host + first_slot, (size_t)n_frames * sizeof(GrBox), hipMemcpyHostToDevice, U));
    k_trr_unpack<<<dim3((t->f.natoms + 255) / 256, n_frames), dim3(256), 0, U>>>(D, reinterpret_cast<const uint64_t *>(D + off_sec), reinterpret_cast<const uint32_t *>(D + off_rs),
                                                                                  c->frames, c->frame_stride, reinterpret_cast<const uint32_t *>(D + off_slot), t->f.natoms);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->xtc_unpacked[bank], U));
    for (uint32_t k = 0; k < n_frames; ++k) {
        HIPCHK(c, hipEventRecord(c->ev_ready[first_slot + k], U));
        c->upload_pending[first_slot + k] = 1;
    }
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ text front end */
struct gr_structure { grt::Structure s; };
struct gr_ndx { std::vector<grt::NdxGroup> g; };
static int parse_result(int rc, const std::string &d, int *code, char *detail, size_t cap) {
    if (code) *code = rc;
    if (detail && cap) { const size_t n = std::min(cap - 1, d.size()); memcpy(detail, d.data(), n); detail[n] = 0; }
    return rc == grt::P_OK ? GR_OK : (rc == grt::P_FILE_NOT_FOUND ? GR_E_IO : GR_E_FORMAT);
}
int gr_gro_read(const char *path, gr_structure **out, int *code, char *detail, size_t cap) try {
    if (!path || !out) return GR_E_INVALID_ARG;
    *out = nullptr;
    gr_structure *s = new gr_structure();
    std::string d;
    const int rc = grt::read_gro(path, s->s, d);
    if (rc != grt::P_OK) { delete s; return parse_result(rc, d, code, detail, cap); }
    *out = s;
    return parse_result(rc, d, code, detail, cap);
} catch (...) { return gr_abi_guard(); }
void gr_structure_free(gr_structure *s) { delete s; }
uint64_t gr_structure_n_atoms(const gr_structure *s) { return s ? s->s.atoms.size() : 0; }
const char *gr_structure_title(const gr_structure *s) { return s ? s->s.title.c_str() : ""; }
int gr_structure_box(const gr_structure *s, float box9[9]) try {
    if (!s || !box9) return GR_E_INVALID_ARG;
    if (!s->s.has_box) return GR_E_NO_BOX;
    memcpy(box9, s->s.box9, 9 * sizeof(float));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_structure_positions(const gr_structure *s, float *xyz) try {
    if (!s || !xyz) return GR_E_INVALID_ARG;
    for (size_t i = 0; i < s->s.atoms.size(); ++i) memcpy(xyz + 3 * i, s->s.atoms[i].pos, 12);
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_structure_velocities(const gr_structure *s, float *vel) try {
    if (!s || !vel) return GR_E_INVALID_ARG;
    for (size_t i = 0; i < s->s.atoms.size(); ++i) memcpy(vel + 3 * i, s->s.atoms[i].vel, 12);
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_structure_atom(const gr_structure *s, uint64_t i, uint64_t *resid, uint64_t *atomid, char resname[8], char atomname[8]) try {
    if (!s) return GR_E_INVALID_ARG;
    if (i >= s->s.atoms.size()) return GR_E_OUT_OF_RANGE;
    const grt::Atom &a = s->s.atoms[i];
    if (resid) *resid = a.resid;
    if (atomid) *atomid = a.atomid;
    if (resname) { strncpy(resname, a.resname.c_str(), 7); resname[7] = 0; }
    if (atomname) { strncpy(atomname, a.atomname.c_str(), 7); atomname[7] = 0; }
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_ndx_read(const char *path, uint64_t n_atoms, gr_ndx **out, int *code, char *detail, size_t cap) try {
    if (!path || !out) return GR_E_INVALID_ARG;
    *out = nullptr;
    gr_ndx *x = new gr_ndx();
    std::string d; uint64_t bad = 0;
    const int rc = grt::read_ndx(path, n_atoms, x->g, d, bad);
    if (rc != grt::P_OK) { delete x; return parse_result(rc, d, code, detail, cap); }
    *out = x;
    return parse_result(rc, d, code, detail, cap);
} catch (...) { return gr_abi_guard(); }
void gr_ndx_free(gr_ndx *x) { delete x; }
size_t gr_ndx_n_groups(const gr_ndx *x) { return x ? x->g.size() : 0; }
const char *gr_ndx_group_name(const gr_ndx *x, size_t g) { return (x && g < x->g.size()) ? x->g[g].name.c_str() : ""; }
size_t gr_ndx_group_size(const gr_ndx *x, size_t g) { return (x && g < x->g.size()) ? x->g[g].indices.size() : 0; }
int gr_ndx_group_indices(const gr_ndx *x, size_t g, uint64_t *out) try {
    if (!x || g >= x->g.size() || !out) return GR_E_INVALID_ARG;
    memcpy(out, x->g[g].indices.data(), x->g[g].indices.size() * sizeof(uint64_t));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_ndx_install(const gr_ndx *x, gr_ctx *c, size_t *n_invalid, size_t *n_dup) try {
    if (!x || !c) return GR_E_INVALID_ARG;
    (void)hipSetDevice(c->device);
    std::set<std::string> invalid, dup;
    for (const auto &g : x->g) {
        if (!name_is_valid(g.name.c_str())) { invalid.insert(g.name); continue; }                       // Groups::add -> InvalidName: not added
        const int st = install_group(c, g.name.c_str(), grc::from_indices(g.indices, c->n));
        if (st == GR_E_GROUP_EXISTS) dup.insert(g.name);
        else if (st != GR_OK) return st;
    }
    if (n_invalid) *n_invalid = invalid.size();
    if (n_dup) *n_dup = dup.size();
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ xtc writer */
struct gr_xtc_writer { FILE *fp = nullptr; };

gr_xtc_writer *gr_xtc_writer_open(const char *path, int *status) try {
    int dummy; if (!status) status = &dummy;
    if (!path) { *status = GR_E_INVALID_ARG; return nullptr; }
    FILE *fp = fopen(path, "wb");
    if (!fp) { *status = GR_E_IO; return nullptr; }
    gr_xtc_writer *w = new gr_xtc_writer(); w->fp = fp; *status = GR_OK;
    return w;
} catch (...) { return nullptr; }
int gr_xtc_writer_close(gr_xtc_writer *w) try {
    if (!w) return GR_E_INVALID_ARG;
    const int rc = w->fp ? fclose(w->fp) : 0;
    delete w;
    return rc == 0 ? GR_OK : GR_E_IO;
} catch (...) { return gr_abi_guard(); }
// gro-order box9 -> rows = box vectors (simbox2matrix, xdrfile.rs:188-200); NULL -> zero matrix
static void box9_rows(const float *b, float m[9]) {
    if (!b) { memset(m, 0, 9 * sizeof(float)); return; }
    m[0] = b[0]; m[1] = b[3]; m[2] = b[4]; m[3] = b[5]; m[4] = b[1]; m[5] = b[6]; m[6] = b[7]; m[7] = b[8]; m[8] = b[2];
}
int gr_xtc_write_frame(gr_xtc_writer *w, uint64_t n, const float *xyz, const float box9[9], int64_t step, float time, float precision) try {
    if (!w || !w->fp || (!xyz && n) || n > 0x7fffffffull) return GR_E_INVALID_ARG;
    float m[9]; box9_rows(box9, m);
    std::vector<unsigned char> out; grx::EncodedFrame sc; std::vector<int> ints;
    // coordinate x precision beyond the format's 32-bit integers (or NaN in y / z): nothing is written
    if (!grx::serialise_frame(out, (uint32_t)n, (int32_t)step, time, m, xyz, precision, sc, ints)) return GR_E_OUT_OF_RANGE;
    return fwrite(out.data(), 1, out.size(), w->fp) == out.size() ? GR_OK : GR_E_IO;
} catch (...) { return gr_abi_guard(); }
// gr_xtc_write_slots on the device encoder: rounds of frames whose scratch (30 bytes per atom and frame) stays below ~4 GB
static int xe_reserve(gr_ctx *c, int k, size_t bytes) {
    if (bytes <= c->xe_cap[k]) return GR_OK;
    if (c->xe_dev[k]) (void)hipFree(c->xe_dev[k]);
    c->xe_dev[k] = nullptr; c->xe_cap[k] = 0;
    HIPCHK(c, hipMalloc(&c->xe_dev[k], bytes));
    c->xe_cap[k] = bytes;
    return GR_OK;
}
// *declined: the scratch buffers could not be had (nothing has been written): the caller takes the host encoders
static int xtc_write_slots_device(gr_xtc_writer *w, gr_ctx *c, uint32_t first_slot, uint32_t n_frames, const Group *g,
                                  const int64_t *steps, const float *times, float precision, bool *declined) {
    *declined = false;
    if (!(precision > 0.0f)) precision = 1000.0f;
    const uint32_t n = (uint32_t)(g ? g->n : c->n);
    GrSel sel;
    if (g) sel = make_sel(*g);
    else { Group all; all.n = c->n; all.contiguous = true; all.start = 0; sel = make_sel(all); }
    // Rounds of frames: while the calling thread drives the kernels and copies of round r + 1, a writer thread puts round r into the
    // file (the file write is the slowest stage: 4-5 GB/s from one thread).  A round is ~64 MB of output (so the overlap has something
    // to overlap with) and at most ~2 GB of scratch (30 bytes per atom and frame); two pinned banks take the rounds in turn.
    const size_t per_frame = (size_t)n * 30u;
    uint32_t round = (uint32_t)std::max<size_t>(1, std::min<size_t>(n_frames, ((size_t)2 << 30) / per_frame));
    round = std::min<uint32_t>(round, (uint32_t)std::max<size_t>(8, ((size_t)64 << 20) / ((size_t)n * 4u)));
    round = std::min<uint32_t>(round, 65535u);      // (a round's frames are the y dimension of the kernels' grids: a small group written from very many slots)
    int st;
    if ((st = xe_reserve(c, 0, (size_t)round * n * 12u)) || (st = xe_reserve(c, 1, (size_t)round * n * 8u)) || (st = xe_reserve(c, 2, (size_t)round * n * 8u)) ||
        (st = xe_reserve(c, 3, (size_t)round * n * 2u + 16u)) || (st = xe_reserve(c, 4, (size_t)round * (sizeof(GrXencHdr) + 8u)))) {
        (void)hipGetLastError(); *declined = true; return GR_OK;
    }
    int *ints = (int *)c->xe_dev[0]; unsigned long long *enc = (unsigned long long *)c->xe_dev[1]; GrXencRun *runs = (GrXencRun *)c->xe_dev[2]; uint16_t *meta = (uint16_t *)c->xe_dev[3];
    GrXencHdr *hdr_dev = (GrXencHdr *)c->xe_dev[4]; unsigned long long *off_dev = (unsigned long long *)((unsigned char *)c->xe_dev[4] + (size_t)round * sizeof(GrXencHdr));
    const size_t head_bytes = (size_t)round * (sizeof(GrXencHdr) + 8u);
    std::thread writer;                 // the round before this one, on its way into the file
    std::atomic<int> io_err(0);
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{ writer };     // (every return path waits for it)
    uint32_t ri = 0;
    for (uint32_t r0 = 0; r0 < n_frames; r0 += round, ++ri) {
        const uint32_t nf = std::min<uint32_t>(round, n_frames - r0), s0 = first_slot + r0, bank = ri & 1u;
        SlotUse use(c, s0, nf);
        // (bank `bank` was last used by round ri - 2, whose writer was joined before round ri - 1's was started)
        auto bank_reserve = [&](size_t bytes, bool keep_head) -> int {
            if (bytes <= c->xe_host_cap[bank]) return GR_OK;
            unsigned char *bigger = nullptr;
            HIPCHK(c, hipHostMalloc(&bigger, bytes, hipHostMallocDefault));
            if (keep_head && c->xe_host[bank]) memcpy(bigger, c->xe_host[bank], head_bytes);
            if (c->xe_host[bank]) (void)hipHostFree(c->xe_host[bank]);
            c->xe_host[bank] = bigger; c->xe_host_cap[bank] = bytes;
            return GR_OK;
        };
        if ((st = bank_reserve(head_bytes + (size_t)nf * n * 4u, false))) return st;      // (~4 B per atom: the usual stream; grown below when a round needs more)
        GrXencHdr *hdr = (GrXencHdr *)c->xe_host[bank]; unsigned long long *off = (unsigned long long *)(c->xe_host[bank] + (size_t)round * sizeof(GrXencHdr));
        for (uint32_t f = 0; f < nf; ++f) { GrXencHdr h = {}; for (int a = 0; a < 3; ++a) { h.mn[a] = INT_MAX; h.mx[a] = INT_MIN; } h.mindiff = (uint32_t)INT_MAX; hdr[f] = h; }
        HIPCHK(c, hipMemcpyAsync(hdr_dev, hdr, (size_t)nf * sizeof(GrXencHdr), hipMemcpyHostToDevice, c->stream));
        k_xenc_quant<<<dim3(std::min<uint32_t>((n + 255u) / 256u, 2048u), nf), dim3(256), 0, c->stream>>>(c->frames, c->frame_stride, s0, sel, n, precision, ints, hdr_dev);
        k_xenc_enc<<<dim3(std::min<uint32_t>((n + 255u) / 256u, 2048u), nf), dim3(256), 0, c->stream>>>(ints, n, hdr_dev, enc);
        k_xenc_plan<<<dim3(nf), dim3(256), 0, c->stream>>>(enc, n, hdr_dev, runs, meta, ri == 0 ? 1 : 0);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hdr, hdr_dev, (size_t)nf * sizeof(GrXencHdr), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // a first round of dense chains (k_xenc_plan declined them): nothing is in the file yet, the host encoders take the whole call
        if (ri == 0) { bool dense = false; for (uint32_t f = 0; f < nf; ++f) dense = dense || (hdr[f].flags & 2u) != 0u; if (dense) { *declined = true; return GR_OK; } }
        // frames up to the first one the format cannot hold (exactly what a loop of write_frame calls would leave in the file)
        uint32_t n_good = 0; uint32_t max_runs = 0; unsigned long long total = 0;
        for (; n_good < nf; ++n_good) {
            const GrXencHdr &h = hdr[n_good];
            bool ok = (h.flags & 1u) == 0u;
            for (int a = 0; a < 3; ++a) if ((float)h.mx[a] - (float)h.mn[a] >= (float)(INT_MAX - 2)) ok = false;
            if (!ok) break;
            off[n_good] = total;
            total += (((unsigned long long)(h.n_bits + 7u) / 8u + 3ull) & ~3ull) + 8ull;
            max_runs = std::max(max_runs, h.n_runs);
        }
        if (n_good) {
            if ((st = xe_reserve(c, 5, (size_t)total))) return st;
            unsigned char *out_dev = (unsigned char *)c->xe_dev[5];
            if ((st = bank_reserve(head_bytes + (size_t)total, true))) return st;
            hdr = (GrXencHdr *)c->xe_host[bank]; off = (unsigned long long *)(c->xe_host[bank] + (size_t)round * sizeof(GrXencHdr));
            HIPCHK(c, hipMemsetAsync(out_dev, 0, (size_t)total, c->stream));
            HIPCHK(c, hipMemcpyAsync(off_dev, off, (size_t)n_good * 8u, hipMemcpyHostToDevice, c->stream));
            k_xenc_emit<<<dim3(std::max<uint32_t>(1u, std::min<uint32_t>((max_runs + 255u) / 256u, 4096u)), n_good), dim3(256), 0, c->stream>>>(ints, n, hdr_dev, runs, meta, off_dev, out_dev);
            HIPCHK(c, hipGetLastError());
            unsigned char *streams = c->xe_host[bank] + head_bytes;
            HIPCHK(c, hipMemcpyAsync(streams, out_dev, (size_t)total, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            // the frames' 92-byte headers, then the round goes to the writer (behind the round before it: the file is written in order)
            std::vector<unsigned char> heads((size_t)n_good * 92u);
            std::vector<unsigned char> o;
            for (uint32_t f = 0; f < n_good; ++f) {
                const GrXencHdr &h = hdr[f];
                const uint32_t nbytes = (h.n_bits + 7u) / 8u, k = r0 + f;
                float m[9]; box9_rows(c->box9_set[s0 + f] ? &c->box9_host[9 * (size_t)(s0 + f)] : nullptr, m);
                o.clear();
                grx::put_be32(o, 1995u); grx::put_be32(o, n); grx::put_be32(o, (uint32_t)(int32_t)(steps ? steps[k] : 0)); grx::put_bef(o, times ? times[k] : 0.0f);
                for (int q = 0; q < 9; ++q) grx::put_bef(o, m[q]);
                grx::put_be32(o, n);
                grx::put_bef(o, precision);
                for (int a = 0; a < 3; ++a) grx::put_be32(o, (uint32_t)h.mn[a]);
                for (int a = 0; a < 3; ++a) grx::put_be32(o, (uint32_t)h.mx[a]);
                grx::put_be32(o, (uint32_t)h.smallidx0);
                grx::put_be32(o, nbytes);
                memcpy(&heads[(size_t)f * 92u], o.data(), 92u);
            }
            if (writer.joinable()) writer.join();
            if (io_err.load()) return fail(c, GR_E_IO, "short write");
            FILE *fp = w->fp;
            auto put = [fp, streams, off, hdr, n_good, &io_err](std::vector<unsigned char> hd) {
                for (uint32_t f = 0; f < n_good; ++f) {
                    const size_t padded = (((size_t)hdr[f].n_bits + 7u) / 8u + 3u) & ~(size_t)3u;      // (the stream's tail is zero: the pad bytes are already there)
                    if (fwrite(&hd[(size_t)f * 92u], 1, 92u, fp) != 92u || fwrite(streams + off[f], 1, padded, fp) != padded) { io_err.store(1); return; }
                }
            };
            bool threaded = r0 + nf < n_frames && n_good == nf;       // (the last round -- or the one that ends the output -- is written here)
            if (threaded) { try { writer = std::thread(put, std::move(heads)); } catch (const std::system_error &) { threaded = false; } }
            if (!threaded) { put(std::move(heads)); if (io_err.load()) return fail(c, GR_E_IO, "short write"); }
            c->xtc_dev_frames += n_good;
        }
        if (n_good < nf) {
            if (writer.joinable()) writer.join();
            if (io_err.load()) return fail(c, GR_E_IO, "short write");
            return fail(c, GR_E_OUT_OF_RANGE, "coordinates do not fit the xtc integers at this precision; the frames before this slot were written", s0 + n_good);
        }
    }
    if (writer.joinable()) writer.join();
    if (io_err.load()) return fail(c, GR_E_IO, "short write");
    return GR_OK;
}

int gr_xtc_write_slots(gr_xtc_writer *w, gr_ctx *c, uint32_t first_slot, uint32_t n_frames, const char *group,
                       const int64_t *steps, const float *times, float precision, int host_threads) try {
    if (!w || !w->fp || !c) return GR_E_INVALID_ARG;
    int st = slot_check(c, first_slot, n_frames); if (st) return st;
    (void)hipSetDevice(c->device);
    const Group *g = nullptr;
    std::vector<uint64_t> members;
    if (group) {
        g = find_group(c, group);
        if (!g) return fail(c, GR_E_GROUP_NOT_FOUND, group);     // WriteTrajError::GroupNotFound
        members = grc::expand(g->blocks);
    }
    const uint64_t n_out = g ? g->n : c->n;
    // large outputs are compressed on the device (gr_xtc_enc_dev.h): what crosses PCIe is the stream, and no host thread encodes
    if (c->xtc_dev_encode && n_out > 9 && n_out <= (1ull << 24) && n_out * n_frames >= 200000ull) {
        bool declined = false;
        st = xtc_write_slots_device(w, c, first_slot, n_frames, g, steps, times, precision, &declined);
        if (!declined) return st;
    }
    const size_t fb = (size_t)c->n * 3 * sizeof(float);
    // D2H of the whole batch on the compute stream into one pinned buffer (frames are ordered behind the kernels that wrote them);
    // per-frame events let the encoders start as soon as their frame has landed
    if (fb * n_frames > c->wr_cap) {
        if (c->wr_host) (void)hipHostFree(c->wr_host);
        c->wr_host = nullptr; c->wr_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->wr_host, fb * n_frames, hipHostMallocDefault));
        c->wr_cap = fb * n_frames;
    }
    float *host = c->wr_host;
    std::vector<hipEvent_t> ev(n_frames, nullptr);
    {
        SlotUse use(c, first_slot, n_frames);
        for (uint32_t k = 0; k < n_frames; ++k) {
            // slot -> packed records (k_untile) -> pinned host, in stream order: the one staging buffer is rewritten only after its copy
            hipError_t e = untile_slot(c, first_slot + k) == GR_OK ? hipSuccess : hipErrorUnknown;
            if (e == hipSuccess) e = hipMemcpyAsync(host + (size_t)k * c->n * 3, c->aos_dl, fb, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(ev[k], c->stream);
            if (e != hipSuccess) { for (auto x : ev) if (x) (void)hipEventDestroy(x); c->err = hipGetErrorString(e); return GR_E_HIP; }
        }
    }
    // encoders take frames in order; the calling thread writes frame k as soon as it is encoded, while later frames are still
    // being encoded (a frame the format cannot hold stops the output there: the frames before it are in the file, exactly as
    // a loop of write_frame calls would leave it)
    std::vector<std::vector<unsigned char>> frames_out(n_frames);
    std::vector<std::atomic<int>> done(n_frames);           // 0 pending, 1 encoded, 2 refused, 3 copy failed
    for (auto &d : done) d.store(0);
    std::atomic<uint32_t> next(0);
    std::atomic<int> stop(0);
    const int dev = c->device;
    auto work = [&]() {
        (void)hipSetDevice(dev);
        grx::EncodedFrame sc; std::vector<int> ints; std::vector<float> gathered;
        for (;;) {
            const uint32_t k = next.fetch_add(1);
            if (k >= n_frames || stop.load()) return;
            if (hipEventSynchronize(ev[k]) != hipSuccess) { done[k].store(3); return; }
            const float *src = host + (size_t)k * c->n * 3;
            if (g) {
                gathered.resize(3 * (size_t)n_out);
                for (size_t j = 0; j < members.size(); ++j) memcpy(&gathered[3 * j], src + 3 * members[j], 12);
                src = gathered.data();
            }
            float m[9]; box9_rows(c->box9_set[first_slot + k] ? &c->box9_host[9 * (size_t)(first_slot + k)] : nullptr, m);
            const bool ok = grx::serialise_frame(frames_out[k], (uint32_t)n_out, (int32_t)(steps ? steps[k] : 0), times ? times[k] : 0.0f, m, src, precision, sc, ints);
            done[k].store(ok ? 1 : 2);
        }
    };
    uint32_t nt = host_threads > 0 ? (uint32_t)host_threads : std::min<uint32_t>(n_frames, 16u);
    nt = std::max<uint32_t>(1u, std::min<uint32_t>(nt, n_frames));
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < nt; ++t) { try { th.emplace_back(work); } catch (const std::system_error &) { break; } }
    if (th.empty()) work();   // no worker could be started: encode here, then write
    int result = GR_OK; uint32_t failed_at = 0;
    for (uint32_t k = 0; k < n_frames && result == GR_OK; ++k) {
        int d;
        while ((d = done[k].load()) == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (d == 1) {
            if (fwrite(frames_out[k].data(), 1, frames_out[k].size(), w->fp) != frames_out[k].size()) { result = GR_E_IO; failed_at = k; }
            std::vector<unsigned char>().swap(frames_out[k]);
        } else { result = d == 2 ? GR_E_OUT_OF_RANGE : GR_E_HIP; failed_at = k; }
    }
    stop.store(1);
    for (auto &t : th) t.join();
    for (auto x : ev) if (x) (void)hipEventDestroy(x);
    if (result == GR_E_HIP) return fail(c, GR_E_HIP, "device-to-host copy failed while writing frames");
    if (result == GR_E_OUT_OF_RANGE) return fail(c, GR_E_OUT_OF_RANGE, "coordinates do not fit the xtc integers at this precision; the frames before this slot were written", first_slot + failed_at);
    if (result == GR_E_IO) return fail(c, GR_E_IO, "short write");
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

/* ------------------------------------------------------------ frame-sharded map-reduce: in-process pool */
gr_pool *gr_pool_create(const int *devices, int n_workers, uint64_t n_atoms, uint32_t n_slots, int *status) try {
    int dummy; if (!status) status = &dummy;
    if (!devices || n_workers <= 0 || n_workers > 1024) { *status = GR_E_INVALID_ARG; return nullptr; }
    gr_pool *p = new gr_pool();
    for (int w = 0; w < n_workers; ++w) {
        int st = GR_OK;
        gr_ctx *c = gr_ctx_create(devices[w], n_atoms, n_slots, &st);     // System::clone per worker (parallel.rs:236)
        if (!c) { *status = st; gr_pool_destroy(p); return nullptr; }
        p->ctx.push_back(c); p->device.push_back(devices[w]);
    }
    *status = GR_OK;
    return p;
} catch (...) { return nullptr; }
void gr_pool_destroy(gr_pool *p) try {
    if (!p) return;
    for (gr_ctx *c : p->ctx) gr_ctx_destroy(c);
    delete p;
} catch (...) { }
int gr_pool_size(const gr_pool *p) { return p ? (int)p->ctx.size() : 0; }
gr_ctx *gr_pool_ctx(gr_pool *p, int worker) { return (p && worker >= 0 && worker < (int)p->ctx.size()) ? p->ctx[worker] : nullptr; }
const char *gr_pool_last_error(const gr_pool *p) { return p ? p->err.c_str() : "null pool"; }

int gr_pool_map_range(gr_pool *p, uint64_t first_frame, uint64_t end_frame, uint64_t step, gr_pool_body body, void *user, size_t width, float *results,
                      gr_pool_progress progress, void *progress_user, uint64_t *frames_done, uint64_t *error_frame) try {
    if (!p || !body || (width && !results) || step == 0) return GR_E_INVALID_ARG;
    const uint64_t T = p->ctx.size();
    const uint64_t n_visit = end_frame > first_frame ? (end_frame - first_frame + step - 1) / step : 0;   // frames first, first + step, ... < end
    std::atomic<int> flag(0);                 // the shared AtomicBool of parallel.rs:230
    std::atomic<int> first_status(GR_OK);
    std::atomic<uint64_t> first_frame_err(0), done(0), last_read(first_frame);
    auto worker = [&](uint64_t w) {
        (void)hipSetDevice(p->device[w]);
        uint64_t i = 0;
        // worker w skips w * step frames, then advances by step * T (parallel.rs:425-448): ordinals k = w, w + T, ...
        for (uint64_t k = w; k < n_visit; k += T, ++i) {
            if (i % GR_POOL_ERROR_FLAG_FREQ == 0 && flag.load()) return;  // polled every 10 frames (:453-459)
            const uint64_t f = first_frame + k * step;
            int st;
            try { st = body(p->ctx[w], (int)w, f, user, width ? results + k * width : nullptr); } catch (...) { st = GR_E_HIP; }
            if (st != GR_OK) {                                            // the first error wins (:468-471)
                int expect = 0;
                if (flag.compare_exchange_strong(expect, 1)) { first_status = st; first_frame_err = f; }
                return;
            }
            const uint64_t d = done.fetch_add(1) + 1;
            uint64_t seen = last_read.load();
            while (f > seen && !last_read.compare_exchange_weak(seen, f)) { }
            if (w == 0 && progress) progress(progress_user, GR_PROGRESS_RUNNING, f, d);   // the master thread's printer (:417-422)
        }
    };
    std::vector<std::thread> th;
    for (uint64_t w = 1; w < T; ++w) { try { th.emplace_back(worker, w); } catch (const std::system_error &) { flag = 1; first_status = GR_E_HIP; break; } }
    worker(0);
    for (auto &t : th) t.join();
    if (frames_done) *frames_done = done.load();
    if (flag.load()) {
        if (error_frame) *error_frame = first_frame_err.load();
        if (progress) progress(progress_user, GR_PROGRESS_FAILED, first_frame_err.load(), done.load());
        p->err = "frame body failed at frame " + std::to_string(first_frame_err.load());
        return first_status.load();                                       // Err for the whole call (:288-321)
    }
    if (progress) progress(progress_user, GR_PROGRESS_COMPLETED, last_read.load(), done.load());   // the last frame ANY worker read (:300-317)
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_pool_map(gr_pool *p, uint64_t n_frames, gr_pool_body body, void *user, size_t width, float *results, uint64_t *frames_done, uint64_t *error_frame) {
    return gr_pool_map_range(p, 0, n_frames, 1, body, user, width, results, nullptr, nullptr, frames_done, error_frame);
}

/* ------------------------------------------------------------ frame-sharded map-reduce: one process per GPU, RCCL over xGMI */
int gr_comm_set_library(const char *path) try {
    if (!path) return GR_E_INVALID_ARG;
    // the library is resolved ONCE per process: an override that arrives after that could not take effect and is refused
    return grn::library_override_set(path) ? GR_OK : GR_E_INVALID_ARG;
} catch (...) { return gr_abi_guard(); }

int gr_comm_unique_id(void *id128) try {
    if (!id128) return GR_E_INVALID_ARG;
    grn::Api &a = grn::api();
    if (!a.ok) return GR_E_NO_DEVICE;
    grn::unique_id id;
    if (a.GetUniqueId(&id) != grn::Success) return GR_E_HIP;
    memcpy(id128, &id, sizeof id);
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

gr_comm *gr_comm_create(int device, int rank, int world, const void *id128, int *status) try {
    int dummy; if (!status) status = &dummy;
    if (!id128 || world < 1 || rank < 0 || rank >= world) { *status = GR_E_INVALID_ARG; return nullptr; }
    grn::Api &a = grn::api();
    if (!a.ok) { *status = GR_E_NO_DEVICE; return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { *status = GR_E_NO_DEVICE; return nullptr; }
    gr_comm *c = new gr_comm();
    c->device = device; c->rank = rank; c->world = world;
    grn::unique_id id; memcpy(&id, id128, sizeof id);
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess && hipMalloc(&c->flag, 2 * sizeof(int)) == hipSuccess;
    if (ok) ok = a.CommInitRank(&c->comm, world, id, rank) == grn::Success;
    if (!ok) { *status = GR_E_HIP; gr_comm_destroy(c); return nullptr; }
    *status = GR_OK;
    return c;
} catch (...) { return nullptr; }

void gr_comm_destroy(gr_comm *c) try {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)grn::api().CommDestroy(c->comm);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    if (c->flag) (void)hipFree(c->flag);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
} catch (...) { }
const char *gr_comm_last_error(const gr_comm *c) { return c ? c->err.c_str() : "null communicator"; }
const char *gr_comm_library(void) { return grn::api().source.c_str(); }

#define COMMCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (c)->err = std::string(#call) + ": " + hipGetErrorString(e_); return GR_E_HIP; } } while (0)
int gr_comm_gather_per_frame(gr_comm *c, const float *local, uint64_t n_total, size_t width, float *out) try {
    if (!c || !out || width == 0) return GR_E_INVALID_ARG;
    (void)hipSetDevice(c->device);
    const uint64_t G = (uint64_t)c->world, per = (n_total + G - 1) / G;                        // ceil(F / G) rows from every rank
    const uint64_t mine = n_total > (uint64_t)c->rank ? (n_total - c->rank + G - 1) / G : 0;  // frames rank, rank + G, ...
    if (mine && !local) return GR_E_INVALID_ARG;
    const size_t row = per * width;
    if (row > c->cap) {
        if (c->send) (void)hipFree(c->send);
        if (c->recv) (void)hipFree(c->recv);
        c->send = c->recv = nullptr; c->cap = 0;
        COMMCHK(c, hipMalloc(&c->send, (row ? row : 1) * sizeof(float)));
        COMMCHK(c, hipMalloc(&c->recv, (row ? row : 1) * G * sizeof(float)));
        c->cap = row;
    }
    if (row == 0) return GR_OK;
    COMMCHK(c, hipMemsetAsync(c->send, 0, row * sizeof(float), c->stream));
    if (mine) COMMCHK(c, hipMemcpyAsync(c->send, local, mine * width * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const int rc = grn::api().AllGather(c->send, c->recv, row, grn::Float32, c->comm, c->stream);      // the run's one collective
    if (rc != grn::Success) { c->err = std::string("ncclAllGather: ") + (grn::api().GetErrorString ? grn::api().GetErrorString(rc) : "failed"); return GR_E_HIP; }
    std::vector<float> all(row * G);
    COMMCHK(c, hipMemcpyAsync(all.data(), c->recv, all.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    COMMCHK(c, hipStreamSynchronize(c->stream));
    gr_deinterleave(all.data(), c->world, per, width, n_total, out);
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_comm_any_error(gr_comm *c, int local_flag, int *any) try {
    if (!c || !any) return GR_E_INVALID_ARG;
    (void)hipSetDevice(c->device);
    const int v = local_flag ? 1 : 0;
    COMMCHK(c, hipMemcpyAsync(c->flag, &v, sizeof(int), hipMemcpyHostToDevice, c->stream));
    const int rc = grn::api().AllReduce(c->flag, c->flag + 1, 1, grn::Int32, grn::Max, c->comm, c->stream);
    if (rc != grn::Success) { c->err = "ncclAllReduce failed"; return GR_E_HIP; }
    int r = 0;
    COMMCHK(c, hipMemcpyAsync(&r, c->flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    COMMCHK(c, hipStreamSynchronize(c->stream));
    *any = r;
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

void gr_shard_deinterleave(const float *gathered, int world, uint64_t n_total, size_t width, float *out) try {
    if (!gathered || !out || world < 1) return;
    gr_deinterleave(gathered, world, (n_total + (uint64_t)world - 1) / (uint64_t)world, width, n_total, out);
} catch (...) { }

/* ------------------------------------------------------------ measurement / synthetic data */
int gr_timer_start(gr_ctx *c) { if (!c) return GR_E_INVALID_ARG; HIPCHK(c, hipEventRecord(c->ev0, c->stream)); return GR_OK; }
int gr_timer_stop(gr_ctx *c, float *ms) try {
    if (!c) return GR_E_INVALID_ARG;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float t = 0.f;
    HIPCHK(c, hipEventElapsedTime(&t, c->ev0, c->ev1));
    if (ms) *ms = t;
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_profile_enable(gr_ctx *c, int on) try {
    if (!c) return GR_E_INVALID_ARG;
    c->profile = on ? 1 : 0;
    for (int k = 0; k < 4; ++k) { c->prof_ms[k] = 0; c->prof_launches[k] = 0; c->prof_frames[k] = 0; }
    return GR_OK;
} catch (...) { return gr_abi_guard(); }
int gr_profile_read(const gr_ctx *c, int kernel, double *ms_total, uint64_t *launches, uint64_t *frames) try {
    if (!c || kernel < 0 || kernel > 3) return GR_E_INVALID_ARG;
    if (ms_total) *ms_total = c->prof_ms[kernel];
    if (launches) *launches = c->prof_launches[kernel];
    if (frames) *frames = c->prof_frames[kernel];
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_synth_reference(gr_ctx *c, uint32_t slot, const float *box9, float radius, uint64_t seed) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    HIPCHK(c, sync_ingest(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    st = set_box(c, slot, box9); if (st) return st;
    st = box_check(c, slot); if (st) return st;
    k_synth_reference<<<dim3((uint32_t)((c->n + 255) / 256)), dim3(256), 0, c->stream>>>(c->frames + (size_t)slot * c->frame_stride, (uint32_t)c->n, c->boxes_dev + slot, radius, seed);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_synth_frames(gr_ctx *c, uint32_t ref_slot, uint32_t first_slot, uint32_t n_frames, uint64_t first_frame_index,
                    uint64_t frame_index_stride, float sigma, uint64_t seed) try {
    int st = slot_check(c, ref_slot); if (st) return st;
    st = slot_check(c, first_slot, n_frames); if (st) return st;
    if (ref_slot >= first_slot && ref_slot < first_slot + n_frames) return fail(c, GR_E_INVALID_ARG, "reference slot inside the output range");
    (void)hipSetDevice(c->device);
    st = box_check(c, ref_slot); if (st) return st;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, sync_ingest(c));
    // every generated frame carries the reference's box: the table is built ONCE, replicated on the host, and the whole range goes
    // over in ONE copy.  (Rounds 1-3 queued one 444-byte hipMemcpyAsync per slot -- 9216 back-to-back blit dispatches ahead of the
    // generator kernels for the 3072-frames-per-step shape; under `rocprofv3 --pmc`, which brackets every dispatch with a counter
    // read-out, that queue is where the profiled runs stopped: DESIGN.md "The --pmc stall".)
    st = set_boxes_same(c, first_slot, n_frames, c->box9_set[ref_slot] ? &c->box9_host[9 * (size_t)ref_slot] : nullptr); if (st) return st;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += 1024) {
        const uint32_t nf = std::min<uint32_t>(1024, n_frames - f0);
        k_synth_frames<<<dim3((uint32_t)((c->n + 255) / 256), nf), dim3(256), 0, c->stream>>>(c->frames + (size_t)ref_slot * c->frame_stride, c->frames, c->frame_stride, first_slot + f0, (uint32_t)c->n, c->boxes_dev + ref_slot, first_frame_index + (uint64_t)f0 * frame_index_stride, frame_index_stride, sigma, seed);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

int gr_synth_uniform(gr_ctx *c, uint32_t slot, const float *box9, uint64_t seed) try {
    int st = slot_check(c, slot); if (st) return st;
    (void)hipSetDevice(c->device);
    HIPCHK(c, sync_ingest(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    st = set_box(c, slot, box9); if (st) return st;
    st = box_check(c, slot); if (st) return st;
    k_synth_uniform<<<dim3((uint32_t)((c->n + 255) / 256)), dim3(256), 0, c->stream>>>(c->frames + (size_t)slot * c->frame_stride, (uint32_t)c->n, c->boxes_dev + slot, seed);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GR_OK;
} catch (...) { return gr_abi_guard(); }

}  // extern "C"
