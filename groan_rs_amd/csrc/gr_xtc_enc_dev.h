// gr_xtc_enc_dev.h -- xtc coordinate COMPRESSION on the device (fitted-trajectory output, XtcWriter::write_frame of the reference,
// src/io/xtc_io/mod.rs:256-331 over xdrfile's write_xtc).
//
// The host encoder (gr_xtc.h::encode_coords) costs 10.8 ms per 5e5-atom frame and thread: behind a fit pass that turns out 200 000
// frames a second, writing the result was the slowest stage of read -> fit -> write by two orders of magnitude, and what crossed PCIe
// was the 12 B/atom of the positions.  The format is sequential only in its DECISIONS -- where a run of small differences ends, when
// the small-range index moves -- and those depend on nothing but distances between neighbouring atoms; the arithmetic (quantising,
// the mixed-radix packing, the bit stream) is independent per run once its start is known.  So, per batch of frames:
//
//   k_xenc_quant   every atom: float -> the format's integers (bit for bit the host's quantise()), per-frame minima / maxima / the
//                  smallest L1 step (atomics)
//   k_xenc_enc     every atom j: what a run STARTING at j would be -- its length and whether it lets the small-range index move --
//                  for each of the nine values that index can take in the frame: one 64-bit word per atom (see the kernel)
//   k_xenc_plan    one workgroup per frame walks the frame's runs -- one word and a few bit operations per run, 256 segments side by
//                  side between atoms where a run must start whatever came before (see the kernel) -- and writes one descriptor
//                  per run: first atom, bit offset, run length, the +-1 step of the small-range index, whether the flag bits announce it
//   k_xenc_emit    one lane per run: the run's big integer (mixed radix, up to 72 bits), flag bits and small triples, OR-ed into
//                  the zeroed stream at the run's bit offset
//
// and the host adds the 56-byte header per frame.  Byte-identical to the host encoder (tests/test_gpu_xtc_writer.py), which is
// byte-identical to the reference's writer.  A frame whose coordinates the format cannot hold is flagged and refused exactly as the
// host path refuses it.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include "gr_xtc_dev.h"      // gr_xtc_magic
#include "gr_kernels.h"      // GrSel, gr_pos_load

#define GR_XENC_FIRSTIDX 9
#define GR_XENC_LASTIDX 73
#define GR_XENC_NODIST 0xFFFFFFFFu     /* distance to an atom that does not exist: never "near" */

// per frame: what the header needs and what the plan pass leaves for the host
struct GrXencHdr {
    int mn[3], mx[3];
    uint32_t mindiff;          // smallest L1 step between consecutive atoms, clamped to INT_MAX
    uint32_t flags;            // bit 0: some coordinate does not fit the format's integers (the frame is refused); bit 1: k_xenc_plan declined the frame (one dense chain)
    int smallidx0;             // the small-range index the frame starts with (header field)
    uint32_t n_runs;
    uint32_t n_bits;           // length of the bit stream
    uint32_t pad;
};
struct GrXencRun { uint32_t atom0, bitpos; };      // + one meta half-word per run: n_small | (is_smaller + 1) << 4 | flag << 6 | smallidx << 8

__device__ __forceinline__ bool gr_xenc_quantise(float x, float precision, int &q) {      // gr_xtc.h::quantise
    const float prod = x * precision;
    const float lf = (float)((double)prod + (x >= 0.0f ? 0.5 : -0.5));
    q = (int)lf;
    return fabsf(lf) <= (float)(INT_MAX - 2);
}

// ints[frame][j] = the quantised atom j of the OUTPUT order (the whole system, or the group's members)
__global__ __launch_bounds__(256) void k_xenc_quant(const float *__restrict__ frames, size_t frame_stride, uint32_t first_slot, GrSel sel, uint32_t n,
                                                    float precision, int *__restrict__ ints, GrXencHdr *__restrict__ hdr) {
    const uint32_t frame = blockIdx.y;
    const float *xyz = frames + (size_t)(first_slot + frame) * frame_stride;
    int *qi = ints + (size_t)frame * n * 3;
    __shared__ int red[6][4];
    __shared__ uint32_t redu[2][4];
    int mn[3] = { INT_MAX, INT_MAX, INT_MAX }, mx[3] = { INT_MIN, INT_MIN, INT_MIN };
    uint32_t mind = (uint32_t)INT_MAX, bad = 0u;
    auto atom = [&](uint32_t j, int (&q)[3]) {
        const uint32_t a = sel.contiguous ? sel.start + j : sel.idx[j];
        float x, y, z;
        gr_pos_load(xyz, a, x, y, z);
        if (x != x) { x = 0.0f; y = 0.0f; z = 0.0f; }       // None travels as NaN in x: written as the origin (xtc_io/mod.rs:296-301)
        bool ok = gr_xenc_quantise(x, precision, q[0]);
        ok = gr_xenc_quantise(y, precision, q[1]) && ok;
        ok = gr_xenc_quantise(z, precision, q[2]) && ok;
        if (!ok) bad = 1u;
    };
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n; j += gridDim.x * 256u) {
        int q[3], p[3];
        atom(j, q);
        for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], q[a]); mx[a] = max(mx[a], q[a]); }
        if (j > 0) {
            atom(j - 1, p);
            const unsigned long long d = (unsigned long long)llabs((long long)p[0] - q[0]) + (unsigned long long)llabs((long long)p[1] - q[1]) + (unsigned long long)llabs((long long)p[2] - q[2]);
            mind = min(mind, (uint32_t)min(d, (unsigned long long)INT_MAX));
        }
        qi[3 * (size_t)j] = q[0]; qi[3 * (size_t)j + 1] = q[1]; qi[3 * (size_t)j + 2] = q[2];
    }
    // block -> frame
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) { mn[a] = min(mn[a], __shfl_xor(mn[a], off, 64)); mx[a] = max(mx[a], __shfl_xor(mx[a], off, 64)); }
    }
    for (int off = 32; off > 0; off >>= 1) { mind = min(mind, (uint32_t)__shfl_xor((int)mind, off, 64)); bad |= (uint32_t)__shfl_xor((int)bad, off, 64); }
    if (lane == 0) { for (int a = 0; a < 3; ++a) { red[a][wave] = mn[a]; red[3 + a][wave] = mx[a]; } redu[0][wave] = mind; redu[1][wave] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        GrXencHdr &h = hdr[frame];
        for (int a = 0; a < 3; ++a) {
            atomicMin(&h.mn[a], min(min(red[a][0], red[a][1]), min(red[a][2], red[a][3])));
            atomicMax(&h.mx[a], max(max(red[3 + a][0], red[3 + a][1]), max(red[3 + a][2], red[3 + a][3])));
        }
        atomicMin(&h.mindiff, min(min(redu[0][0], redu[0][1]), min(redu[0][2], redu[0][3])));
        if (redu[1][0] | redu[1][1] | redu[1][2] | redu[1][3]) atomicOr(&h.flags, 1u);
    }
}

// the frame's constants as both later passes need them
struct GrXencConst {
    uint32_t sizeint[3]; int bitsizeint[3]; int bitsize, big_bits;
    int smallidx0, maxidx, minidx, larger;
    bool ok;
};
__device__ __forceinline__ int gr_xenc_bitlen64(unsigned long long v) { return v ? 64 - __builtin_clzll(v) : 0; }
__device__ __forceinline__ GrXencConst gr_xenc_const(const GrXencHdr &h) {
    GrXencConst c;
    c.ok = (h.flags & 1u) == 0u;
    for (int a = 0; a < 3; ++a) {
        if ((float)h.mx[a] - (float)h.mn[a] >= (float)(INT_MAX - 2)) c.ok = false;      // value - minint would not fit
        c.sizeint[a] = (uint32_t)h.mx[a] - (uint32_t)h.mn[a] + 1u;
        c.bitsizeint[a] = 0;
    }
    if ((c.sizeint[0] | c.sizeint[1] | c.sizeint[2]) > 0xffffffu) {
        for (int a = 0; a < 3; ++a) { const int b = gr_xenc_bitlen64(c.sizeint[a]); c.bitsizeint[a] = b > 32 ? 32 : b; }
        c.bitsize = 0;
        c.big_bits = c.bitsizeint[0] + c.bitsizeint[1] + c.bitsizeint[2];
    } else {
        const unsigned long long p = (unsigned long long)c.sizeint[0] * c.sizeint[1];              // < 2^48
        const unsigned long long lo = p * c.sizeint[2], hi = __umul64hi(p, (unsigned long long)c.sizeint[2]);
        c.bitsize = hi ? 64 + gr_xenc_bitlen64(hi) : gr_xenc_bitlen64(lo);
        c.big_bits = c.bitsize;
    }
    int smallidx = GR_XENC_FIRSTIDX;
    const long long mindiff = (long long)h.mindiff;
    while (smallidx < GR_XENC_LASTIDX && gr_xtc_magic[smallidx < GR_XENC_LASTIDX ? smallidx : GR_XENC_LASTIDX - 1] < mindiff) ++smallidx;
    if (smallidx > GR_XENC_LASTIDX - 1) smallidx = GR_XENC_LASTIDX - 1;
    c.smallidx0 = smallidx;
    c.maxidx = min(GR_XENC_LASTIDX - 1, smallidx + 8);
    c.minidx = c.maxidx - 8;
    c.larger = gr_xtc_magic[c.maxidx] / 2;
    return c;
}

// What a run that starts at atom j would look like, for every value the small-range index can take in this frame (it stays within
// [minidx, minidx + 8]): one 64-bit word per atom, so that the sequential walk below is ONE load and a few bit operations per run.
// With the format's "water swap" the chain previous -> current inside a run goes j + 1, j, j + 2, j + 3, ..., so small atom 0 is atom
// j against j + 1, small atom 1 is atom j + 2 against j, small atom t >= 2 is atom j + 1 + t against j + t:
//   bits 5 r .. 5 r + 3   k(r): how many small atoms (0 .. 8) follow the run's big atom when the index is minidx + r (every one of
//                         them within smallnum(r) of its predecessor, max norm)
//   bit  5 r + 4          some of those k steps has a (32-bit wrapped) square length >= smaller(r)^2: the index may not go down
//   bit 45 / 46           the atom written before atom j -- atom j - 1, or atom j - 2 when the run before held exactly one small atom
//                         (the swap wrote j - 1 first) -- lies within `larger` of atom j: the index may go up
__global__ __launch_bounds__(256) void k_xenc_enc(const int *__restrict__ ints, uint32_t n, const GrXencHdr *__restrict__ hdr, unsigned long long *__restrict__ enc) {
    const uint32_t frame = blockIdx.y;
    const GrXencHdr &H = hdr[frame];
    const GrXencConst C = gr_xenc_const(H);
    if (!C.ok) return;
    const int *Q = ints + (size_t)frame * n * 3;
    unsigned long long *E = enc + (size_t)frame * n;
    uint32_t sn[9]; int lim[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int s = C.minidx + r;
        sn[r] = (uint32_t)(gr_xtc_magic[s] / 2);
        const uint32_t sm = (uint32_t)(gr_xtc_magic[max(GR_XENC_FIRSTIDX, s - 1)] / 2);
        lim[r] = (int)(sm * sm);
    }
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n; j += gridDim.x * 256u) {
        // atoms j - 2 .. j + 9 (those that exist)
        int q[12][3];
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            const long long a = (long long)j + t - 2;
            const bool have = a >= 0 && a < (long long)n;
            const size_t b = have ? 3 * (size_t)a : 0;
            q[t][0] = Q[b]; q[t][1] = Q[b + 1]; q[t][2] = Q[b + 2];
        }
        auto have = [&](int t) { const long long a = (long long)j + t - 2; return a >= 0 && a < (long long)n; };
        auto dist = [&](int ta, int tb, uint32_t &dmax, uint32_t &usum) {   // |difference| per axis in 64 bits, square sum in wrapping 32 bits
            unsigned long long m = 0; uint32_t u = 0u;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const long long d = (long long)q[ta][a] - (long long)q[tb][a];
                m = max(m, (unsigned long long)llabs(d));
                const uint32_t w = (uint32_t)(int)d;
                u += w * w;
            }
            dmax = (have(ta) && have(tb)) ? (uint32_t)min(m, (unsigned long long)0xFFFFFFFEu) : GR_XENC_NODIST;
            usum = u;
        };
        uint32_t e[8], u[8];
        dist(2, 3, e[0], u[0]);                    // small 0: atom j against j + 1
        dist(4, 2, e[1], u[1]);                    // small 1: atom j + 2 against j
#pragma unroll
        for (int t = 2; t < 8; ++t) dist(3 + t, 2 + t, e[t], u[t]);      // small t: atom j + 1 + t against j + t
        uint32_t dA, dB, unused;
        dist(2, 1, dA, unused);                    // atom j against j - 1
        dist(2, 0, dB, unused);                    // atom j against j - 2
        unsigned long long word = 0ull;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            uint32_t k = 0; bool open = true, big = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                open = open && e[t] < sn[r];
                if (open) { ++k; big = big || ((int)u[t] >= lim[r]); }
            }
            word |= (unsigned long long)(k | (big ? 16u : 0u)) << (5 * r);
        }
        if (dA < (uint32_t)C.larger) word |= 1ull << 45;
        if (dB < (uint32_t)C.larger) word |= 1ull << 46;
        E[j] = word;
    }
}

// One workgroup per frame walks the runs: xdrfile's decision procedure (gr_xtc.h::encode_coords) over the words of k_xenc_enc.
//
// The walk is a pointer chase -- run r + 1 starts where run r ends, and the small-range index it uses is the one run r left -- and
// as ONE chain it costs 0.6 us per run: 100 ms for a water-like frame of 5e5 atoms (the first version of this kernel).  But the chain
// re-synchronises: an atom j whose word has bits 45 and 46 clear is farther than `larger` (>= every smallnum) from atom j - 1 and from
// atom j - 2, so no run that starts before j can hold it -- not as a small atom (it would have to be near its predecessor in the chain,
// j - 1 or, right after the swap, j - 2) and not as the swapped big atom of a run starting at j - 1 -- and a run STARTS at j whatever
// happened before.  What is not known there is the index (nine values) and the length of the run before (for the flag bit:
// flag_r = run_r != run_(r-1) || index moves -- the format's `prevrun` always equals the previous run's length).  So:
//   1. thread t looks for the first such atom in its 1/256th of the frame (its anchor; thread 0: atom 0)
//   2. the segment from an anchor to the next one is walked once for EACH of the nine possible entry indices (by the anchor's thread
//      and the anchorless threads behind it, sharing the nine out), keeping per entry index:
//      runs, bits (without the first run's flag field), the index at the exit, the first run's length and index step, the last run's length
//   3. thread 0 threads the true entry states through the 256 x 9 table (a sequential scan of 256 look-ups)
//   4. every thread walks its segment once more with its true entry state and writes the run descriptors
// Ten walks of 1/256th of the frame each, side by side: measured see DESIGN.md.  A frame without such atoms (one dense chain) is walked
// by thread 0 alone, as before.
struct GrXencSeg { uint32_t runs, bits; uint16_t packed; };     // packed: exit index (4) | first k (4) << 4 | (first step + 1) << 8 | last k << 10
__global__ __launch_bounds__(256) void k_xenc_plan(const unsigned long long *__restrict__ enc, uint32_t n, GrXencHdr *__restrict__ hdr,
                                                   GrXencRun *__restrict__ runs, uint16_t *__restrict__ meta, int decline_dense) {
    const uint32_t frame = blockIdx.x, t = threadIdx.x;
    const unsigned long long *E = enc + (size_t)frame * n;
    GrXencRun *R = runs + (size_t)frame * n;
    uint16_t *M = meta + (size_t)frame * n;
    GrXencHdr &H = hdr[frame];
    const GrXencConst C = gr_xenc_const(H);
    if (!C.ok) { if (t == 0) { H.smallidx0 = C.smallidx0; H.n_runs = 0; H.n_bits = 0; } return; }
    constexpr uint32_t NONE = 0xFFFFFFFFu, T = 256;
    __shared__ uint32_t anchor[T];
    __shared__ GrXencSeg tab[T][9];
    __shared__ uint32_t ent_s[T], ent_k[T], ent_run[T], ent_bit[T];
    // 1. anchors
    const uint32_t S = (n + T - 1u) / T;
    uint32_t a = NONE;
    if (t == 0) a = 0u;
    else {
        const uint32_t j1 = min(n, (t + 1u) * S);
        for (uint32_t j = t * S; j < j1; ++j) if (((E[j] >> 45) & 3ull) == 0ull) { a = j; break; }
    }
    anchor[t] = a;
    __syncthreads();
    // a thread without an anchor of its own helps the last thread before it that has one (its OWNER): the nine summary walks of a
    // long segment -- a protein is one run after another with no place where a run must start -- are shared out over the owner and the
    // threads behind it instead of all falling to one
    uint32_t owner = t;
    while (anchor[owner] == NONE) --owner;                            // (thread 0 always has one)
    uint32_t nxt = owner + 1u;
    while (nxt < T && anchor[nxt] == NONE) ++nxt;
    const uint32_t group = nxt - owner, rank = t - owner;
    const uint32_t seg_a = anchor[owner], seg_e = nxt < T ? anchor[nxt] : n;
    // A segment is walked by ONE thread (nine times over for the summaries, shared out over nine threads at most): a frame whose atoms
    // are one dense chain -- a protein-only output: no atom farther than `larger` from its predecessor, so no place where a run must
    // start -- is a single walk of ~0.6 us per run, several ms per 1e5 atoms, where 16 host encoder threads need ~1.  With `decline_dense`
    // (the host sets it for a call's first round, before anything is in the file) such a frame is flagged (bit 1) instead of walked,
    // and the host encoders take the call.
    if (decline_dense && n >= 32768u) {
        __shared__ uint32_t longest;
        if (t == 0) longest = 0u;
        __syncthreads();
        if (t == owner) atomicMax(&longest, seg_e - seg_a);
        __syncthreads();
        if (longest > n / 2u) { if (t == 0) { H.flags |= 2u; H.smallidx0 = C.smallidx0; H.n_runs = 0; H.n_bits = 0; } return; }
    }
    // one walk of [seg_a, seg_e): `s` = entry index - minidx, `pk` = the previous run's length (255: none); EMIT: descriptors out
    auto walk = [&](uint32_t s, uint32_t pk, bool emit, uint32_t run0, uint32_t bit0, GrXencSeg &out) {
        const uint32_t e = seg_e;
        int smallidx = C.minidx + (int)s;
        uint32_t i = seg_a, r = 0, bits = 0, first_k = 0, last_k = 0; int first_step = 0;
        bool last_k1 = false;
        while (i < e) {
            const unsigned long long word = E[i];
            const uint32_t f = (uint32_t)(word >> (5 * (smallidx - C.minidx))) & 31u, k = f & 15u;
            const bool near = ((word >> (last_k1 ? 46 : 45)) & 1ull) != 0ull;
            int step = 0;
            if (smallidx < C.maxidx && near) step = 1;
            else if (smallidx > C.minidx) step = -1;
            if (step == -1 && (k == 0u || (f & 16u))) step = 0;
            if (r == 0) { first_k = k; first_step = step; }
            const bool flag = (r == 0 && !emit) ? false : (k != pk || step != 0);      // (the first run's flag of a summary walk is settled by the scan)
            if (emit) {
                R[run0 + r].atom0 = i; R[run0 + r].bitpos = bit0 + bits;
                M[run0 + r] = (uint16_t)(k | ((uint32_t)(step + 1) << 4) | ((flag ? 1u : 0u) << 6) | ((uint32_t)smallidx << 8));
            }
            bits += (uint32_t)C.big_bits + k * (uint32_t)smallidx + ((r == 0 && !emit) ? 0u : (flag ? 6u : 1u));
            ++r;
            i += 1u + k;
            last_k1 = k == 1u; last_k = k; pk = k;
            smallidx += step;
        }
        out.runs = r; out.bits = bits;
        out.packed = (uint16_t)((uint32_t)(smallidx - C.minidx) | (first_k << 4) | ((uint32_t)(first_step + 1) << 8) | (last_k << 10));
    };
    // 2. summaries for every entry index (thread 0 knows its own)
    const uint32_t s0 = (uint32_t)(C.smallidx0 - C.minidx);
    // (a frame with no anchor but atom 0 -- one dense chain -- has nothing to summarise: thread 0 writes the descriptors in one walk)
    const bool solo = owner == 0u && nxt == T;
    if (owner == 0u) { if (t == 0u && !solo) walk(s0, 255u, false, 0u, 0u, tab[0][s0]); }
    else for (uint32_t s = rank; s < 9u; s += group) walk(s, 255u, false, 0u, 0u, tab[owner][s]);
    __syncthreads();
    // 3. the true entry states
    if (t == 0 && solo) { ent_s[0] = s0; ent_k[0] = 255u; ent_run[0] = 0u; ent_bit[0] = 0u; }
    if (t == 0 && !solo) {
        uint32_t s = s0, pk = 255u, run_base = 0u, bit_base = 0u;
        for (uint32_t u = 0; u < T; ++u) {
            if (anchor[u] == NONE) continue;
            ent_s[u] = s; ent_k[u] = pk; ent_run[u] = run_base; ent_bit[u] = bit_base;
            const GrXencSeg sm = tab[u][s];
            const uint32_t fk = (sm.packed >> 4) & 15u, lk = (sm.packed >> 10) & 15u; const int fs = (int)((sm.packed >> 8) & 3u) - 1;
            bit_base += sm.bits + ((fk != pk || fs != 0) ? 6u : 1u);
            run_base += sm.runs;
            pk = lk; s = sm.packed & 15u;
        }
        H.smallidx0 = C.smallidx0; H.n_runs = run_base; H.n_bits = bit_base;
    }
    __syncthreads();
    // 4. descriptors
    if (rank == 0u) {
        GrXencSeg done;
        walk(ent_s[t], ent_k[t], true, ent_run[t], ent_bit[t], done);
        if (solo) { H.smallidx0 = C.smallidx0; H.n_runs = done.runs; H.n_bits = done.bits; }      // (t == 0: every other thread has owner 0 and a rank > 0)
    }
}

// MSB-first bit stream as big-endian 32-bit words, OR-ed into zeroed memory
struct GrXencBits {
    uint32_t *w; uint32_t pos;
    __device__ __forceinline__ void put(int nbits, uint32_t v) {            // 0 <= nbits <= 32, v < 2^nbits
        if (nbits == 0) return;
        const uint32_t k = pos >> 5, o = pos & 31u;
        const unsigned long long sh = (unsigned long long)v << (64u - o - (uint32_t)nbits);
        const uint32_t hi = (uint32_t)(sh >> 32), lo = (uint32_t)sh;
        if (hi) atomicOr(w + k, __builtin_bswap32(hi));
        if (lo) atomicOr(w + k + 1, __builtin_bswap32(lo));
        pos += (uint32_t)nbits;
    }
    // a packed integer of `nbits` bits: its bytes go out least significant first, the last (partial) chunk holds the top bits
    __device__ __forceinline__ void put_packed64(int nbits, unsigned long long v) {      // nbits <= 64
        const int m = nbits >> 3, rest = nbits & 7;
        if (m > 0) {
            const unsigned long long sw = __builtin_bswap64(v) >> (64 - 8 * m);
            if (m > 4) { put(8 * (m - 4), (uint32_t)(sw >> 32)); put(32, (uint32_t)sw); }
            else put(8 * m, (uint32_t)sw);
        }
        if (rest) put(rest, (uint32_t)((m < 8 ? v >> (8 * m) : 0ull) & ((1u << rest) - 1u)));
    }
    __device__ __forceinline__ void put_packed128(int nbits, unsigned long long lo, unsigned long long hi) {
        if (nbits <= 64) { put_packed64(nbits, lo); return; }
        put_packed64(64, lo);
        put_packed64(nbits - 64, hi);
    }
};
// (a * s1 + b) * s2 + c in 128 bits: a, b, c < 2^32 (< 2^24 wherever the product is used), s1, s2 < 2^25
__device__ __forceinline__ void gr_xenc_pack3(uint32_t a, uint32_t b, uint32_t c, uint32_t s1, uint32_t s2, unsigned long long &lo, unsigned long long &hi) {
    const unsigned long long t = (unsigned long long)a * s1 + b;
    lo = t * s2; hi = __umul64hi(t, (unsigned long long)s2);
    const unsigned long long l2 = lo + c;
    if (l2 < lo) ++hi;
    lo = l2;
}

// one lane per run; `out` = the batch's streams back to back, frame f at byte offset out_off[f] (a multiple of 4, zeroed, padded)
__global__ __launch_bounds__(256) void k_xenc_emit(const int *__restrict__ ints, uint32_t n, const GrXencHdr *__restrict__ hdr, const GrXencRun *__restrict__ runs,
                                                   const uint16_t *__restrict__ meta, const unsigned long long *__restrict__ out_off, unsigned char *__restrict__ out) {
    const uint32_t frame = blockIdx.y;
    const GrXencHdr &H = hdr[frame];
    const uint32_t n_runs = H.n_runs;
    if (blockIdx.x * 256u >= n_runs) return;
    const GrXencConst C = gr_xenc_const(H);
    const int *Q = ints + (size_t)frame * n * 3;
    const GrXencRun *R = runs + (size_t)frame * n;
    const uint16_t *M = meta + (size_t)frame * n;
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < n_runs; r += gridDim.x * 256u) {
        const GrXencRun run = R[r];
        const uint32_t m = M[r], k = m & 15u, flag = (m >> 6) & 1u;
        const int is_smaller = (int)((m >> 4) & 3u) - 1, smallidx = (int)(m >> 8);
        GrXencBits B; B.w = reinterpret_cast<uint32_t *>(out + out_off[frame]); B.pos = run.bitpos;
        const uint32_t ib = run.atom0 + (k ? 1u : 0u);                  // the swap: the big atom is the run's second one
        int prev[3] = { Q[3 * (size_t)ib], Q[3 * (size_t)ib + 1], Q[3 * (size_t)ib + 2] };
        const uint32_t v0 = (uint32_t)(prev[0] - H.mn[0]), v1 = (uint32_t)(prev[1] - H.mn[1]), v2 = (uint32_t)(prev[2] - H.mn[2]);
        if (C.bitsize == 0) { B.put(C.bitsizeint[0], v0); B.put(C.bitsizeint[1], v1); B.put(C.bitsizeint[2], v2); }
        else { unsigned long long lo, hi; gr_xenc_pack3(v0, v1, v2, C.sizeint[1], C.sizeint[2], lo, hi); B.put_packed128(C.bitsize, lo, hi); }
        if (flag) { B.put(1, 1u); B.put(5, (uint32_t)(3 * (int)k + is_smaller + 1)); } else B.put(1, 0u);
        const int smallnum = gr_xtc_magic[smallidx] / 2;
        const uint32_t sizesmall = (uint32_t)gr_xtc_magic[smallidx];
        for (uint32_t s = 0; s < k; ++s) {
            const uint32_t ia = s == 0 ? run.atom0 : run.atom0 + 1u + s;
            const int cur[3] = { Q[3 * (size_t)ia], Q[3 * (size_t)ia + 1], Q[3 * (size_t)ia + 2] };
            const uint32_t a = (uint32_t)(cur[0] - prev[0] + smallnum), b = (uint32_t)(cur[1] - prev[1] + smallnum), c = (uint32_t)(cur[2] - prev[2] + smallnum);
            unsigned long long lo, hi; gr_xenc_pack3(a, b, c, sizesmall, sizesmall, lo, hi);
            B.put_packed128(smallidx, lo, hi);
            prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
        }
    }
}
