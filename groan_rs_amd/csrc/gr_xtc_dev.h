// gr_xtc_dev.h -- xtc coordinate unpacking on the device.
//
// The xtc bit stream (GROMACS xdrfile `xdr3dfcoord`; the reference reads it through the molly crate or the vendored C
// xdrfile, src/io/xtc_io/*) is sequential only in its FRAMING: where a group of atoms starts depends on the flag and
// run-length fields of the groups before it.  The arithmetic -- recovering three mixed-radix digits from one packed
// integer (two divisions), the delta chain inside a run, int -> float -- is independent per group once the start state
// (bit position, small-range index, pending run length) is known.  So the host skims the framing (gr_xtc.h::skim_frame,
// ~10x cheaper than decoding) and records that state every 32 atoms; here one lane unpacks one 32-atom segment, all
// segments of all frames of a batch in one launch.  What crosses PCIe is the compressed stream (~3.5 B/atom at
// precision 1000) instead of 12 B/atom.  Bit-identical to the host decoder (gr_xtc.h::decode_frame), which is
// bit-identical to the reference's decoders on the reference's own files (tests/test_xtc_decoder.py).
#pragma once
#include <hip/hip_runtime.h>
#include "gr_xtc.h"
#include "gr_layout.h"

__constant__ int gr_xtc_magic[73] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024, 1290, 1625,
    2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003, 16384, 20642, 26007, 32768, 41285, 52015, 65536, 82570, 104031, 131072,
    165140, 208063, 262144, 330280, 416127, 524287, 660561, 832255, 1048576, 1321122, 1664510, 2097152, 2642245, 3329021, 4194304,
    5284491, 6658042, 8388607, 10568983, 13316085, 16777216 };

// MSB-first bit reader over big-endian 32-bit words (the stream buffer is 16-byte aligned and zero padded)
struct GrBitsDev {
    const uint32_t *w; uint32_t idx; uint64_t win; int have;
    __device__ __forceinline__ void init(const unsigned char *stream, uint32_t bitpos) {
        w = reinterpret_cast<const uint32_t *>(stream); idx = bitpos >> 5; win = 0; have = 0;
        const int skip = (int)(bitpos & 31u);
        if (skip) { win = __builtin_bswap32(w[idx++]); have = 32 - skip; win &= (1ull << have) - 1ull; }
    }
    __device__ __forceinline__ uint32_t get(int n) {   // 0 <= n <= 32
        if (n == 0) return 0u;
        if (have < n) { win = (win << 32) | (uint64_t)__builtin_bswap32(w[idx++]); have += 32; }
        have -= n;
        const uint32_t v = (uint32_t)(win >> have) & (n == 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
        win &= (have == 0) ? 0ull : ((1ull << have) - 1ull);
        return v;
    }
    // the packed integer: bytes in read order are its little-endian bytes, the trailing partial byte the most significant
    __device__ __forceinline__ void get_limbs(int nbits, uint32_t (&l)[3]) {
        l[0] = l[1] = l[2] = 0u;
        int k = 0;
        while (nbits >= 32) { l[k++] = __builtin_bswap32(get(32)); nbits -= 32; }
        uint32_t v = 0u; int sh = 0;
        while (nbits > 8) { v |= get(8) << sh; sh += 8; nbits -= 8; }
        if (nbits > 0) v |= get(nbits) << sh;
        if (k < 3) l[k] = v;
    }
};

// three mixed-radix digits of the packed integer: v = (d0 * sz1 + d1) * sz2 + d2
__device__ __forceinline__ void gr_xtc_unpack3(GrBitsDev &b, int nbits, uint32_t sz1, uint32_t sz2, int (&out)[3]) {
    uint32_t l[3];
    b.get_limbs(nbits, l);
    if (nbits <= 32) {
        const uint32_t v = l[0], q2 = v / sz2, q1 = q2 / sz1;
        out[2] = (int)(v - q2 * sz2); out[1] = (int)(q2 - q1 * sz1); out[0] = (int)q1;
    } else if (nbits <= 64) {
        const uint64_t v = ((uint64_t)l[1] << 32) | l[0], q2 = v / sz2, q1 = q2 / sz1;
        out[2] = (int)(uint32_t)(v - q2 * sz2); out[1] = (int)(uint32_t)(q2 - q1 * sz1); out[0] = (int)(uint32_t)q1;
    } else {
        // up to 96 bits: schoolbook division by a 32-bit divisor, 32-bit limbs, most significant first
        uint32_t q[3]; uint64_t r = 0;
        for (int k = 2; k >= 0; --k) { const uint64_t cur = (r << 32) | l[k]; q[k] = (uint32_t)(cur / sz2); r = cur % sz2; }
        out[2] = (int)(uint32_t)r;
        uint32_t p[3]; r = 0;
        for (int k = 2; k >= 0; --k) { const uint64_t cur = (r << 32) | q[k]; p[k] = (uint32_t)(cur / sz1); r = cur % sz1; }
        out[1] = (int)(uint32_t)r; out[0] = (int)p[0];
    }
}

// one lane = one checkpoint segment (~32 atoms) of one frame; grid (ceil(max n_cp / 256), frames)
__global__ __launch_bounds__(256) void k_xtc_unpack(const unsigned char *__restrict__ streams, const grx::FrameDesc *__restrict__ descs,
                                                     const grx::Checkpoint *__restrict__ cps, float *__restrict__ frames, size_t frame_stride,
                                                     const uint32_t *__restrict__ slots, uint32_t n_atoms, const uint32_t *__restrict__ mask) {
    const grx::FrameDesc &d = descs[blockIdx.y];
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= d.n_cp) return;
    const grx::Checkpoint cp = cps[d.cp_off + m];
    // (a partial skim -- the reference's GroupXtcReader -- covers atoms [0, d.n_end) only; `mask` = one bit per atom: positions of
    // atoms outside the group are decoded on the way but NOT written: "all other atoms are left unchanged", molly_xtc.rs:585-587)
    const uint32_t end_atom = (m + 1 < d.n_cp) ? cps[d.cp_off + m + 1].atom : d.n_end;
    (void)n_atoms;
    if (cp.atom >= end_atom) return;
    GrBitsDev bits;
    bits.init(streams + d.stream_off, cp.bitpos);
    int smallidx = (int)(cp.state & 0xFFu), run = (int)((cp.state >> 8) & 0xFFu);
    const float inv = d.inv_precision;
    float *slot = frames + (size_t)slots[blockIdx.y] * frame_stride;
    uint32_t o = cp.atom;          // next atom to be written (the slot is pair-tiled: gr_pos_store)
    auto put = [&](uint32_t a, float x, float y, float z) { if (!mask || ((mask[a >> 5] >> (a & 31u)) & 1u)) gr_pos_store(slot, a, x, y, z); };
    uint32_t i = cp.atom;
    while (i < end_atom) {
        int cur[3];
        if (d.bitsize == 0) { cur[0] = (int)bits.get(d.bitsizeint[0]); cur[1] = (int)bits.get(d.bitsizeint[1]); cur[2] = (int)bits.get(d.bitsizeint[2]); }
        else gr_xtc_unpack3(bits, d.bitsize, d.sizeint[1], d.sizeint[2], cur);
        cur[0] += d.minint[0]; cur[1] += d.minint[1]; cur[2] += d.minint[2];
        ++i;
        int change = 0;
        if (bits.get(1)) {
            run = (int)bits.get(5);
            change = run % 3;
            run -= change;
            change -= 1;
        }
        if (run > 0) {
            const uint32_t szs = (uint32_t)gr_xtc_magic[smallidx];
            const int smallnum = (int)(szs / 2u);
            int prev[3] = { cur[0], cur[1], cur[2] };
            for (int k = 0; k < run; k += 3) {
                int dl[3];
                gr_xtc_unpack3(bits, smallidx, szs, szs, dl);
                const int nxt[3] = { dl[0] + prev[0] - smallnum, dl[1] + prev[1] - smallnum, dl[2] + prev[2] - smallnum };
                ++i;
                if (k == 0) {   // the first small atom is stored AFTER its successor: emit it first
                    put(o, nxt[0] * inv, nxt[1] * inv, nxt[2] * inv);
                    put(o + 1, prev[0] * inv, prev[1] * inv, prev[2] * inv);
                    o += 2;
                } else {
                    put(o, nxt[0] * inv, nxt[1] * inv, nxt[2] * inv);
                    o += 1;
                }
                prev[0] = nxt[0]; prev[1] = nxt[1]; prev[2] = nxt[2];
            }
        } else {
            put(o, cur[0] * inv, cur[1] * inv, cur[2] * inv);
            o += 1;
        }
        smallidx += change;
    }
}
