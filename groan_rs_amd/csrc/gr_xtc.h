// gr_xtc.h -- GROMACS .xtc trajectory decoder (host side, thread-safe random access), written for this project.
//
// This is the stage immediately in front of the geometry path (NEXT-1): the reference reads xtc
// through the `molly` crate or the vendored C xdrfile (src/io/xtc_io/molly_xtc.rs:268-308, src/io/xtc_io/xdrfile_xtc.rs:42-104,
// external/xdrfile/xdrfile.c:742-948).  Only the FILE FORMAT is shared with those; the implementation is new:
//   * the file is indexed once (frame offsets from the fixed-size headers), so any frame can be decoded by any thread
//     with pread() -- one frame per host thread is the unit of parallelism (the bit stream is serial within a frame);
//   * bits are pulled from a 64-bit window; the three packed integers of an atom are recovered from ONE integer
//     (<= 64 bits in a machine word, otherwise unsigned __int128) with two divisions instead of byte-wise long division;
//   * coordinates are written straight into the caller's buffer -- typically pinned memory that gr_frame_upload then
//     copies to the GPU on the copy stream while other threads decode the next frames.
//
// Format recap (XDR, big-endian): magic (1995, or 2023 with a 64-bit byte count), natoms, step, time, box[3][3], natoms;
// natoms <= 9: raw floats.  Otherwise: precision, minint[3], maxint[3], smallidx, nbytes, bit stream (padded to 4 bytes).
// Per atom: the full-range triple (mixed radix sizeint[] in `bitsize` bits, or three fixed-width fields when a range
// exceeds 24 bits), a flag bit, optionally 5 bits (run length of following "small" atoms + a -1/0/+1 change of the
// small-range index); small atoms are deltas in the mixed radix magic[smallidx]^3; the first small atom of a run is
// swapped with its predecessor (water oxygens/hydrogens).  magic[] is the format's table of ~2^(i/3) (with its
// historical irregular entries).
#pragma once
#include <climits>
#include <algorithm>
#include <cstdlib>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace grx {

static const int kMagic[] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024, 1290, 1625,
    2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003, 16384, 20642, 26007, 32768, 41285, 52015, 65536, 82570, 104031, 131072,
    165140, 208063, 262144, 330280, 416127, 524287, 660561, 832255, 1048576, 1321122, 1664510, 2097152, 2642245, 3329021, 4194304,
    5284491, 6658042, 8388607, 10568983, 13316085, 16777216 };
static const int kFirstIdx = 9;
static const int kLastIdx = (int)(sizeof(kMagic) / sizeof(kMagic[0]));

enum { XTC_OK = 0, XTC_E_IO = 1, XTC_E_FORMAT = 2, XTC_E_RANGE = 3, XTC_E_BOX = 4 };

inline uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
inline float bef(const unsigned char *p) { uint32_t u = be32(p); float f; memcpy(&f, &u, 4); return f; }
inline uint64_t be64(const unsigned char *p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
inline int bit_length(unsigned __int128 v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }
// two's-complement wrap-around sums: what the C decoder's `int` arithmetic does on a corrupt file, without the undefined behaviour
inline int wadd(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
inline int wsub(int a, int b) { return (int)((uint32_t)a - (uint32_t)b); }

struct FrameIndex {
    uint64_t offset;      // of the frame's magic number
    uint64_t data_offset; // of the bit stream (or of the raw floats when natoms <= 9)
    uint64_t nbytes;      // bit stream length
    int32_t step; float time; float box[9]; float precision;
    int32_t minint[3], maxint[3], smallidx;
};

struct File {
    int fd = -1;
    uint64_t size = 0;
    uint32_t natoms = 0;
    std::vector<FrameIndex> frames;
    std::string error;
};

// MSB-first bit reader over a byte buffer (the buffer carries GR_XTC_PAD bytes of zero padding behind the stream); the
// window is refilled four bytes at a time
struct Bits {
    const unsigned char *p; size_t pos = 0;   // next byte to pull
    uint64_t win = 0; int have = 0;           // `have` (< 32 between calls) valid bits at the bottom of win
    explicit Bits(const unsigned char *buf) : p(buf) {}
    inline uint32_t get(int n) {              // 0 <= n <= 32
        if (n == 0) return 0;
        if (have < n) { win = (win << 32) | (uint64_t)be32(p + pos); pos += 4; have += 32; }
        have -= n;
        return (uint32_t)((win >> have) & ((n == 32) ? 0xFFFFFFFFull : ((1ull << n) - 1)));
    }
    // the integer that packs three mixed-radix digits: bytes in read order are its little-endian bytes, the trailing
    // partial byte is the most significant.  The m whole bytes are read as one big-endian field and byte-swapped.
    inline uint64_t get_packed64(int nbits) {  // nbits <= 64
        const int m = nbits >> 3, rest = nbits & 7;
        uint64_t v = 0;
        if (m > 0) {
            uint64_t be = 0;
            if (m > 4) { be = (uint64_t)get(8 * (m - 4)) << 32; be |= get(32); } else be = get(8 * m);
            v = __builtin_bswap64(be << (64 - 8 * m));
        }
        if (rest) v |= (uint64_t)get(rest) << (8 * m);     // (m <= 7 here)
        return v;
    }
    inline unsigned __int128 get_packed(int nbits) {
        if (nbits <= 64) return get_packed64(nbits);
        const uint64_t lo = get_packed64(64);
        return (unsigned __int128)lo | ((unsigned __int128)get_packed64(nbits - 64) << 64);
    }
};

// quotient and remainder by a divisor known in advance: below 2^52 the product with the stored reciprocal is within one of
// the true quotient (the numerator converts exactly, the reciprocal is correctly rounded), one compare fixes it -- a third of
// the latency of the 64-bit division; above that the division itself
inline uint64_t divrem(uint64_t v, uint32_t d, double inv, uint32_t &rem) {
    if (v < (1ull << 52)) {
        uint64_t q = (uint64_t)((double)v * inv);
        int64_t r = (int64_t)(v - q * d);
        if (r < 0) { --q; r += d; } else if (r >= (int64_t)d) { ++q; r -= d; }
        rem = (uint32_t)r;
        return q;
    }
    const uint64_t q = v / d; rem = (uint32_t)(v - q * d);
    return q;
}
// inv[0] = 1 / sz[1], inv[1] = 1 / sz[2]
inline void unpack3(Bits &b, int nbits, const uint32_t sz[3], const double inv[2], int out[3]) {
    if (nbits <= 64) {
        const uint64_t v = b.get_packed64(nbits);
        uint32_t r2, r1;
        const uint64_t q2 = divrem(v, sz[2], inv[1], r2);
        const uint64_t q1 = divrem(q2, sz[1], inv[0], r1);
        out[2] = (int)r2; out[1] = (int)r1; out[0] = (int)(uint32_t)q1;
    } else {
        unsigned __int128 v = b.get_packed(nbits);
        const unsigned __int128 q2 = v / sz[2]; out[2] = (int)(uint64_t)(v - q2 * sz[2]);
        const unsigned __int128 q1 = q2 / sz[1]; out[1] = (int)(uint64_t)(q2 - q1 * sz[1]);
        out[0] = (int)(uint32_t)(uint64_t)q1;
    }
}

inline bool pread_all(int fd, void *buf, size_t n, uint64_t off) {
    unsigned char *p = (unsigned char *)buf;
    while (n) {
        ssize_t r = pread(fd, p, n, (off_t)off);
        if (r <= 0) return false;
        p += r; n -= (size_t)r; off += (uint64_t)r;
    }
    return true;
}

// index the whole file: every frame's header is read, its payload skipped
inline int open_file(File &f, const char *path) {
    f.fd = ::open(path, O_RDONLY);
    if (f.fd < 0) { f.error = std::string("cannot open ") + path; return XTC_E_IO; }
    struct stat st;
    if (fstat(f.fd, &st) != 0) { f.error = "fstat failed"; return XTC_E_IO; }
    f.size = (uint64_t)st.st_size;
    uint64_t off = 0;
    unsigned char h[128];
    while (off + 16 <= f.size) {
        if (!pread_all(f.fd, h, 16, off)) { f.error = "short read in frame header"; return XTC_E_IO; }
        const uint32_t magic = be32(h);
        if (magic != 1995 && magic != 2023) { f.error = "bad magic number"; return XTC_E_FORMAT; }
        FrameIndex fi; memset(&fi, 0, sizeof(fi));
        fi.offset = off;
        const uint32_t natoms = be32(h + 4);
        if (f.frames.empty()) f.natoms = natoms;
        else if (natoms != f.natoms) { f.error = "number of atoms changes between frames"; return XTC_E_FORMAT; }
        fi.step = (int32_t)be32(h + 8); fi.time = bef(h + 12);
        uint64_t p = off + 16;
        if (!pread_all(f.fd, h, 40, p)) { f.error = "short read in box"; return XTC_E_IO; }
        for (int k = 0; k < 9; ++k) fi.box[k] = bef(h + 4 * k);
        if (be32(h + 36) != natoms) { f.error = "atom count mismatch inside a frame"; return XTC_E_FORMAT; }
        p += 40;
        if (natoms <= 9) {
            fi.data_offset = p; fi.nbytes = (uint64_t)natoms * 12; fi.precision = 0.0f;
            p += fi.nbytes;
        } else {
            const size_t hl = (magic == 2023) ? 40 : 36;
            if (!pread_all(f.fd, h, hl, p)) { f.error = "short read in compression header"; return XTC_E_IO; }
            fi.precision = bef(h);
            for (int k = 0; k < 3; ++k) { fi.minint[k] = (int32_t)be32(h + 4 + 4 * k); fi.maxint[k] = (int32_t)be32(h + 16 + 4 * k); }
            fi.smallidx = (int32_t)be32(h + 28);
            fi.nbytes = (magic == 2023) ? be64(h + 32) : (uint64_t)be32(h + 32);
            p += hl;
            fi.data_offset = p;
            // the byte count comes from the file (64 bits wide for magic 2023): it must fit what is left of the file BEFORE it
            // is rounded up or added to anything -- 2^64 - 1 would round to 0, pass the size test below and later size the
            // decoder's scratch buffer by wrap-around.  2^29 is the decoder's own limit (bit positions are 32-bit).
            if (p > f.size || fi.nbytes > f.size - p || fi.nbytes >= (1ull << 29)) { f.error = "frame byte count exceeds the file"; return XTC_E_FORMAT; }
            p += (fi.nbytes + 3) & ~(uint64_t)3;
            if (fi.smallidx < kFirstIdx || fi.smallidx >= kLastIdx) { f.error = "small-range index out of the table"; return XTC_E_FORMAT; }
        }
        if (p > f.size) { f.error = "truncated frame"; return XTC_E_FORMAT; }
        f.frames.push_back(fi);
        off = p;
    }
    return XTC_OK;
}

// decode one frame into xyz[natoms][3]; scratch is grown as needed (one per calling thread)
// zero bytes kept behind a bit stream: the position is checked once per group, and one group of a corrupt stream can pull
// 96 + 6 + 8 * 72 bits (< 96 bytes) past the last valid position before that check sees it
#define GR_XTC_PAD 128
// n_want < natoms = PARTIAL decode (the reference's GroupXtcReader through the molly crate, molly_xtc.rs:475-560: "all positions
// up to the end of the group must be loaded" and nothing behind it): the bit stream is walked only until atom n_want - 1 has
// been produced and only `have` bytes of it are in `scratch` (the caller reads a prefix and retries with more when the walk
// runs off its end: XTC_E_RANGE).  xyz receives n_want atoms; groups that straddle n_want are decoded into `spill`.
inline int decode_frame(const File &f, const FrameIndex &fi, float *xyz, std::vector<unsigned char> &scratch, uint32_t n_want = 0xFFFFFFFFu, size_t have = (size_t)-1) {
    const uint32_t n = f.natoms;
    const bool partial = n_want < n;
    if (!partial) {
        n_want = n;
        have = (size_t)fi.nbytes;
        scratch.resize((size_t)fi.nbytes + GR_XTC_PAD);
        if (!pread_all(f.fd, scratch.data(), (size_t)fi.nbytes, fi.data_offset)) return XTC_E_IO;
        memset(scratch.data() + fi.nbytes, 0, GR_XTC_PAD);
    }
    if (n <= 9) {
        for (uint32_t k = 0; k < 3 * n_want; ++k) xyz[k] = bef(scratch.data() + 4 * k);
        return XTC_OK;
    }
    uint32_t sizeint[3];
    for (int k = 0; k < 3; ++k) sizeint[k] = (uint32_t)fi.maxint[k] - (uint32_t)fi.minint[k] + 1u;
    int bitsizeint[3] = { 0, 0, 0 }, bitsize;
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffffu) {
        for (int k = 0; k < 3; ++k) { int b = bit_length(sizeint[k]); bitsizeint[k] = b > 32 ? 32 : b; }
        bitsize = 0;
    } else {
        bitsize = bit_length((unsigned __int128)sizeint[0] * sizeint[1] * sizeint[2]);
    }
    int smallidx = fi.smallidx;
    int smaller = kMagic[smallidx - 1 > kFirstIdx ? smallidx - 1 : kFirstIdx] / 2;
    int smallnum = kMagic[smallidx] / 2;
    uint32_t sizesmall[3] = { (uint32_t)kMagic[smallidx], (uint32_t)kMagic[smallidx], (uint32_t)kMagic[smallidx] };
    if (sizeint[0] == 0 || sizeint[1] == 0 || sizeint[2] == 0) return XTC_E_FORMAT;   // maxint - minint + 1 wrapped: a corrupt header
    const double inv_large[2] = { 1.0 / (double)sizeint[1], 1.0 / (double)sizeint[2] };
    double inv_small[2] = { 1.0 / (double)sizesmall[1], 1.0 / (double)sizesmall[2] };
    const float inv_precision = (float)(1.0 / (double)fi.precision);
    Bits bits(scratch.data());
    const size_t limit = (size_t)fi.nbytes + 8;
    // partial decode: atoms land in a buffer that also takes the tail of a group straddling n_want (a group holds <= 9 atoms)
    std::vector<float> part;
    float *out = xyz;
    if (partial) { part.resize(3 * ((size_t)n_want + 10)); out = part.data(); }
    const float *out0 = out;
    int run = 0;
    uint32_t i = 0;
    while (i < n_want) {
        int cur[3];
        if (bitsize == 0) { cur[0] = (int)bits.get(bitsizeint[0]); cur[1] = (int)bits.get(bitsizeint[1]); cur[2] = (int)bits.get(bitsizeint[2]); }
        else unpack3(bits, bitsize, sizeint, inv_large, cur);
        cur[0] = wadd(cur[0], fi.minint[0]); cur[1] = wadd(cur[1], fi.minint[1]); cur[2] = wadd(cur[2], fi.minint[2]);
        ++i;
        int change = 0;
        if (bits.get(1)) {
            run = (int)bits.get(5);
            change = run % 3;
            run -= change;
            change -= 1;
        }
        if (run > 0) {
            if ((uint64_t)i + (uint64_t)(run / 3) > n) return XTC_E_FORMAT;   // a run must not overshoot the frame
            int prev[3] = { cur[0], cur[1], cur[2] };
            for (int k = 0; k < run; k += 3) {
                int d[3];
                unpack3(bits, smallidx, sizesmall, inv_small, d);
                int nxt[3] = { wsub(wadd(d[0], prev[0]), smallnum), wsub(wadd(d[1], prev[1]), smallnum), wsub(wadd(d[2], prev[2]), smallnum) };
                ++i;
                if (k == 0) {
                    // the first small atom is stored AFTER its successor (water: O H H -> H O H): emit it first
                    *out++ = nxt[0] * inv_precision; *out++ = nxt[1] * inv_precision; *out++ = nxt[2] * inv_precision;
                    *out++ = prev[0] * inv_precision; *out++ = prev[1] * inv_precision; *out++ = prev[2] * inv_precision;
                    prev[0] = nxt[0]; prev[1] = nxt[1]; prev[2] = nxt[2];   // the delta chain continues from the decoded atom
                } else {
                    *out++ = nxt[0] * inv_precision; *out++ = nxt[1] * inv_precision; *out++ = nxt[2] * inv_precision;
                    prev[0] = nxt[0]; prev[1] = nxt[1]; prev[2] = nxt[2];
                }
            }
        } else {
            *out++ = cur[0] * inv_precision; *out++ = cur[1] * inv_precision; *out++ = cur[2] * inv_precision;
        }
        if (bits.pos > limit) return XTC_E_FORMAT;
        // (the window pulls 4 bytes at a time: what counts is the bits CONSUMED, not the bytes pulled)
        if (partial && (uint64_t)bits.pos * 8 - (uint64_t)bits.have > (uint64_t)have * 8) return XTC_E_RANGE;      // walked off the prefix that was read: the caller reads more
        smallidx += change;
        if (smallidx < kFirstIdx || smallidx >= kLastIdx) return XTC_E_FORMAT;
        if (change < 0) { smallnum = smaller; smaller = smallidx > kFirstIdx ? kMagic[smallidx - 1] / 2 : 0; }
        else if (change > 0) { smaller = smallnum; smallnum = kMagic[smallidx] / 2; }
        if (change != 0) { sizesmall[0] = sizesmall[1] = sizesmall[2] = (uint32_t)kMagic[smallidx]; inv_small[0] = inv_small[1] = 1.0 / (double)sizesmall[1]; }
    }
    if (partial) memcpy(xyz, out0, 3 * (size_t)n_want * sizeof(float));
    return XTC_OK;
}

// decode the first n_want atoms of a frame, reading only as much of its bit stream as the walk needs (estimate from the
// frame's mean bits per atom, doubled on a miss)
inline int decode_frame_prefix(const File &f, const FrameIndex &fi, uint32_t n_want, float *xyz, std::vector<unsigned char> &scratch, size_t *bytes_read = nullptr) {
    if (n_want >= f.natoms || f.natoms <= 9) { if (bytes_read) *bytes_read = (size_t)fi.nbytes; return decode_frame(f, fi, xyz, scratch); }
    if (n_want == 0) { if (bytes_read) *bytes_read = 0; return XTC_OK; }
    size_t want = (size_t)((double)fi.nbytes * ((double)n_want / (double)f.natoms) * 1.3) + 256;
    for (;;) {
        if (want > (size_t)fi.nbytes) want = (size_t)fi.nbytes;
        scratch.resize(want + GR_XTC_PAD);
        if (!pread_all(f.fd, scratch.data(), want, fi.data_offset)) return XTC_E_IO;
        memset(scratch.data() + want, 0, GR_XTC_PAD);
        const int r = decode_frame(f, fi, xyz, scratch, n_want, want);
        if (r != XTC_E_RANGE || want == (size_t)fi.nbytes) { if (bytes_read) *bytes_read = want; return r == XTC_E_RANGE ? (int)XTC_E_FORMAT : r; }
        want *= 2;
    }
}


// ---- GPU-side unpacking (gr_xtc_dev.h): the host only SKIMS a frame's bit stream -- group boundaries depend on the
// flag / run-length fields alone, no unpacking, no division -- and records the decoder state every GR_XTC_CP_ATOMS
// atoms; the device then unpacks all the segments of all the frames of a batch in parallel.
#define GR_XTC_CP_ATOMS 32
struct Checkpoint { uint32_t bitpos, atom, state; };   // state = smallidx | run << 8, at the first group that starts at atom >= 32 m
struct FrameDesc {                                      // one per frame of a batch (plain data, copied to the device)
    uint64_t stream_off;       // byte offset of the frame's bit stream in the batch's stream buffer (16-byte aligned, zero padded)
    uint64_t cp_off;           // index of the frame's first checkpoint
    uint32_t nbytes, n_cp;
    int32_t minint[3]; uint32_t sizeint[3];
    int32_t bitsize, bitsizeint[3];
    float inv_precision;
    uint32_t n_end;            // atoms [0, n_end) are covered by the checkpoints (n_atoms for a full skim)
};

inline uint32_t peek_bits(const unsigned char *p, uint64_t bitpos, int n) {   // n <= 32; p is padded by >= 8 bytes
    const uint64_t w = be64(p + (bitpos >> 3));
    return (uint32_t)((w >> (64 - (int)(bitpos & 7) - n)) & ((n == 32) ? 0xFFFFFFFFull : ((1ull << n) - 1)));
}

// Walk the groups of one frame (natoms > 9): fills the descriptor's decoding constants and the checkpoints.
// Where the next group starts depends on the flag / run bits of this one: one dependent load per group, ~170 k groups per
// 5e5-atom water frame.  (Walking two frames in lockstep in one thread -- two dependency chains in flight -- and a
// branch-free flag / run selection were both measured slower: the branches predict well, and speculation past them is what
// hides the load latency.)
// n_stop < n: PARTIAL skim (GroupXtcReader, molly_xtc.rs:475-560): stop at the first group that starts at or behind atom n_stop;
// the checkpoints then cover atoms [0, n_end) and d.nbytes is the length of the stream prefix they need.  have_bytes: the
// prefix of the stream that is in memory -- walking past it is XTC_E_RANGE (the caller reads more and walks again).
inline int skim_frame(const unsigned char *stream, const FrameIndex &fi, uint32_t n, FrameDesc &d, std::vector<Checkpoint> &cps, uint32_t n_stop = 0xFFFFFFFFu,
                      uint64_t have_bytes = ~0ull) {
    if (n_stop > n) n_stop = n;
    for (int k = 0; k < 3; ++k) { d.minint[k] = fi.minint[k]; d.sizeint[k] = (uint32_t)fi.maxint[k] - (uint32_t)fi.minint[k] + 1u; }
    if (d.sizeint[0] == 0 || d.sizeint[1] == 0 || d.sizeint[2] == 0) return XTC_E_FORMAT;   // corrupt header (the decoder refuses it too)
    d.bitsizeint[0] = d.bitsizeint[1] = d.bitsizeint[2] = 0;
    int large_bits;
    if ((d.sizeint[0] | d.sizeint[1] | d.sizeint[2]) > 0xffffffu) {
        for (int k = 0; k < 3; ++k) { int b = bit_length(d.sizeint[k]); d.bitsizeint[k] = b > 32 ? 32 : b; }
        d.bitsize = 0;
        large_bits = d.bitsizeint[0] + d.bitsizeint[1] + d.bitsizeint[2];
    } else {
        d.bitsize = bit_length((unsigned __int128)d.sizeint[0] * d.sizeint[1] * d.sizeint[2]);
        large_bits = d.bitsize;
    }
    d.inv_precision = (float)(1.0 / (double)fi.precision);
    d.nbytes = (uint32_t)fi.nbytes;
    if (fi.nbytes >= (1ull << 29)) return XTC_E_FORMAT;        // bit positions are 32-bit
    const uint64_t limit_bits = (fi.nbytes + 8) * 8;
    const uint64_t have_bits = have_bytes > (~0ull >> 3) ? ~0ull : have_bytes * 8;
    const uint64_t safe_bits = limit_bits < have_bits ? limit_bits : have_bits;
    const uint32_t want = (n_stop + GR_XTC_CP_ATOMS - 1) / GR_XTC_CP_ATOMS;
    cps.resize((size_t)want + 1);                              // written through a bare pointer below: no capacity check per group
    Checkpoint *cp = cps.data();
    uint32_t n_cp = 0;
    int smallidx = fi.smallidx, run = 0;
    uint64_t bitpos = 0;
    uint32_t i = 0, next_cp = 0;
    while (i < n_stop) {
        // (at most one window boundary is passed per group -- a group holds at most 9 atoms, a window 32 -- and i < n_stop
        // keeps n_cp below `want`)
        if (i >= next_cp) { cp[n_cp++] = Checkpoint{ (uint32_t)bitpos, i, (uint32_t)smallidx | ((uint32_t)run << 8) }; next_cp += GR_XTC_CP_ATOMS; }
        {   // fast stretch: a group whose flag bit is clear has the shape of the one before it (same run, same small range) --
            // the common case by far (3 % of the groups of a water frame carry a flag) -- so up to the next window boundary
            // the walk is "test one bit, add a constant".  At most 32 groups fit (a group holds >= 1 atom), and the stretch
            // is entered only when that many, plus the last flag / run field, lie inside the stream that is in memory.
            const uint32_t adv = 1u + (uint32_t)run / 3u;
            const uint64_t stride = (uint64_t)large_bits + 1u + (uint64_t)((uint32_t)run / 3u) * (uint64_t)smallidx;
            const uint32_t lim = next_cp < n_stop ? next_cp : n_stop;
            if (bitpos + GR_XTC_CP_ATOMS * stride + (uint64_t)large_bits + 6 <= safe_bits) {
                while (i < lim) {
                    const uint64_t fp = bitpos + (uint64_t)large_bits;
                    if ((stream[fp >> 3] >> (7u - (unsigned)(fp & 7u))) & 1u) break;
                    bitpos += stride; i += adv;
                }
                if (i > n) return XTC_E_FORMAT;                 // a run went past the last atom
                if (i >= lim) continue;                         // next window (its checkpoint) or the end
            }
        }
        bitpos += (uint64_t)large_bits;
        if (bitpos + 6 > limit_bits) return XTC_E_FORMAT;       // the flag / run field below must lie inside the (padded) stream
        ++i;
        int change = 0;
        const uint32_t six = peek_bits(stream, bitpos, 6);      // flag bit + the 5 run bits behind it, one load
        if (six & 32u) { run = (int)(six & 31u); bitpos += 6; change = run % 3; run -= change; change -= 1; } else bitpos += 1;
        if (run > 0) {
            if ((uint64_t)i + (uint64_t)(run / 3) > n) return XTC_E_FORMAT;
            bitpos += (uint64_t)(run / 3) * (uint64_t)smallidx;
            i += (uint32_t)(run / 3);
        }
        if (bitpos > limit_bits) return XTC_E_FORMAT;
        if (bitpos > have_bits) return XTC_E_RANGE;             // partial read: the walk left the prefix that is in memory
        smallidx += change;
        if (smallidx < kFirstIdx || smallidx >= kLastIdx) return XTC_E_FORMAT;
    }
    // one checkpoint per started window of 32 atoms of [0, n_stop); pad so that the count is exactly ceil(n_stop / 32)
    while (n_cp < want) cp[n_cp++] = Checkpoint{ (uint32_t)bitpos, i, (uint32_t)smallidx };
    cps.resize(want);
    d.n_cp = want;
    d.n_end = i < n ? i : n;                  // first atom NOT covered by the checkpoints' segments (partial: >= n_stop)
    if (n_stop < n) d.nbytes = (uint32_t)std::min<uint64_t>((bitpos + 7) / 8 + 16, fi.nbytes);
    return XTC_OK;
}

// ================================================================================================ writer
// The encoder of the same format (what the reference's XtcWriter produces through xdrfile's write_xtc,
// src/io/xtc_io/mod.rs:256-331): byte-for-byte the stream GROMACS' xdr3dfcoord writes for the same coordinates, because a
// reader cannot tell which of several valid encodings a writer chose and the reference's golden fitted trajectories
// (short_trajectory_fit.xtc) are compared as files.  The choices that shape the stream:
//   * quantisation q = trunc(x * precision +- 0.5) in single precision (the 0.5 is added in double, the sum rounded to float);
//   * the starting small-range index = first table entry >= the smallest L1 step between consecutive atoms;
//   * an atom whose successor lies within the small range is SWAPPED with it (water: O H H is stored H O H) and opens a
//     run of up to 8 small atoms; the range index moves down by one when a whole group stayed inside the next smaller
//     range, up by one when the group's first atom was near its predecessor ("larger" range) ...
//   * ... signalled by one flag bit + 5 bits (run length * 3 + change + 1) only when run length or index change.
// Provenance: the xtc stream format has no specification beyond its one implementation, so `encode_coords` below FOLLOWS THE
// DECISION PROCEDURE of xdrfile's `xdrfile_compress_coord_float` (xdrfile.c of the GROMACS xdrfile library, BSD 2-clause
// licence, (c) 2009-2014 Erik Lindahl, David van der Spoel; the reference vendors it) -- which atoms are swapped, when a run
// ends, when the small-range index moves, which bits announce it -- because any other valid encoding would differ from the
// reference writer's bytes.  The code is written anew (64-bit packing, 32-bit bit writer, different structure); the decisions are theirs.
// MSB-first bit writer
struct BitWriter {
    std::vector<unsigned char> &out; unsigned char *p; uint64_t acc = 0; int nacc = 0;   // nacc < 32 pending bits at the bottom of acc
    // `o` must already be sized for the worst case (encode_coords: 16 bytes per atom + slack); finish() trims it
    explicit BitWriter(std::vector<unsigned char> &o) : out(o), p(o.data()) {}
    inline void put(int nbits, uint32_t v) {   // 0 <= nbits <= 32, v < 2^nbits
        acc = (acc << nbits) | (uint64_t)v; nacc += nbits;
        if (nacc >= 32) {                      // four bytes at a time, most significant first
            const uint32_t w = __builtin_bswap32((uint32_t)(acc >> (nacc - 32)));
            memcpy(p, &w, 4); p += 4; nacc -= 32;
            acc &= (1ull << nacc) - 1ull;
        }
    }
    // a packed integer of `nbits` bits: its bytes go out least significant first, the last (partial) chunk holds the top bits.
    // m whole bytes b0 b1 .. in stream order are the byte-swapped low m bytes of v, written as at most two fields
    inline void put_packed64(int nbits, uint64_t v) {   // nbits <= 64
        const int m = nbits >> 3, rest = nbits & 7;
        if (m > 0) {
            const uint64_t sw = __builtin_bswap64(v) >> (64 - 8 * m);   // b0 is now the most significant of the m bytes
            if (m > 4) { put(8 * (m - 4), (uint32_t)(sw >> 32)); put(32, (uint32_t)sw); }
            else put(8 * m, (uint32_t)sw);
        }
        if (rest) put(rest, (uint32_t)((m < 8 ? v >> (8 * m) : 0ull) & ((1u << rest) - 1u)));
    }
    inline void put_packed(int nbits, unsigned __int128 v) {
        if (nbits <= 64) { put_packed64(nbits, (uint64_t)v); return; }
        put_packed64(64, (uint64_t)v);
        put_packed64(nbits - 64, (uint64_t)(v >> 64));
    }
    inline void finish() {
        while (nacc >= 8) { *p++ = (unsigned char)(acc >> (nacc - 8)); nacc -= 8; }
        if (nacc > 0) { *p++ = (unsigned char)((acc << (8 - nacc)) & 0xff); nacc = 0; }
        acc = 0;
        out.resize((size_t)(p - out.data()));
    }
};

struct EncodedFrame {
    float precision; int32_t minint[3], maxint[3], smallidx;
    std::vector<unsigned char> bytes;   // the bit stream (not yet padded to 4 bytes)
};

// false when the scaled value does not fit the format's integers (the C code prints "Internal overflow compressing
// coordinates." and converts anyway -- undefined; here the frame is refused)
inline bool quantise(float x, float precision, int &q) {
    const float prod = x * precision;
    const float lf = (float)((double)prod + (x >= 0.0f ? 0.5 : -0.5));
    if (!(fabsf(lf) <= (float)(INT_MAX - 2))) return false;   // also NaN
    q = (int)lf;
    return true;
}

// xyz[n][3] (n > 9) -> header integers + bit stream; `ints` is scratch (3 n ints)
inline bool encode_coords(const float *xyz, uint32_t n, float precision, EncodedFrame &e, std::vector<int> &ints) {
    if (!(precision > 0.0f)) precision = 1000.0f;
    e.precision = precision;
    ints.resize(3 * (size_t)n);
    int mn[3] = { INT_MAX, INT_MAX, INT_MAX }, mx[3] = { INT_MIN, INT_MIN, INT_MIN };
    long long mindiff = INT_MAX;
    int old[3] = { 0, 0, 0 };
    for (uint32_t i = 0; i < n; ++i) {
        int q[3];
        const bool missing = xyz[3 * (size_t)i] != xyz[3 * (size_t)i];   // None travels as NaN in x: written as the origin (xtc_io/mod.rs:296-301)
        for (int a = 0; a < 3; ++a) {
            const float x = missing ? 0.0f : xyz[3 * (size_t)i + a];
            if (!quantise(x, precision, q[a])) return false;
            if (q[a] < mn[a]) mn[a] = q[a];
            if (q[a] > mx[a]) mx[a] = q[a];
            ints[3 * (size_t)i + a] = q[a];
        }
        const long long diff = llabs((long long)old[0] - q[0]) + llabs((long long)old[1] - q[1]) + llabs((long long)old[2] - q[2]);
        if (i > 0 && diff < mindiff) mindiff = diff;
        old[0] = q[0]; old[1] = q[1]; old[2] = q[2];
    }
    uint32_t sizeint[3];
    for (int a = 0; a < 3; ++a) {
        if ((float)mx[a] - (float)mn[a] >= (float)(INT_MAX - 2)) return false;   // value - minint would not fit
        e.minint[a] = mn[a]; e.maxint[a] = mx[a]; sizeint[a] = (uint32_t)mx[a] - (uint32_t)mn[a] + 1u;
    }
    int bitsizeint[3] = { 0, 0, 0 }, bitsize;
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffffu) {
        for (int a = 0; a < 3; ++a) { int b = bit_length(sizeint[a]); bitsizeint[a] = b > 32 ? 32 : b; }
        bitsize = 0;
    } else {
        bitsize = bit_length((unsigned __int128)sizeint[0] * sizeint[1] * sizeint[2]);
    }
    int smallidx = kFirstIdx;
    while (smallidx < kLastIdx && kMagic[smallidx < kLastIdx ? smallidx : kLastIdx - 1] < mindiff) ++smallidx;
    if (smallidx > kLastIdx - 1) smallidx = kLastIdx - 1;   // (the C code would index one past its table here; unreachable for real data)
    e.smallidx = smallidx;
    int maxidx = std::min(kLastIdx - 1, smallidx + 8);
    const int minidx = maxidx - 8;
    int smaller = kMagic[std::max(kFirstIdx, smallidx - 1)] / 2;
    int smallnum = kMagic[smallidx] / 2;
    uint32_t sizesmall = (uint32_t)kMagic[smallidx];
    const int larger = kMagic[maxidx] / 2;
    e.bytes.resize((size_t)n * 16 + 16);   // worst case: 96 bits + 6 flag bits per atom
    BitWriter bw(e.bytes);
    int prev[3] = { 0, 0, 0 };
    int prevrun = -1;
    uint32_t i = 0;
    auto near = [](const int *a, const int *b, int lim) { return std::abs(a[0] - b[0]) < lim && std::abs(a[1] - b[1]) < lim && std::abs(a[2] - b[2]) < lim; };
    while (i < n) {
        int *cur = &ints[3 * (size_t)i];
        int is_smaller = 0, is_small = 0;
        if (smallidx < maxidx && i >= 1 && near(cur, prev, larger)) is_smaller = 1;
        else if (smallidx > minidx) is_smaller = -1;
        if (i + 1 < n && near(cur, cur + 3, smallnum)) {
            for (int a = 0; a < 3; ++a) std::swap(cur[a], cur[3 + a]);
            is_small = 1;
        }
        if (bitsize == 0) {
            for (int a = 0; a < 3; ++a) bw.put(bitsizeint[a], (uint32_t)(cur[a] - mn[a]));
        } else if (bitsize <= 64) {
            bw.put_packed64(bitsize, ((uint64_t)(uint32_t)(cur[0] - mn[0]) * sizeint[1] + (uint32_t)(cur[1] - mn[1])) * sizeint[2] + (uint32_t)(cur[2] - mn[2]));
        } else {
            const unsigned __int128 v = ((unsigned __int128)(uint32_t)(cur[0] - mn[0]) * sizeint[1] + (uint32_t)(cur[1] - mn[1])) * sizeint[2] + (uint32_t)(cur[2] - mn[2]);
            bw.put_packed(bitsize, v);
        }
        prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
        cur += 3; ++i;
        int run = 0;
        uint32_t small[24];
        if (is_small == 0 && is_smaller == -1) is_smaller = 0;
        while (is_small && run < 24) {
            // the C code does this test in 32-bit ints; beyond ~46 k quanta per step the squares wrap, and a byte-compatible
            // stream has to wrap with them
            uint32_t usum = 0;
            for (int a = 0; a < 3; ++a) { const uint32_t d = (uint32_t)(cur[a] - prev[a]); usum += d * d; }
            if (is_smaller == -1 && (int32_t)usum >= (int32_t)((uint32_t)smaller * (uint32_t)smaller)) is_smaller = 0;
            for (int a = 0; a < 3; ++a) small[run++] = (uint32_t)(cur[a] - prev[a] + smallnum);
            prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
            ++i; cur += 3;
            is_small = (i < n && near(cur, prev, smallnum)) ? 1 : 0;
        }
        if (run != prevrun || is_smaller != 0) {
            prevrun = run;
            bw.put(1, 1u);
            bw.put(5, (uint32_t)(run + is_smaller + 1));
        } else {
            bw.put(1, 0u);
        }
        for (int k = 0; k < run; k += 3) {
            if (smallidx <= 64) bw.put_packed64(smallidx, ((uint64_t)small[k] * sizesmall + small[k + 1]) * sizesmall + small[k + 2]);
            else bw.put_packed(smallidx, ((unsigned __int128)small[k] * sizesmall + small[k + 1]) * sizesmall + small[k + 2]);
        }
        if (is_smaller != 0) {
            smallidx += is_smaller;
            if (is_smaller < 0) { smallnum = smaller; smaller = kMagic[smallidx - 1] / 2; }
            else { smaller = smallnum; smallnum = kMagic[smallidx] / 2; }
            sizesmall = (uint32_t)kMagic[smallidx];
        }
    }
    bw.finish();
    return true;
}

inline void put_be32(std::vector<unsigned char> &o, uint32_t v) { o.push_back((unsigned char)(v >> 24)); o.push_back((unsigned char)(v >> 16)); o.push_back((unsigned char)(v >> 8)); o.push_back((unsigned char)v); }
inline void put_bef(std::vector<unsigned char> &o, float f) { uint32_t u; memcpy(&u, &f, 4); put_be32(o, u); }

// one whole frame as it appears in the file (magic 1995): header, box (rows = box vectors), coordinates
inline bool serialise_frame(std::vector<unsigned char> &o, uint32_t natoms, int32_t step, float time, const float box_rows[9], const float *xyz,
                            float precision, EncodedFrame &scratch, std::vector<int> &ints) {
    o.clear();
    put_be32(o, 1995u); put_be32(o, natoms); put_be32(o, (uint32_t)step); put_bef(o, time);
    for (int k = 0; k < 9; ++k) put_bef(o, box_rows[k]);
    put_be32(o, natoms);
    if (natoms <= 9) {
        for (uint32_t k = 0; k < 3 * natoms; ++k) put_bef(o, (xyz[3 * (k / 3)] != xyz[3 * (k / 3)]) ? 0.0f : xyz[k]);
        return true;
    }
    if (!encode_coords(xyz, natoms, precision, scratch, ints)) { o.clear(); return false; }
    put_bef(o, scratch.precision);
    for (int a = 0; a < 3; ++a) put_be32(o, (uint32_t)scratch.minint[a]);
    for (int a = 0; a < 3; ++a) put_be32(o, (uint32_t)scratch.maxint[a]);
    put_be32(o, (uint32_t)scratch.smallidx);
    put_be32(o, (uint32_t)scratch.bytes.size());
    o.insert(o.end(), scratch.bytes.begin(), scratch.bytes.end());
    while (o.size() & 3u) o.push_back(0);
    return true;
}

}  // namespace grx
