// Own reader for the GROMACS trr trajectory format (the reference reads it through the vendored C xdrfile:
// src/io/trr_io.rs:30-135 over external/xdrfile/xdrfile_trr.c).  Host only, no HIP.
//
// A frame is an XDR (big-endian) header -- magic 1993, the version string "GMX_trn_file", thirteen integers (the byte sizes of
// the optional sections, the atom count, step, nre), time and lambda -- followed by box / virial / pressure matrices and the
// position / velocity / force arrays, every real in ONE precision per frame (float or double, told by section size / count).
// The file is indexed once (frames are random-access, `pread`, thread-safe like the xtc reader); coordinates are converted
// to f32 (the reference's rvec) straight into the caller's buffers.  Sections a frame does not carry read as zeros -- what
// xdrfile's read_trr leaves in the reference's zero-initialised vectors, and what TrrFrameData::update_system turns into
// "no position / velocity / force" (trr_io.rs:101-126).
#pragma once
#include <cstdint>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace grtr {

enum { TRR_OK = 0, TRR_E_IO = 1, TRR_E_FORMAT = 2 };

inline uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
inline uint64_t be64(const unsigned char *p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
inline float bef(const unsigned char *p) { const uint32_t u = be32(p); float f; memcpy(&f, &u, 4); return f; }
inline double bed(const unsigned char *p) { const uint64_t u = be64(p); double d; memcpy(&d, &u, 8); return d; }

struct FrameIndex {
    uint64_t offset;                 // of the magic number
    uint64_t x_off, v_off, f_off;    // of the arrays (0 = the frame does not carry that section)
    uint32_t real_size;              // 4 or 8
    int32_t step;
    float time, lambda;
    float box[9];                    // rows = box vectors; zeros when the frame has no box
    uint32_t has_box;
};

struct File {
    int fd = -1;
    uint64_t size = 0;
    uint32_t natoms = 0;
    std::vector<FrameIndex> frames;
    std::string error;
};

inline bool pread_all(int fd, void *buf, size_t n, uint64_t off) {
    unsigned char *p = (unsigned char *)buf;
    while (n) {
        const ssize_t r = pread(fd, p, n, (off_t)off);
        if (r <= 0) return false;
        p += r; n -= (size_t)r; off += (uint64_t)r;
    }
    return true;
}

// index the whole file: every frame's header is parsed and checked, its payload skipped
inline int open_file(File &f, const char *path) {
    f.fd = ::open(path, O_RDONLY);
    if (f.fd < 0) { f.error = std::string("cannot open ") + path; return TRR_E_IO; }
    struct stat st;
    if (fstat(f.fd, &st) != 0) { f.error = "fstat failed"; return TRR_E_IO; }
    f.size = (uint64_t)st.st_size;
    uint64_t off = 0;
    unsigned char h[96];
    while (off < f.size) {
        // magic, length of the version string + 1, XDR string (length, bytes padded to 4)
        if (off + 12 > f.size || !pread_all(f.fd, h, 12, off)) { f.error = "short read in frame header"; return f.frames.empty() ? TRR_E_FORMAT : TRR_E_IO; }
        if (be32(h) != 1993u) { f.error = "bad magic number"; return TRR_E_FORMAT; }
        const uint32_t slen = be32(h + 4), sl = be32(h + 8);
        if (slen != 13u || sl != 12u) { f.error = "not a GMX_trn_file header"; return TRR_E_FORMAT; }
        uint64_t p = off + 12;
        if (p + 12 + 52 > f.size || !pread_all(f.fd, h, 12 + 52, p)) { f.error = "truncated frame header"; return TRR_E_FORMAT; }
        if (memcmp(h, "GMX_trn_file", 12) != 0) { f.error = "not a GMX_trn_file header"; return TRR_E_FORMAT; }
        const unsigned char *q = h + 12;
        const uint32_t ir = be32(q), e = be32(q + 4), box = be32(q + 8), vir = be32(q + 12), pres = be32(q + 16), top = be32(q + 20), sym = be32(q + 24);
        const uint32_t xs = be32(q + 28), vs = be32(q + 32), fs = be32(q + 36), natoms = be32(q + 40);
        FrameIndex fi; memset(&fi, 0, sizeof fi);
        fi.offset = off;
        fi.step = (int32_t)be32(q + 44);
        p += 12 + 52;
        if (f.frames.empty()) f.natoms = natoms;
        else if (natoms != f.natoms) { f.error = "number of atoms changes between frames"; return TRR_E_FORMAT; }
        // one precision per frame, told by the first section present (xdrfile_trr.c nFloatSize)
        uint64_t rs = 0;
        const uint64_t n3 = (uint64_t)natoms * 3;
        if (box) rs = box / 9; else if (xs && n3) rs = xs / n3; else if (vs && n3) rs = vs / n3; else if (fs && n3) rs = fs / n3;
        if (rs != 4 && rs != 8) { f.error = "cannot tell the precision of the frame"; return TRR_E_FORMAT; }
        fi.real_size = (uint32_t)rs;
        if ((box && box != 9 * rs) || (vir && vir != 9 * rs) || (pres && pres != 9 * rs) || (xs && xs != n3 * rs) || (vs && vs != n3 * rs) || (fs && fs != n3 * rs)) {
            f.error = "section sizes do not match the atom count"; return TRR_E_FORMAT;
        }
        if (p + 2 * rs > f.size || !pread_all(f.fd, h, 2 * (size_t)rs, p)) { f.error = "truncated frame header"; return TRR_E_FORMAT; }
        if (rs == 4) { fi.time = bef(h); fi.lambda = bef(h + 4); } else { fi.time = (float)bed(h); fi.lambda = (float)bed(h + 8); }
        p += 2 * rs;
        p += (uint64_t)ir + e;            // (never written by GROMACS; skipped like top / sym below)
        if (box) {
            unsigned char b[72];
            if (p + box > f.size || !pread_all(f.fd, b, box, p)) { f.error = "truncated box"; return TRR_E_FORMAT; }
            for (int k = 0; k < 9; ++k) fi.box[k] = rs == 4 ? bef(b + 4 * k) : (float)bed(b + 8 * k);
            fi.has_box = 1;
            p += box;
        }
        p += (uint64_t)vir + pres + top + sym;
        if (xs) { fi.x_off = p; p += xs; }
        if (vs) { fi.v_off = p; p += vs; }
        if (fs) { fi.f_off = p; p += fs; }
        if (p > f.size) { f.error = "truncated frame"; return TRR_E_FORMAT; }
        f.frames.push_back(fi);
        off = p;
    }
    return TRR_OK;
}

// one section (positions, velocities or forces) of a frame as f32[natoms][3]; a missing section reads as zeros
inline int read_section(const File &f, const FrameIndex &fi, uint64_t sec_off, float *out, std::vector<unsigned char> &scratch) {
    const size_t n3 = (size_t)f.natoms * 3;
    if (!out) return TRR_OK;
    if (sec_off == 0) { memset(out, 0, n3 * sizeof(float)); return TRR_OK; }
    const size_t bytes = n3 * fi.real_size;
    scratch.resize(bytes);
    if (!pread_all(f.fd, scratch.data(), bytes, sec_off)) return TRR_E_IO;
    const unsigned char *p = scratch.data();
    if (fi.real_size == 4) for (size_t k = 0; k < n3; ++k) out[k] = bef(p + 4 * k);
    else for (size_t k = 0; k < n3; ++k) out[k] = (float)bed(p + 8 * k);
    return TRR_OK;
}

// ---- writer: what the reference's TrrWriter produces through xdrfile's write_trr (src/io/trr_io.rs:441-520,
// xdrfile_trr.c do_trn / do_trnheader / do_htrn): single precision, box + positions + velocities + forces sections (a NULL
// array omits its section), nre = 0, no virial / pressure.  A position without value (NaN in x) is written as zeros.
inline void put_be32(std::vector<unsigned char> &o, uint32_t v) { o.push_back((unsigned char)(v >> 24)); o.push_back((unsigned char)(v >> 16)); o.push_back((unsigned char)(v >> 8)); o.push_back((unsigned char)v); }
inline void put_bef(std::vector<unsigned char> &o, float f) { uint32_t u; memcpy(&u, &f, 4); put_be32(o, u); }
inline void serialise_frame(std::vector<unsigned char> &o, uint32_t natoms, int32_t step, float time, float lambda, const float *box_rows,
                            const float *x, const float *v, const float *f) {
    o.clear();
    o.reserve(96 + 36 + (size_t)natoms * 36);
    put_be32(o, 1993u); put_be32(o, 13u); put_be32(o, 12u);
    o.insert(o.end(), (const unsigned char *)"GMX_trn_file", (const unsigned char *)"GMX_trn_file" + 12);
    const uint32_t n12 = natoms * 12u;
    put_be32(o, 0); put_be32(o, 0); put_be32(o, box_rows ? 36u : 0u); put_be32(o, 0); put_be32(o, 0); put_be32(o, 0); put_be32(o, 0);   // ir e box vir pres top sym
    put_be32(o, x ? n12 : 0u); put_be32(o, v ? n12 : 0u); put_be32(o, f ? n12 : 0u);
    put_be32(o, natoms); put_be32(o, (uint32_t)step); put_be32(o, 0u);
    put_bef(o, time); put_bef(o, lambda);
    if (box_rows) for (int k = 0; k < 9; ++k) put_bef(o, box_rows[k]);
    const size_t n3 = (size_t)natoms * 3;
    if (x) for (size_t k = 0; k < n3; ++k) put_bef(o, (x[3 * (k / 3)] != x[3 * (k / 3)]) ? 0.0f : x[k]);
    if (v) for (size_t k = 0; k < n3; ++k) put_bef(o, v[k]);
    if (f) for (size_t k = 0; k < n3; ++k) put_bef(o, f[k]);
}

}  // namespace grtr
