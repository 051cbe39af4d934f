// Sanitizer fuzz driver for the host-side parsers of the library (no GPU, no HIP): the xtc decoder / skimmer / encoder
// (groan_rs_amd/csrc/gr_xtc.h) and the gro / ndx readers (gr_textio.h).  Built with -fsanitize=address,undefined by
// tests/cpp/Makefile and run by tests/test_fuzz_host.py: corrupted, truncated and hostile inputs must come back as error
// codes -- never a crash, an out-of-bounds access, or a hang.  (The reference answers such files with ReadTrajError /
// ParseGroError / ParseNdxError values, src/errors.rs; GPU sanitizers are not available on the pool, so the device path
// relies on these host checks: k_xtc_unpack only ever reads bit ranges the skim has validated.)
//
//   fuzz_host <iterations> <seed> <xtc files...> -- <gro / ndx files...>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../groan_rs_amd/csrc/gr_xtc.h"
#include "../../groan_rs_amd/csrc/gr_textio.h"
#include "../../groan_rs_amd/csrc/gr_trr.h"

static std::vector<unsigned char> slurp(const char *path) {
    std::vector<unsigned char> v;
    FILE *fp = fopen(path, "rb");
    if (!fp) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    unsigned char buf[65536]; size_t r;
    while ((r = fread(buf, 1, sizeof buf, fp)) > 0) v.insert(v.end(), buf, buf + r);
    fclose(fp);
    return v;
}
static void spit(const std::string &path, const std::vector<unsigned char> &v) {
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
    if (!v.empty()) fwrite(v.data(), 1, v.size(), fp);
    fclose(fp);
}

struct Tally { long ok = 0, rejected = 0; };

// the skim without its fast stretch: one group at a time, every check per group -- the statement the library's walk is tested against
static int skim_plain(const unsigned char *stream, const grx::FrameIndex &fi, uint32_t n, std::vector<grx::Checkpoint> &cps) {
    uint32_t sz[3];
    for (int k = 0; k < 3; ++k) { sz[k] = (uint32_t)fi.maxint[k] - (uint32_t)fi.minint[k] + 1u; if (!sz[k]) return grx::XTC_E_FORMAT; }
    int large = 0;
    if ((sz[0] | sz[1] | sz[2]) > 0xffffffu) for (int k = 0; k < 3; ++k) large += std::min(32, grx::bit_length(sz[k]));
    else large = grx::bit_length((unsigned __int128)sz[0] * sz[1] * sz[2]);
    if (fi.nbytes >= (1ull << 29)) return grx::XTC_E_FORMAT;
    const uint64_t limit = (fi.nbytes + 8) * 8;
    cps.clear();
    int smallidx = fi.smallidx, run = 0;
    uint64_t bitpos = 0;
    uint32_t i = 0;
    while (i < n) {
        if (i >= cps.size() * GR_XTC_CP_ATOMS) cps.push_back(grx::Checkpoint{ (uint32_t)bitpos, i, (uint32_t)smallidx | ((uint32_t)run << 8) });
        bitpos += (uint64_t)large;
        if (bitpos + 6 > limit) return grx::XTC_E_FORMAT;
        ++i;
        int change = 0;
        if (grx::peek_bits(stream, bitpos, 1)) { run = (int)grx::peek_bits(stream, bitpos + 1, 5); bitpos += 6; change = run % 3 - 1; run -= run % 3; }
        else bitpos += 1;
        if ((uint64_t)i + (uint64_t)(run / 3) > n) return grx::XTC_E_FORMAT;
        bitpos += (uint64_t)(run / 3) * (uint64_t)smallidx; i += (uint32_t)(run / 3);
        if (bitpos > limit) return grx::XTC_E_FORMAT;
        smallidx += change;
        if (smallidx < grx::kFirstIdx || smallidx >= grx::kLastIdx) return grx::XTC_E_FORMAT;
    }
    while (cps.size() < (n + GR_XTC_CP_ATOMS - 1) / GR_XTC_CP_ATOMS) cps.push_back(grx::Checkpoint{ (uint32_t)bitpos, i, (uint32_t)smallidx });
    return grx::XTC_OK;
}

// open + decode + skim every frame of a (possibly corrupt) xtc file
static void run_xtc(const std::string &path, Tally &t) {
    grx::File f;
    const int st = grx::open_file(f, path.c_str());
    if (st != grx::XTC_OK) { if (f.fd >= 0) close(f.fd); t.rejected++; return; }
    if (f.natoms > (1u << 22)) { close(f.fd); t.rejected++; return; }   // a corrupted atom count: the caller's buffer decides, not the file
    std::vector<float> xyz(3 * (size_t)f.natoms + 3);
    std::vector<unsigned char> scratch;
    std::vector<grx::Checkpoint> cps;
    bool all = true;
    for (const grx::FrameIndex &fi : f.frames) {
        const int r = grx::decode_frame(f, fi, xyz.data(), scratch);
        if (r != grx::XTC_OK) all = false;
        if (f.natoms > 9) {
            scratch.resize((size_t)fi.nbytes + 16);
            if (grx::pread_all(f.fd, scratch.data(), (size_t)fi.nbytes, fi.data_offset)) {
                memset(scratch.data() + fi.nbytes, 0, 16);
                grx::FrameDesc d; memset(&d, 0, sizeof d);
                {   // partial walks (GroupXtcReader): any stop atom, any prefix length of the stream -- an error code or a valid table, never a crash
                    std::vector<grx::Checkpoint> pc; grx::FrameDesc pd; memset(&pd, 0, sizeof pd);
                    const uint32_t stop = (uint32_t)((fi.nbytes * 2654435761ull + fi.step) % (f.natoms + 1));
                    const uint64_t have = fi.nbytes ? (fi.nbytes * 40503ull + 7) % (fi.nbytes + 1) : 0;
                    const int ps = grx::skim_frame(scratch.data(), fi, f.natoms, pd, pc, stop, have);
                    if (ps == grx::XTC_OK && (pc.size() != (stop + GR_XTC_CP_ATOMS - 1) / GR_XTC_CP_ATOMS || pd.n_end > f.natoms || pd.nbytes > fi.nbytes)) { fprintf(stderr, "partial skim table\n"); abort(); }
                    std::vector<float> part(3 * (size_t)stop + 3);
                    std::vector<unsigned char> sc2;
                    (void)grx::decode_frame_prefix(f, fi, stop, part.data(), sc2);
                }
                const int s = grx::skim_frame(scratch.data(), fi, f.natoms, d, cps);
                {   // the fast stretch changes nothing: same verdict, same table as the group-by-group walk
                    std::vector<grx::Checkpoint> ref;
                    const int sp = skim_plain(scratch.data(), fi, f.natoms, ref);
                    if (sp != s) { fprintf(stderr, "skim verdict %d vs plain walk %d\n", s, sp); abort(); }
                    if (s == grx::XTC_OK && (ref.size() != cps.size() || memcmp(ref.data(), cps.data(), ref.size() * sizeof(grx::Checkpoint)) != 0)) { fprintf(stderr, "skim table differs from the plain walk\n"); abort(); }
                }
                // whatever the decoder accepts the skimmer must accept, with one checkpoint per 32 atoms inside the stream
                if (r == grx::XTC_OK && s == grx::XTC_OK) {
                    if (cps.size() != (f.natoms + GR_XTC_CP_ATOMS - 1) / GR_XTC_CP_ATOMS) { fprintf(stderr, "checkpoint count\n"); abort(); }
                    for (const grx::Checkpoint &c : cps) if ((uint64_t)c.bitpos > (fi.nbytes + 8) * 8 || c.atom > f.natoms) { fprintf(stderr, "checkpoint outside the stream\n"); abort(); }
                }
            }
        }
    }
    close(f.fd);
    if (all) t.ok++; else t.rejected++;
}

// open + read every section of every frame of a (possibly corrupt) trr file
static void run_trr(const std::string &path, Tally &t) {
    grtr::File f;
    const int st = grtr::open_file(f, path.c_str());
    if (st != grtr::TRR_OK) { if (f.fd >= 0) close(f.fd); t.rejected++; return; }
    if (f.natoms > (1u << 22)) { close(f.fd); t.rejected++; return; }
    std::vector<float> x(3 * (size_t)f.natoms + 3);
    std::vector<unsigned char> scratch;
    bool all = true;
    for (const grtr::FrameIndex &fi : f.frames)
        for (uint64_t off : { fi.x_off, fi.v_off, fi.f_off })
            if (grtr::read_section(f, fi, off, x.data(), scratch) != grtr::TRR_OK) all = false;
    close(f.fd);
    if (all) t.ok++; else t.rejected++;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: fuzz_host <iterations> <seed> <xtc...> -- <gro/ndx...>\n"); return 2; }
    const int iters = atoi(argv[1]);
    std::mt19937_64 rng((uint64_t)atoll(argv[2]));
    std::vector<std::string> xtcs, texts, trrs;
    bool second = false;
    for (int k = 3; k < argc; ++k) {
        if (!strcmp(argv[k], "--")) { second = true; continue; }
        const std::string a = argv[k];
        if (a.size() > 4 && a.substr(a.size() - 4) == ".trr") trrs.push_back(a); else (second ? texts : xtcs).push_back(a);
    }
    const char *tmpdir = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
    const std::string tmp = std::string(tmpdir) + "/fuzz_host_" + std::to_string((long)getpid());
    auto rnd = [&](uint64_t n) { return n ? rng() % n : 0; };

    // ---- xtc: the pristine files decode; mutants never crash
    Tally tx;
    for (const std::string &p : xtcs) {
        const std::vector<unsigned char> orig = slurp(p.c_str());
        { Tally t0; run_xtc(p, t0); if (t0.ok != 1) { fprintf(stderr, "pristine %s rejected\n", p.c_str()); return 1; } }
        for (int it = 0; it < iters; ++it) {
            std::vector<unsigned char> m = orig;
            switch (rnd(6)) {
            case 0: m.resize(rnd(m.size() + 1)); break;                                                      // truncation anywhere
            case 1: for (int k = 0, n = 1 + (int)rnd(8); k < n && !m.empty(); ++k) m[rnd(m.size())] ^= (unsigned char)(1u << rnd(8)); break;   // bit flips
            case 2: for (int k = 0, n = 1 + (int)rnd(4); k < n && m.size() >= 4; ++k) { size_t o = rnd(m.size() - 3); uint32_t v = (uint32_t)rng(); memcpy(&m[o], &v, 4); } break;
            case 3: if (m.size() >= 96) { size_t o = 4 * rnd(24); static const uint32_t evil[] = { 0u, 0xffffffffu, 0x7fffffffu, 0x80000000u, 1u, 0x00ffffffu, 0x01000000u };   // header words
                        uint32_t v = evil[rnd(7)]; unsigned char b[4] = { (unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v }; memcpy(&m[o], b, 4); } break;
            case 4: if (m.size() > 100) { size_t o = 92 + rnd(m.size() - 92), n = 1 + rnd(64); for (size_t k = o; k < m.size() && k < o + n; ++k) m[k] = (unsigned char)rng(); } break;   // noise in the bit stream
            default: { size_t o = rnd(m.size() + 1), n = rnd(4096); m.insert(m.begin() + (long)o, n, (unsigned char)rnd(256)); } break;       // inserted run
            }
            spit(tmp + ".xtc", m);
            run_xtc(tmp + ".xtc", tx);
        }
    }
    // ---- crafted magic-2023 frames: that variant carries a 64-bit byte count, taken from the file.  The first frame of every
    // pristine file is rewritten as magic 2023 with byte counts that wrap when rounded up to 4, when added to the file
    // offset, or when they size a buffer (2^64 - 1 ... ) -- all must be refused at open; the honest count must still decode.
    long crafted_ok = 0, crafted_rej = 0;
    for (const std::string &p : xtcs) {
        const std::vector<unsigned char> orig = slurp(p.c_str());
        if (orig.size() < 96 || grx::be32(orig.data() + 4) <= 9) continue;
        const uint32_t honest = grx::be32(orig.data() + 88);
        const uint64_t evil[] = { (uint64_t)honest, ~0ull, ~0ull - 1, ~0ull - 2, ~0ull - 3, ~0ull - 127, 1ull << 63, (1ull << 63) - 1, 1ull << 32, (1ull << 32) | honest,
                                  (uint64_t)orig.size(), (uint64_t)orig.size() + 4, ~0ull - (uint64_t)orig.size(), 0ull - 96ull, 0ull - 92ull, (1ull << 29), (1ull << 29) - 1 };
        for (uint64_t nb : evil) {
            std::vector<unsigned char> m = orig;
            m[0] = 0; m[1] = 0; m[2] = 0x07; m[3] = 0xE7;                       // magic 2023
            unsigned char b[8]; for (int k = 0; k < 8; ++k) b[k] = (unsigned char)(nb >> (56 - 8 * k));
            m.insert(m.begin() + 88, 4, (unsigned char)0);
            memcpy(&m[88], b, 8);
            spit(tmp + ".xtc", m);
            Tally t; run_xtc(tmp + ".xtc", t);
            if (nb == honest) { if (t.ok != 1) { fprintf(stderr, "honest magic-2023 frame rejected (%s)\n", p.c_str()); return 1; } crafted_ok++; }
            else { if (t.ok != 0) { fprintf(stderr, "a frame with byte count %llu was accepted (%s)\n", (unsigned long long)nb, p.c_str()); return 1; } crafted_rej++; }
        }
    }
    printf("crafted 2023 frames: %ld decoded, %ld refused\n", crafted_ok, crafted_rej);
    // ---- trr: pristine files read; mutants (same mutators) never crash
    Tally tr;
    for (const std::string &p : trrs) {
        const std::vector<unsigned char> orig = slurp(p.c_str());
        { Tally t0; run_trr(p, t0); if (t0.ok != 1) { fprintf(stderr, "pristine %s rejected\n", p.c_str()); return 1; } }
        for (int it = 0; it < iters; ++it) {
            std::vector<unsigned char> m = orig;
            switch (rnd(5)) {
            case 0: m.resize(rnd(m.size() + 1)); break;
            case 1: for (int k = 0, n = 1 + (int)rnd(8); k < n && !m.empty(); ++k) m[rnd(m.size())] ^= (unsigned char)(1u << rnd(8)); break;
            case 2: if (m.size() >= 96) { size_t o = 4 * rnd(24); static const uint32_t evil[] = { 0u, 0xffffffffu, 0x7fffffffu, 0x80000000u, 1u, 12u, 0x01000000u };
                        uint32_t v = evil[rnd(7)]; unsigned char b[4] = { (unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v }; memcpy(&m[o], b, 4); } break;
            case 3: for (int k = 0, n = 1 + (int)rnd(4); k < n && m.size() >= 4; ++k) { size_t o = rnd(m.size() - 3); uint32_t v = (uint32_t)rng(); memcpy(&m[o], &v, 4); } break;
            default: { size_t o = rnd(m.size() + 1), n = rnd(4096); m.insert(m.begin() + (long)o, n, (unsigned char)rnd(256)); } break;
            }
            spit(tmp + ".trr", m);
            run_trr(tmp + ".trr", tr);
        }
    }
    // ---- encoder: hostile coordinates (NaN, infinities, 1e30, denormals) round-trip or are rejected, never crash
    long enc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t n = 10 + (uint32_t)rnd(300);
        std::vector<float> xyz(3 * (size_t)n);
        std::uniform_real_distribution<float> u(-5.0f, 5.0f);
        for (float &v : xyz) v = u(rng);
        static const float evil[] = { NAN, INFINITY, -INFINITY, 1e30f, -1e30f, 1e-40f, 2147483.0f, -2147484.0f, 0.0f };
        for (int k = 0, ne = (int)rnd(6); k < ne; ++k) xyz[rnd(xyz.size())] = evil[rnd(9)];
        const float box[9] = { 5, 0, 0, 0, 5, 0, 0, 0, 5 };
        std::vector<unsigned char> out; std::vector<int> ints; grx::EncodedFrame e;
        static const float precs[] = { 1000.0f, 100.0f, 1e6f, 0.0f, -1.0f, NAN };
        if (!grx::serialise_frame(out, n, it, 0.5f * it, box, xyz.data(), precs[rnd(6)], e, ints)) continue;   // refused: does not fit the format
        spit(tmp + ".xtc", out);
        Tally t; run_xtc(tmp + ".xtc", t);
        if (t.ok != 1) { fprintf(stderr, "an accepted encode does not decode\n"); abort(); }
        enc += t.ok;
    }
    // ---- gro / ndx text: mutants parse or are rejected with a code
    Tally tt;
    for (const std::string &p : texts) {
        const std::vector<unsigned char> orig = slurp(p.c_str());
        const bool ndx = p.size() > 4 && p.substr(p.size() - 4) == ".ndx";
        for (int it = 0; it < iters; ++it) {
            std::vector<unsigned char> m = orig;
            switch (rnd(4)) {
            case 0: m.resize(rnd(m.size() + 1)); break;
            case 1: for (int k = 0, n = 1 + (int)rnd(6); k < n && !m.empty(); ++k) m[rnd(m.size())] = (unsigned char)(" \n\t-+.eE0123456789[]xyz\0\xff"[rnd(26)]); break;
            case 2: if (!m.empty()) { size_t o = rnd(m.size()), n = rnd(200); m.erase(m.begin() + (long)o, m.begin() + (long)std::min(m.size(), o + n)); } break;
            default: { size_t o = rnd(m.size() + 1); const char *junk[] = { "99999999999999999999999", "-1", "nan", "\n\n\n", "[", "] [ ]", "1e400", "    " }; const char *j = junk[rnd(8)]; m.insert(m.begin() + (long)o, j, j + strlen(j)); } break;
            }
            spit(tmp + (ndx ? ".ndx" : ".gro"), m);
            std::string detail;
            if (ndx) { std::vector<grt::NdxGroup> g; uint64_t bad = 0; const int r = grt::read_ndx((tmp + ".ndx").c_str(), 1 + rnd(100000), g, detail, bad); (r == grt::P_OK ? tt.ok : tt.rejected)++; }
            else { grt::Structure s; const int r = grt::read_gro((tmp + ".gro").c_str(), s, detail); (r == grt::P_OK ? tt.ok : tt.rejected)++; }
        }
    }
    remove((tmp + ".trr").c_str());
    remove((tmp + ".xtc").c_str()); remove((tmp + ".gro").c_str()); remove((tmp + ".ndx").c_str());
    printf("xtc mutants: %ld decoded, %ld rejected; hostile encodes that decode: %ld of %d; text mutants: %ld parsed, %ld rejected; trr mutants: %ld read, %ld rejected\n",
           tx.ok, tx.rejected, enc, iters, tt.ok, tt.rejected, tr.ok, tr.rejected);
    return 0;
}
