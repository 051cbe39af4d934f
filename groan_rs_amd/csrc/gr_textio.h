// gr_textio.h -- the text front end in front of the path: gro structures and ndx index groups (host code).
// Reference: read_gro / line_as_atom / line_as_box (src/io/gro_io/structure.rs:120-231, src/io/gro_io/mod.rs:21-72) and
// Groups::from_ndx / parse_group_name / parse_ndx_line (src/io/ndx_io.rs:104-230).  Same fixed columns, the same accepted
// number spellings as Rust's `str::parse` (no inner blanks, no comma, no hex, `nan` / `inf` parse but are rejected as
// InvalidFloat), the same error variant and payload for every malformed file of the reference's test suite.
#pragma once
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace grt {

enum { P_OK = 0, P_FILE_NOT_FOUND, P_LINE_NOT_FOUND, P_PARSE_LINE, P_PARSE_ATOM_LINE, P_PARSE_BOX_LINE, P_UNSUPPORTED_BOX, P_INVALID_FLOAT,
       P_PARSE_GROUP_NAME, P_INVALID_ATOM_INDEX };

inline std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
inline std::string trim_end(const std::string &s) { size_t b = s.size(); while (b > 0 && isspace((unsigned char)s[b - 1])) --b; return s.substr(0, b); }

// str::parse::<usize>: optional '+', then digits only
inline bool parse_usize(const std::string &t, uint64_t &out) {
    size_t k = 0;
    if (k < t.size() && t[k] == '+') ++k;
    if (k >= t.size()) return false;
    uint64_t v = 0;
    for (; k < t.size(); ++k) {
        if (t[k] < '0' || t[k] > '9') return false;
        const uint64_t nv = v * 10 + (uint64_t)(t[k] - '0');
        if (nv < v) return false;
        v = nv;
    }
    out = v;
    return true;
}
// str::parse::<f32>: decimal or exponent form, "inf" / "infinity" / "nan" (any case), optional sign; nothing else
inline bool parse_f32(const std::string &t, float &out) {
    if (t.empty()) return false;
    size_t k = 0;
    if (t[k] == '+' || t[k] == '-') ++k;
    if (k >= t.size()) return false;
    std::string low;
    for (size_t q = k; q < t.size(); ++q) low.push_back((char)tolower((unsigned char)t[q]));
    if (low == "inf" || low == "infinity") { out = (t[0] == '-') ? -INFINITY : INFINITY; return true; }
    if (low == "nan") { out = NAN; return true; }
    bool digits = false, dot = false, exp = false;
    for (size_t q = k; q < t.size(); ++q) {
        const char ch = t[q];
        if (ch >= '0' && ch <= '9') { digits = true; continue; }
        if (ch == '.' && !dot && !exp) { dot = true; continue; }
        if ((ch == 'e' || ch == 'E') && digits && !exp) {
            exp = true;
            if (q + 1 < t.size() && (t[q + 1] == '+' || t[q + 1] == '-')) ++q;
            if (q + 1 >= t.size()) return false;
            continue;
        }
        return false;
    }
    if (!digits) return false;
    char *end = nullptr;
    out = strtof(t.c_str(), &end);
    return end && *end == '\0';
}

struct Atom { uint64_t resid, atomid; std::string resname, atomname; float pos[3]; float vel[3]; bool has_vel; };
struct Structure { std::string title; std::vector<Atom> atoms; float box9[9]; bool has_box = false; };

// BufRead::read_line: returns false at end of file; the trailing '\n' (only) is removed
inline bool read_line(FILE *fp, std::string &line) {
    line.clear();
    int ch; bool any = false;
    while ((ch = fgetc(fp)) != EOF) { any = true; if (ch == '\n') return true; line.push_back((char)ch); }
    return any;
}

inline int line_as_atom(const std::string &line, Atom &a, std::string &detail) {
    auto bad = [&](int code) { detail = line; return code; };
    if (line.size() < 44) return bad(P_PARSE_ATOM_LINE);
    if (!parse_usize(trim(line.substr(0, 5)), a.resid)) return bad(P_PARSE_ATOM_LINE);
    a.resname = trim(line.substr(5, 5)); if (a.resname.empty()) return bad(P_PARSE_ATOM_LINE);
    a.atomname = trim(line.substr(10, 5)); if (a.atomname.empty()) return bad(P_PARSE_ATOM_LINE);
    if (!parse_usize(trim(line.substr(15, 5)), a.atomid)) return bad(P_PARSE_ATOM_LINE);
    for (int i = 0; i < 3; ++i) {
        if (!parse_f32(trim(line.substr(20 + 8 * i, 8)), a.pos[i])) return bad(P_PARSE_ATOM_LINE);
        if (!std::isfinite(a.pos[i])) return bad(P_INVALID_FLOAT);
    }
    a.has_vel = false; a.vel[0] = a.vel[1] = a.vel[2] = NAN;
    if (trim_end(line).size() >= 68) {
        for (int i = 0; i < 3; ++i) {
            if (line.size() < (size_t)(44 + 8 * i + 8)) return bad(P_PARSE_ATOM_LINE);
            if (!parse_f32(trim(line.substr(44 + 8 * i, 8)), a.vel[i])) return bad(P_PARSE_ATOM_LINE);
            if (!std::isfinite(a.vel[i])) return bad(P_INVALID_FLOAT);
        }
        a.has_vel = true;
    }
    return P_OK;
}

inline int line_as_box(const std::string &line, float box9[9], std::string &detail) {
    for (int k = 0; k < 9; ++k) box9[k] = 0.0f;
    int i = 0; size_t p = 0;
    while (true) {
        while (p < line.size() && isspace((unsigned char)line[p])) ++p;
        if (p >= line.size()) break;
        size_t q = p;
        while (q < line.size() && !isspace((unsigned char)line[q])) ++q;
        float v;
        if (i >= 9 || !parse_f32(line.substr(p, q - p), v)) { detail = line; return P_PARSE_BOX_LINE; }   // (a 10th value panics in the reference)
        box9[i++] = v; p = q;
    }
    if (i != 3 && i != 9) { detail = line; return P_PARSE_BOX_LINE; }
    if (box9[3] != 0.0f || box9[4] != 0.0f || box9[6] != 0.0f) { detail = line; return P_UNSUPPORTED_BOX; }
    return P_OK;
}

inline int read_gro(const char *path, Structure &s, std::string &detail) {
    FILE *fp = fopen(path, "rb");
    if (!fp) { detail = path; return P_FILE_NOT_FOUND; }
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{ fp };
    std::string line;
    if (!read_line(fp, line)) { detail = path; return P_LINE_NOT_FOUND; }
    s.title = trim(line);
    if (!read_line(fp, line)) { detail = path; return P_LINE_NOT_FOUND; }
    uint64_t n = 0;
    if (!parse_usize(trim(line), n)) { detail = trim(line); return P_PARSE_LINE; }
    s.atoms.clear(); s.atoms.reserve((size_t)n);
    for (uint64_t k = 0; k < n; ++k) {
        read_line(fp, line);                       // at end of file the line stays empty and fails as an atom line
        Atom a;
        const int rc = line_as_atom(line, a, detail);
        if (rc != P_OK) return rc;
        s.atoms.push_back(a);
    }
    read_line(fp, line);
    const int rc = line_as_box(line, s.box9, detail);
    if (rc != P_OK) return rc;
    bool zero = true;
    for (int k = 0; k < 9; ++k) if (s.box9[k] != 0.0f) zero = false;
    s.has_box = !zero;                             // SimBox::is_zero -> no box (structure.rs:159-162)
    return P_OK;
}

struct NdxGroup { std::string name; std::vector<uint64_t> indices; };   // 0-based, file order, duplicates kept (from_indices sorts / dedups)

inline int read_ndx(const char *path, uint64_t n_atoms, std::vector<NdxGroup> &groups, std::string &detail, uint64_t &bad_index) {
    FILE *fp = fopen(path, "rb");
    if (!fp) { detail = path; return P_FILE_NOT_FOUND; }
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{ fp };
    groups.clear();
    std::string line, current;
    std::vector<uint64_t> indices;
    while (read_line(fp, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();    // BufRead::lines strips "\r\n" too
        if (trim(line).empty()) continue;
        if (line.find('[') != std::string::npos && line.find(']') != std::string::npos) {
            if (!current.empty()) groups.push_back(NdxGroup{ current, indices });
            indices.clear();
            std::string name;
            for (char ch : line) if (ch != '[' && ch != ']') name.push_back(ch);
            name = trim(name);
            if (name.empty()) { detail = line; return P_PARSE_GROUP_NAME; }
            current = name;
        } else {
            size_t p = 0;
            while (true) {
                while (p < line.size() && isspace((unsigned char)line[p])) ++p;
                if (p >= line.size()) break;
                size_t q = p;
                while (q < line.size() && !isspace((unsigned char)line[q])) ++q;
                uint64_t id;
                if (!parse_usize(line.substr(p, q - p), id)) { detail = line; return P_PARSE_LINE; }
                if (id == 0 || id > n_atoms) { bad_index = id; detail = std::to_string(id); return P_INVALID_ATOM_INDEX; }
                indices.push_back(id - 1);
                p = q;
            }
        }
    }
    if (!current.empty()) groups.push_back(NdxGroup{ current, indices });
    return P_OK;
}

}  // namespace grt
