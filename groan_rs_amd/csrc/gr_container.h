// gr_container.h -- host-side AtomContainer logic (selection = sorted, merged, inclusive index blocks).
// Integer work, bit-exact with src/structures/container.rs of the reference, including its quirks:
//   * from_indices never range-checks the FIRST (smallest) index, and the first out-of-range index
//     closes the current block at n_atoms-1 and stops the scan (container.rs:69-73);
//   * from_blocks treats `current_end == 0 && current_start != 0` as "no current block" (:188).
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace grc {

typedef std::pair<uint64_t, uint64_t> Block;   // inclusive [start, end]; ordered by start, then end (container.rs:337-351)

// container.rs:51-104
inline std::vector<Block> from_indices(std::vector<uint64_t> idx, uint64_t n_atoms) {
    std::vector<Block> blocks;
    if (idx.empty()) return blocks;
    if (!std::is_sorted(idx.begin(), idx.end())) std::sort(idx.begin(), idx.end());   // (the geometry selection hands in a million sorted indices)
    uint64_t start = idx[0], end = idx[0];
    for (size_t k = 1; k < idx.size(); ++k) {
        const uint64_t index = idx[k];
        if (index >= n_atoms) { end = n_atoms - 1; break; }
        if (index == end) continue;
        if (index == end + 1) { end = index; }
        else { blocks.push_back(Block(start, end)); start = index; end = index; }
    }
    blocks.push_back(Block(start, end));
    return blocks;
}

// container.rs:167-215
inline std::vector<Block> from_blocks(std::vector<Block> in) {
    std::vector<Block> out;
    if (in.empty()) return out;
    std::sort(in.begin(), in.end());
    uint64_t cur_s = UINT64_MAX, cur_e = 0;
    for (const Block &b : in) {
        if (b.first > cur_e + 1 || (cur_e == 0 && cur_s != 0)) {
            if (cur_s != UINT64_MAX) out.push_back(Block(cur_s, cur_e));
            cur_s = b.first; cur_e = b.second;
        } else if (b.second > cur_e) {
            cur_e = b.second;
        }
    }
    if (cur_s != UINT64_MAX) out.push_back(Block(cur_s, cur_e));
    return out;
}

// container.rs:122-153
inline std::vector<Block> from_ranges(const uint64_t *s, const uint64_t *e, size_t n, uint64_t n_atoms) {
    std::vector<Block> blocks;
    if (n_atoms == 0) return blocks;
    for (size_t k = 0; k < n; ++k) {
        const uint64_t start = s[k];
        const uint64_t end = e[k] < n_atoms ? e[k] : n_atoms - 1;
        if (start > end) continue;
        blocks.push_back(Block(start, end));
    }
    return from_blocks(blocks);
}

// container.rs:161-165
inline uint64_t n_atoms(const std::vector<Block> &b) {
    uint64_t t = 0;
    for (const Block &x : b) t += x.second - x.first + 1;
    return t;
}

// iteration order of next_index, container.rs:381-411
inline std::vector<uint64_t> expand(const std::vector<Block> &b) {
    std::vector<uint64_t> out;
    out.reserve((size_t)n_atoms(b));
    size_t cur_block = 0;
    uint64_t cur_atom = 0;
    while (cur_block < b.size()) {
        if (cur_atom < b[cur_block].first) { cur_atom = b[cur_block].first + 1; out.push_back(b[cur_block].first); continue; }
        if (cur_atom <= b[cur_block].second) { out.push_back(cur_atom); cur_atom += 1; continue; }
        cur_block += 1;
    }
    return out;
}

// container.rs:241-258
inline bool isin(const std::vector<Block> &b, uint64_t index) {
    for (const Block &x : b) {
        if (index < x.first) return false;
        if (index <= x.second) return true;
    }
    return false;
}

// container.rs:268-276
inline std::vector<Block> set_union(const std::vector<Block> &a, const std::vector<Block> &b) {
    std::vector<Block> all(a);
    all.insert(all.end(), b.begin(), b.end());
    return from_blocks(all);
}

// container.rs:278-291
inline std::vector<Block> set_intersection(const std::vector<Block> &a, const std::vector<Block> &b) {
    std::vector<uint64_t> keep;
    uint64_t mx = 0;
    for (uint64_t i : expand(a)) if (isin(b, i)) { keep.push_back(i); if (i > mx) mx = i; }
    if (keep.empty()) return std::vector<Block>();
    return from_indices(keep, mx + 1);
}

// A block list a device kernel may index with: every block ordered and inside [0, n_atoms).  from_indices keeps the
// reference's quirk of never range-checking the smallest index (container.rs:66; the reference then panics on the first
// access), so `[n_atoms]` yields the block (n, n) and `[n, n + 1]` the block (n, n - 1): such lists must never reach a kernel.
inline bool valid_for(const std::vector<Block> &b, uint64_t n_atoms, uint64_t *bad_index) {
    for (const Block &x : b) {
        if (x.first > x.second || x.second >= n_atoms) {
            if (bad_index) *bad_index = x.first >= n_atoms ? x.first : x.second;
            return false;
        }
    }
    return true;
}

inline std::vector<Block> make(const uint64_t *s, const uint64_t *e, size_t n) {
    std::vector<Block> b(n);
    for (size_t k = 0; k < n; ++k) b[k] = Block(s[k], e[k]);
    return b;
}
inline size_t store(const std::vector<Block> &b, uint64_t *os, uint64_t *oe) {
    for (size_t k = 0; k < b.size(); ++k) { os[k] = b[k].first; oe[k] = b[k].second; }
    return b.size();
}

}  // namespace grc
