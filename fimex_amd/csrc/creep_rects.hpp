// Host-side geometry of the creep fill by rectangles (fill.hip, run_creepfill): pure C++, no device code -- tests/test_creep_rects.py
// compiles it on its own and checks its invariants on random masks.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace fimex_amd {
namespace creep_rects {

struct Rect {
    uint32_t xa, xb, ya, yb;  // inclusive, ring included
    bool operator==(const Rect& o) const { return xa == o.xa && xb == o.xb && ya == o.ya && yb == o.yb; }
};

// maximal runs of set flags -> intervals with one clean element on either side (or the array's end), at least minLen long where
// the array allows, merged where they overlap by more than a shared boundary
inline std::vector<std::pair<uint32_t, uint32_t>> dirty_intervals(const std::vector<unsigned char>& dirty, uint32_t minLen)
{
    const uint32_t n = (uint32_t)dirty.size();
    std::vector<std::pair<uint32_t, uint32_t>> out;
    for (uint32_t i = 0; i < n;) {
        if (!dirty[i]) { ++i; continue; }
        uint32_t j = i;
        while (j + 1 < n && dirty[j + 1]) ++j;
        uint32_t a = i > 0 ? i - 1 : 0, b = j + 1 < n ? j + 1 : n - 1;
        while (b - a + 1 < minLen && (b + 1 < n || a > 0)) {  // grow over clean elements (a later dirty one: merged below)
            if (b + 1 < n) ++b;
            else --a;
        }
        out.emplace_back(a, b);
        i = j + 1;
    }
    // merge: an interval that reaches into the next one's dirty elements
    std::vector<std::pair<uint32_t, uint32_t>> merged;
    for (const auto& iv : out) {
        if (!merged.empty() && iv.first < merged.back().second) merged.back().second = std::max(merged.back().second, iv.second);
        else merged.push_back(iv);
    }
    // the ends must be clean or the array's ends: growing may have stopped on a dirty element of a neighbour that was merged away
    for (auto& iv : merged) {
        while (iv.first > 0 && dirty[iv.first]) --iv.first;
        while (iv.second + 1 < n && dirty[iv.second]) ++iv.second;
    }
    std::vector<std::pair<uint32_t, uint32_t>> fin;
    for (const auto& iv : merged) {
        if (!fin.empty() && iv.first < fin.back().second) fin.back().second = std::max(fin.back().second, iv.second);
        else fin.push_back(iv);
    }
    return fin;
}

// Undefined cells inside rectangle r (whose ring is defined, or the field's border) -> the rectangles that hold them: runs of rows
// with undefined cells x runs of columns with undefined cells in those rows, each looked at again on its own (the rows a column
// run needs are often fewer than the row run it came from).
inline void refine_rect(const uint32_t* bits, uint32_t words, const Rect& r, int depth, std::vector<Rect>& out)
{
    const uint32_t w = r.xb - r.xa + 1, h = r.yb - r.ya + 1;
    const uint32_t k0 = r.xa >> 5, k1 = r.xb >> 5;
    auto word_mask = [&](uint32_t k) -> uint32_t {
        uint32_t m = 0xFFFFFFFFu;
        if (k == k0) m &= 0xFFFFFFFFu << (r.xa & 31);
        if (k == k1) m &= 0xFFFFFFFFu >> (31 - (r.xb & 31));
        return m;
    };
    std::vector<unsigned char> rowDirty(h, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint32_t any = 0;
        for (uint32_t k = k0; k <= k1; ++k) any |= bits[(size_t)(r.ya + y) * words + k] & word_mask(k);
        rowDirty[y] = any != 0;
    }
    for (const auto& rv : dirty_intervals(rowDirty, 4)) {
        std::vector<uint32_t> orw(k1 - k0 + 1, 0);
        for (uint32_t y = rv.first; y <= rv.second; ++y)
            for (uint32_t k = k0; k <= k1; ++k) orw[k - k0] |= bits[(size_t)(r.ya + y) * words + k] & word_mask(k);
        std::vector<unsigned char> colDirty(w, 0);
        for (uint32_t x = 0; x < w; ++x) { const uint32_t ax = r.xa + x; colDirty[x] = (orw[(ax >> 5) - k0] >> (ax & 31)) & 1u; }
        for (const auto& cv : dirty_intervals(colDirty, 4)) {
            const Rect q{r.xa + cv.first, r.xa + cv.second, r.ya + rv.first, r.ya + rv.second};
            if (q == r || depth == 0) out.push_back(q);
            else refine_rect(bits, words, q, depth - 1, out);
        }
    }
}

// the rectangles of one slice from its NaN bitmap; false: not worth it, take the whole slice
inline bool slice_rects(const uint32_t* bits, uint32_t nx, uint32_t ny, uint32_t words, std::vector<Rect>& rects)
{
    rects.clear();
    refine_rect(bits, words, Rect{0, nx - 1, 0, ny - 1}, 4, rects);
    size_t area = 0;
    for (const Rect& r : rects) area += (size_t)(r.xb - r.xa + 1) * (r.yb - r.ya + 1);
    return !rects.empty() && rects.size() <= 64 && area * 2 <= (size_t)nx * ny;
}

}  // namespace creep_rects
}  // namespace fimex_amd
