// Forward-mapping regrid (bucket reductions) for gfx950.
//
// Replaces CachedForwardInterpolation (src/CachedForwardInterpolation.cc:38-131): the reference
// scans the source slice and push_back()s every value into a std::vector per target cell, then
// aggregates each vector (sum / mean / median / max / min).  Here the scatter is inverted once,
// at plan creation, into a CSR list of source cells per target cell, kept in source scan order;
// the apply kernel is then a gather with no atomics and no allocation, and sums add in exactly the
// reference's order (bit-identical results).
//
// Two apply paths, chosen per plan from its largest bucket:
//  * lane-per-target: one lane walks its bucket; ZC slices are reduced together so that the CSR
//    entries are read once per ZC slices;
//  * wave-per-target (large buckets): the 64 lanes load 64 bucket entries at a time (coalesced
//    index reads), a ballot drops undefined values, and the reduction is finished across the wave
//    (shuffle reduction for max / min, ordered lane scan for sum / mean so that the reference's
//    left-to-right float additions are kept).
#include <cstring>  // before rocprim: its texture iterator calls memset

#include "plan.hpp"

#include <rocprim/rocprim.hpp>

#include <vector>

namespace fimex_amd {

namespace {

__device__ __forceinline__ float undefined_f() { return __uint_as_float(0x7fc00000u); }

// RoundAndClamp(0, n-1, -1), src/Utils.cc:42-58 (round(): half away from zero)
__device__ __forceinline__ int64_t round_clamp(double d, int64_t n)
{
    if (!(fabs(d) < 1073741824.0)) return -1;  // NaN / inf / beyond int: invalid
    const int64_t r = (int64_t)round(d);
    return (r >= 0 && r < n) ? r : -1;
}

// target cell of every source cell, src/CachedForwardInterpolation.cc:72-73 + :104-107
__global__ void __launch_bounds__(kBlock) forward_targets(const double* __restrict__ px, const double* __restrict__ py,
                                                          uint32_t nIn, int64_t outX, int64_t outY, uint32_t* __restrict__ tgt)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nIn) return;
    const int64_t tx = round_clamp(px[i], outX);
    const int64_t ty = round_clamp(py[i], outY);
    tgt[i] = (tx >= 0 && ty >= 0) ? (uint32_t)(ty * outX + tx) : kInvalidPos;
}

struct FwdArgs {
    const float* in;
    float* out;
    const uint32_t* offsets;
    const uint32_t* src;
    uint32_t nOut;
    size_t inLayer;
    uint32_t nz;
    uint32_t zPerBlock;
    uint32_t rankAll;  // tuning build (FWD_MEDIAN_SHORT=0): the median of every bucket by rank counting
};

template <bool UNDEF>
__device__ __forceinline__ bool keep(float v) { return UNDEF || !isnan(v); }

// One bucket, ZC slices at a time: sum / mean / max / min (KIND 0 / 1 / 3 / 4) and the median of buckets of at most two
// source cells (KIND 5: rank size()/2 of one value is the value, of two the larger one -- the second where they compare equal,
// which is what counting "less, or equal and earlier" picks, src/CachedForwardInterpolation.cc:49-53).
// i0: the bucket's first source cell, loaded once per lane outside the slice loop (most buckets hold one cell: no dependent
// index load per slice group is left).
// AHEAD: source cells whose values are requested before the first of them is added (1: none ahead -- plans whose buckets are short)
template <int KIND, bool UNDEF, int ZC, int AHEAD>
__device__ __forceinline__ void reduce_bucket(const FwdArgs& a, uint32_t b, uint32_t e, uint32_t i0, const float* src, const size_t (&koff)[ZC], float (&r)[ZC])
{
    float acc[ZC];
    uint32_t cnt[ZC];
    bool anyNan[ZC];
#pragma unroll
    for (int k = 0; k < ZC; ++k) { acc[k] = 0.f; cnt[k] = 0; anyNan[k] = false; }
    auto take = [&](const float (&v)[ZC]) __attribute__((always_inline)) {  // one source cell of the bucket, all ZC slices, in scan order
#pragma unroll
        for (int k = 0; k < ZC; ++k) {
            if (keep<UNDEF>(v[k])) {
                if (KIND == 0 || KIND == 1) acc[k] = acc[k] + v[k];                        // std::accumulate(.., 0.f)
                else if (KIND == 3) { if (cnt[k] == 0 || acc[k] < v[k]) acc[k] = v[k]; }   // std::max_element
                else if (KIND == 4) { if (cnt[k] == 0 || v[k] < acc[k]) acc[k] = v[k]; }   // std::min_element
                else {  // median of at most two: the larger one, the second where they compare equal -- a running form of it
                    if (UNDEF && isnan(v[k])) anyNan[k] = true;
                    if (cnt[k] == 0 || !(acc[k] > v[k])) acc[k] = v[k];
                }
                cnt[k]++;
            }
        }
    };
    uint32_t j = b;
    // long buckets (a source finer than the target): the values of AHEAD cells are requested before the first is added -- the
    // additions stay in scan order, but a lane no longer waits a memory round trip per cell (0.1-degree global -> 1-degree global,
    // 100 cells per bucket, 100 slices: 3.87 ms without, 1.74 ms with 4, 1.38 ms with 8 cells ahead; the registers this takes cost
    // the sparse plans of configs[3] 13 %, hence the variant without)
    constexpr int kAhead = AHEAD;
    if constexpr (AHEAD > 1)
    for (; j + kAhead <= e; j += kAhead) {
        uint32_t i[kAhead];
#pragma unroll
        for (int q = 0; q < kAhead; ++q) i[q] = (q == 0 && j == b) ? i0 : a.src[j + q];
        float v[kAhead][ZC];
#pragma unroll
        for (int q = 0; q < kAhead; ++q)
#pragma unroll
            for (int k = 0; k < ZC; ++k) v[q][k] = src[koff[k] + i[q]];
#pragma unroll
        for (int q = 0; q < kAhead; ++q) take(v[q]);
    }
    for (; j < e; ++j) {
        const uint32_t i = (j == b) ? i0 : a.src[j];
        float v[ZC];
#pragma unroll
        for (int k = 0; k < ZC; ++k) v[k] = src[koff[k] + i];
        take(v);
    }
#pragma unroll
    for (int k = 0; k < ZC; ++k) {
        r[k] = undefined_f();                             // empty bucket, :123-124
        if (cnt[k] != 0) {
            if (KIND == 1) r[k] = acc[k] / (float)cnt[k];  // aggrMean: sum / size()
            else if (KIND == 5) {
                // a NaN inside an "undef" bucket: see median_by_rank
                if (!(UNDEF && anyNan[k])) r[k] = acc[k];
            } else r[k] = acc[k];
        }
    }
}

// value of rank size()/2 among the kept values of one bucket of one slice (std::nth_element, src/CachedForwardInterpolation.cc:49-53)
template <bool UNDEF>
__device__ __forceinline__ float median_by_rank(const FwdArgs& a, uint32_t b, uint32_t e, const float* src)
{
    uint32_t n = 0;
    bool anyNan = false;
    for (uint32_t j = b; j < e; ++j) {
        const float v = src[a.src[j]];
        if (isnan(v)) anyNan = true;
        if (keep<UNDEF>(v)) n++;
    }
    float r = undefined_f();
    // a NaN inside an "undef" bucket: the reference's nth_element result is implementation-defined
    // (comparator not a strict weak order); the documented intent (value + undef = undef) is kept
    if (n != 0 && !(UNDEF && anyNan)) {
        const uint32_t want = n / 2;
        for (uint32_t j = b; j < e; ++j) {
            const float v = src[a.src[j]];
            if (!keep<UNDEF>(v)) continue;
            uint32_t less = 0, equalBefore = 0;
            for (uint32_t q = b; q < e; ++q) {
                const float w = src[a.src[q]];
                if (!keep<UNDEF>(w)) continue;
                less += (w < v);
                equalBefore += (w == v && q < j);
            }
            if (less + equalBefore == want) { r = v; break; }
        }
    }
    return r;
}

// one lane per target cell.  Workgroups are dealt round-robin over the 8 XCDs; XCD x takes the x-th eighth of the target
// cells (blocks of 256 consecutive cells), so that a source line -- whose cells map to neighbouring targets in several target
// rows -- is fetched into ONE L2 instead of into all eight (configs[3]: 1.85 GB -> about 1.1 GB of fabric traffic per 100 slices).
// RANK (median only): some bucket of the plan holds more than two cells; without it the rank-counting path is not compiled in
template <int KIND, bool UNDEF, int ZC, bool RANK = false, int AHEAD = 1>
__global__ void __launch_bounds__(kBlock) forward_apply_lane(FwdArgs a)
{
    const uint32_t perXcd = gridDim.x / kXcds;  // the grid holds 8 * perXcd workgroups per z chunk
    const uint32_t blk = (blockIdx.x % kXcds) * perXcd + blockIdx.x / kXcds;
    const uint32_t t = blk * kBlock + threadIdx.x;
    if (t >= a.nOut) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const uint32_t b = a.offsets[t], e = a.offsets[t + 1];
    const uint32_t i0 = (e > b) ? a.src[b] : 0u;
    if (KIND == 5 && RANK && (e - b > 2 || (kTuningBuild && a.rankAll != 0))) {
        // median of a bucket of three or more source cells: value of rank size()/2 (std::nth_element, :49-53) by rank counting, slice
        // by slice -- no scratch memory, any bucket size; the other lanes of the wave (buckets of at most two cells, by far the
        // most in practice: DESIGN.md gives the occupancy histogram) take the batched path below
        for (uint32_t z = z0; z < z1; ++z) __builtin_nontemporal_store(median_by_rank<UNDEF>(a, b, e, a.in + (size_t)z * a.inLayer), a.out + (size_t)z * a.nOut + t);
        return;
    }
    for (uint32_t z = z0; z < z1; z += ZC) {
        const float* src = a.in + (size_t)z * a.inLayer;
        // slices past the end of the chunk re-read the last one (results dropped): no branch in the gather loop
        size_t koff[ZC];
#pragma unroll
        for (int k = 0; k < ZC; ++k) koff[k] = (size_t)min((uint32_t)k, z1 - 1 - z) * a.inLayer;
        float r[ZC];
        reduce_bucket<KIND, UNDEF, ZC, AHEAD>(a, b, e, i0, src, koff, r);
#pragma unroll
        for (int k = 0; k < ZC; ++k)
            if (z + k < z1) __builtin_nontemporal_store(r[k], a.out + (size_t)(z + k) * a.nOut + t);
    }
}

// ---- wave-per-target path -------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(v, d, kWave); v = (v < o) ? o : v; }
    return v;
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(v, d, kWave); v = (o < v) ? o : v; }
    return v;
}

// one wave per target cell; 4 targets per workgroup
template <int KIND, bool UNDEF>
__global__ void __launch_bounds__(kBlock) forward_apply_wave(FwdArgs a)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t t = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (t >= a.nOut) return;  // wave-uniform
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const uint32_t b = a.offsets[t], e = a.offsets[t + 1];
    for (uint32_t z = z0; z < z1; ++z) {
        const float* src = a.in + (size_t)z * a.inLayer;
        float acc = 0.f;       // running result, identical in every lane
        uint32_t cnt = 0;
        bool firstNan = false;  // max/min with UNDEF: a NaN in first position sticks (std::max_element)
        for (uint32_t base = b; base < e; base += kWave) {
            const uint32_t j = base + lane;
            const bool inRange = j < e;
            const float v = inRange ? src[a.src[j]] : 0.f;
            const bool valid = inRange && keep<UNDEF>(v);
            const unsigned long long mask = __ballot(valid);
            if (mask == 0) continue;
            if (KIND == 0 || KIND == 1) {
                // ordered sum: fold the surviving lanes left to right, as std::accumulate does
                unsigned long long m = mask;
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    acc = acc + __shfl(v, l, kWave);
                    m &= m - 1;
                }
            } else {
                if (UNDEF && cnt == 0) {
                    const int l0 = __ffsll((long long)mask) - 1;
                    firstNan = isnan(__shfl(v, l0, kWave));
                }
                // NaNs after the first position never win a "<" comparison: neutralise them
                const bool usableV = valid && !isnan(v);
                // std::max_element / min_element return the FIRST of the elements that compare equal to the extremum: of +0.0 and
                // -0.0 the one that comes first in the bucket.  The shuffle reduction finds the value, the lowest lane that holds it
                // supplies the bits.
                if (KIND == 3) {
                    const float m = wave_max(usableV ? v : -INFINITY);
                    const unsigned long long at = __ballot(usableV && v == m);
                    if (at) {
                        const float first = __shfl(v, __ffsll((long long)at) - 1, kWave);
                        acc = (cnt == 0 || acc < first) ? first : acc;
                    }
                } else {
                    const float m = wave_min(usableV ? v : INFINITY);
                    const unsigned long long at = __ballot(usableV && v == m);
                    if (at) {
                        const float first = __shfl(v, __ffsll((long long)at) - 1, kWave);
                        acc = (cnt == 0 || first < acc) ? first : acc;
                    }
                }
            }
            cnt += (uint32_t)__popcll(mask);
        }
        if (lane == 0) {
            float r = undefined_f();
            if (cnt != 0) {
                if (KIND == 1) r = acc / (float)cnt;
                else if ((KIND == 3 || KIND == 4) && firstNan) r = undefined_f();
                else r = acc;
            }
            a.out[(size_t)z * a.nOut + t] = r;
        }
    }
}

constexpr int kLaneZc = 8;  // slices a lane reduces together (gathers in flight per lane)

// Median of buckets of up to 64 * Q source cells, one wave per target cell: the bucket's values of one slice sit in Q registers per
// lane (position q * 64 + lane of the bucket's scan order), every kept value is broadcast once (v_readlane) and every lane counts,
// for its own values, how many kept values are smaller or equal-and-earlier -- its rank in the order std::nth_element's result
// follows (src/CachedForwardInterpolation.cc:49-53).  The one value of rank size() / 2 is the median; its lane stores it.
// n * Q comparisons per lane instead of the n * n loads of median_by_rank: 0.1-degree global -> 1-degree global (100 cells per
// bucket), 100 slices: 130 ms -> see DESIGN.md 6.
template <bool UNDEF, int Q>
__global__ void __launch_bounds__(kBlock) forward_apply_median_wave(FwdArgs a)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t t = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (t >= a.nOut) return;  // wave-uniform
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const uint32_t b = a.offsets[t], n0 = a.offsets[t + 1] - b;
    if (n0 == 0) {
        if (lane == 0)
            for (uint32_t z = z0; z < z1; ++z) a.out[(size_t)z * a.nOut + t] = undefined_f();
        return;
    }
    uint32_t idx[Q];
    bool has[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const uint32_t p = (uint32_t)q * kWave + lane;
        has[q] = p < n0;
        idx[q] = has[q] ? a.src[b + p] : 0u;
    }
    for (uint32_t z = z0; z < z1; ++z) {
        const float* src = a.in + (size_t)z * a.inLayer;
        float v[Q];
        bool kept[Q];
        unsigned long long keptMask[Q];
        bool nanHere = false;
        uint32_t n = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            v[q] = has[q] ? src[idx[q]] : 0.f;
            kept[q] = has[q] && keep<UNDEF>(v[q]);
            nanHere = nanHere || (has[q] && isnan(v[q]));
            keptMask[q] = __ballot(kept[q]);
            n += (uint32_t)__popcll(keptMask[q]);
        }
        const bool anyNan = __ballot(nanHere) != 0;
        float* o = a.out + (size_t)z * a.nOut + t;
        // a NaN inside an "undef" bucket: the reference's nth_element result is implementation-defined; value + undef = undef is kept
        if (n == 0 || (UNDEF && anyNan)) {
            if (lane == 0) *o = undefined_f();
            continue;
        }
        uint32_t rank[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) rank[q] = 0;
#pragma unroll
        for (int q2 = 0; q2 < Q; ++q2) {
            unsigned long long m = keptMask[q2];  // wave-uniform: the loop below runs on the scalar unit
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float u = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[q2]), l));
                const uint32_t pu = (uint32_t)q2 * kWave + (uint32_t)l;
#pragma unroll
                for (int q = 0; q < Q; ++q) rank[q] += (u < v[q] || (u == v[q] && pu < (uint32_t)q * kWave + lane)) ? 1u : 0u;
            }
        }
        const uint32_t want = n / 2;
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (kept[q] && rank[q] == want) *o = v[q];  // exactly one value of the bucket has this rank
    }
}

// The same median by SELECTION instead of ranking: the kept values of a bucket, as keys whose unsigned order is the floats' order,
// sit in Q registers per lane; the key of rank size() / 2 is built bit by bit from the top -- "how many keys lie below the candidate?"
// is one comparison per register, whose result IS the ballot, a population count and a scalar decision: 32 rounds of Q vector and a
// handful of scalar instructions, whatever the bucket's length (ranking: one broadcast and 2 Q comparisons per VALUE; 100 cells per
// bucket, 100 slices: 13.1 ms -> see DESIGN.md 6).  The value comes back out of the key.  Only +0.0 and -0.0 compare equal without
// being the same bits: they share a key, and where the median is a zero, the order the reference's result follows ("less, or equal
// and earlier", src/CachedForwardInterpolation.cc:49-53) picks among the bucket's zeros by position.
template <bool UNDEF, int Q>
__global__ void __launch_bounds__(kBlock) forward_apply_median_select(FwdArgs a)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t t = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (t >= a.nOut) return;  // wave-uniform
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const uint32_t b = a.offsets[t], n0 = a.offsets[t + 1] - b;
    if (n0 == 0) {
        if (lane == 0)
            for (uint32_t z = z0; z < z1; ++z) a.out[(size_t)z * a.nOut + t] = undefined_f();
        return;
    }
    uint32_t idx[Q];
    bool has[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const uint32_t p = (uint32_t)q * kWave + lane;
        has[q] = p < n0;
        idx[q] = has[q] ? a.src[b + p] : 0u;
    }
    constexpr uint32_t kZeroKey = 0x80000000u;
    for (uint32_t z = z0; z < z1; ++z) {
        const float* src = a.in + (size_t)z * a.inLayer;
        float v[Q];
        uint32_t key[Q];
        bool nanHere = false;
        uint32_t n = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            v[q] = has[q] ? src[idx[q]] : 0.f;
            const bool kept = has[q] && keep<UNDEF>(v[q]);
            nanHere = nanHere || (has[q] && isnan(v[q]));
            n += (uint32_t)__popcll(__ballot(kept));
            const uint32_t bits = __float_as_uint(v[q]);
            const uint32_t ordered = bits ^ (((int32_t)bits < 0) ? 0xFFFFFFFFu : 0x80000000u);  // unsigned order = float order
            key[q] = !kept ? 0xFFFFFFFFu : (v[q] == 0.f ? kZeroKey : ordered);                    // dropped: above every kept value
        }
        const bool anyNan = __ballot(nanHere) != 0;
        float* o = a.out + (size_t)z * a.nOut + t;
        // a NaN inside an "undef" bucket: the reference's nth_element result is implementation-defined; value + undef = undef is kept
        if (n == 0 || (UNDEF && anyNan)) {
            if (lane == 0) *o = undefined_f();
            continue;
        }
        const uint32_t want = n / 2;
        uint32_t K = 0;  // the largest candidate with at most `want` keys below it = the key of rank `want`
#pragma unroll
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t T = K | (1u << bit);
            uint32_t below = 0;
#pragma unroll
            for (int q = 0; q < Q; ++q) below += (uint32_t)__popcll(__ballot(key[q] < T));
            K = (below <= want) ? T : K;
        }
        uint32_t bits = (K & 0x80000000u) ? (K ^ 0x80000000u) : ~K;
        if (K == kZeroKey) {
            // the median is a zero: the (want - #negative values)-th of the bucket's zeros in scan order supplies the sign
            uint32_t k = want;
#pragma unroll
            for (int q = 0; q < Q; ++q) k -= (uint32_t)__popcll(__ballot(key[q] < kZeroKey));
            bool found = false;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                unsigned long long m = __ballot(key[q] == kZeroKey);
                const uint32_t c = (uint32_t)__popcll(m);
                if (!found && k < c) {
                    for (uint32_t i = 0; i < k; ++i) m &= m - 1;
                    bits = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v[q]), __ffsll((long long)m) - 1);
                    found = true;
                } else if (!found) {
                    k -= c;
                }
            }
        }
        if (lane == 0) *o = __uint_as_float(bits);
    }
}

template <int KIND, bool UNDEF>
void launch_kind(const FwdArgs& a, dim3 grid, bool wavePath, bool longBuckets, hipStream_t stream)
{
    if (wavePath) {
        dim3 g((uint32_t)ceil_div(a.nOut, kBlock / kWave), grid.y, 1);
        forward_apply_wave<KIND, UNDEF><<<g, kBlock, 0, stream>>>(a);
    } else if (longBuckets) {
        forward_apply_lane<KIND, UNDEF, kLaneZc, false, 8><<<grid, kBlock, 0, stream>>>(a);
    } else if (a.nz == 1) {
        // one slice per call is the reference's own call pattern (src/CDMInterpolator.cc:251-259): no eight-slice passes that read
        // the one slice eight times (configs[3], one slice: see DESIGN.md 6)
        forward_apply_lane<KIND, UNDEF, 1><<<grid, kBlock, 0, stream>>>(a);
    } else if (a.nz <= 2) {
        forward_apply_lane<KIND, UNDEF, 2><<<grid, kBlock, 0, stream>>>(a);
    } else if (a.nz <= 4) {
        forward_apply_lane<KIND, UNDEF, 4><<<grid, kBlock, 0, stream>>>(a);
    } else {
        forward_apply_lane<KIND, UNDEF, kLaneZc><<<grid, kBlock, 0, stream>>>(a);
    }
}

__global__ void __launch_bounds__(kBlock) iota_kernel(uint32_t* __restrict__ idx, uint32_t n)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) idx[i] = i;
}

// offsets[t] = first position in the sorted keys that is >= t (t = 0 .. nOut): lower bound by binary search
__global__ void __launch_bounds__(kBlock) csr_offsets_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t nOut, uint32_t* __restrict__ offsets)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t > nOut) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (keys[mid] < t) lo = mid + 1;
        else hi = mid;
    }
    offsets[t] = lo;
}

__global__ void __launch_bounds__(kBlock) csr_stats_kernel(const uint32_t* __restrict__ offsets, uint32_t nOut, unsigned long long* __restrict__ stats)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    uint32_t len = 0;
    bool empty = false;
    if (t < nOut) {
        len = offsets[t + 1] - offsets[t];
        empty = len == 0;
        if (t == nOut - 1) stats[0] = offsets[nOut];  // mapped source cells
    }
    const unsigned long long e = __popcll(__ballot(empty));
    uint32_t m = len;
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (e) atomicAdd(&stats[2], e);
        if (m) atomicMax(&stats[1], (unsigned long long)m);
    }
}

}  // namespace

void build_forward_plan(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream)
{
    const size_t nIn = plan.inX * plan.inY, nOut = plan.outX * plan.outY;
    FA_REQUIRE(nIn > 0 && nIn <= kMaxSliceCells, "input grid must have between 1 and 2^30-1 cells per slice");
    FA_REQUIRE(nOut > 0 && nOut <= 0x7FFFFFFFu, "output grid must have between 1 and 2^31-1 cells");
    DeviceArray<uint32_t> d_tgt(nIn);
    forward_targets<<<dim3((uint32_t)ceil_div(nIn, kBlock)), kBlock, 0, stream>>>(
        d_px, d_py, (uint32_t)nIn, (int64_t)plan.outX, (int64_t)plan.outY, d_tgt.get());
    FA_HIP(hipGetLastError());
    // CSR by a stable radix sort of (target, source index): buckets keep the source scan order, which is the
    // reference's push_back order (src/CachedForwardInterpolation.cc:103-112); unmapped cells (key ~0) sort to the end
    DeviceArray<uint32_t> d_idx(nIn), d_keysSorted(nIn), d_idxSorted(nIn);
    iota_kernel<<<dim3((uint32_t)ceil_div(nIn, kBlock)), kBlock, 0, stream>>>(d_idx.get(), (uint32_t)nIn);
    FA_HIP(hipGetLastError());
    size_t tmpBytes = 0;
    FA_HIP(rocprim::radix_sort_pairs(nullptr, tmpBytes, d_tgt.get(), d_keysSorted.get(), d_idx.get(), d_idxSorted.get(), nIn, 0, 32, stream));
    DeviceArray<unsigned char> tmp(tmpBytes ? tmpBytes : 1);
    FA_HIP(rocprim::radix_sort_pairs(tmp.get(), tmpBytes, d_tgt.get(), d_keysSorted.get(), d_idx.get(), d_idxSorted.get(), nIn, 0, 32, stream));
    plan.offsets.allocate(nOut + 1);
    DeviceArray<unsigned long long> d_stats(3);  // mapped, maxBucket, empty
    FA_HIP(hipMemsetAsync(d_stats.get(), 0, 3 * sizeof(unsigned long long), stream));
    csr_offsets_kernel<<<dim3((uint32_t)ceil_div(nOut + 1, kBlock)), kBlock, 0, stream>>>(d_keysSorted.get(), (uint32_t)nIn, (uint32_t)nOut,
                                                                                          plan.offsets.get());
    FA_HIP(hipGetLastError());
    csr_stats_kernel<<<dim3((uint32_t)ceil_div(nOut, kBlock)), kBlock, 0, stream>>>(plan.offsets.get(), (uint32_t)nOut, d_stats.get());
    FA_HIP(hipGetLastError());
    unsigned long long h_stats[3] = {0, 0, 0};
    FA_HIP(hipMemcpyAsync(h_stats, d_stats.get(), sizeof(h_stats), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    const size_t mapped = (size_t)h_stats[0], maxBucket = (size_t)h_stats[1], empty = (size_t)h_stats[2];
    plan.src.allocate(mapped ? mapped : 1);
    if (mapped) FA_HIP(hipMemcpyAsync(plan.src.get(), d_idxSorted.get(), mapped * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
    FA_HIP(hipStreamSynchronize(stream));  // temporaries are released on return
    plan.info.planBytes = (nOut + 1) * sizeof(uint32_t) + mapped * sizeof(uint32_t);
    plan.info.undefinedCells = empty;
    plan.info.maxBucket = maxBucket;
    plan.info.mappedSourceCells = mapped;
    build_forward_tiles(plan, stream);  // dense mappings: the LDS-staged form (forward_tiled.hip)
}

void launch_forward_apply(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream)
{
    if (nz == 0) return;
    FA_REQUIRE(nz <= 0xFFFFFFFFu, "too many slices");
    if (launch_forward_tiled(plan, d_in, nz, d_out, stream)) return;  // dense mappings, all aggregates but the median
    FwdArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.offsets = plan.offsets.get();
    a.src = plan.src.get();
    a.nOut = (uint32_t)(plan.outX * plan.outY);
    a.inLayer = plan.inX * plan.inY;
    a.nz = (uint32_t)nz;
    a.rankAll = tuning("FWD_MEDIAN_SHORT", 1) == 0 ? 1u : 0u;
    // z chunks: a workgroup reads its CSR entries once per chunk, so chunks are long (the CSR of configs[3] is 10 MB, a slice's
    // must-move bytes about the same); several chunks only where the target grid alone would not fill the chip
    const size_t blocks = ceil_div(a.nOut, kBlock);
    uint32_t zpb = (uint32_t)tuning("FWD_ZPB", blocks >= 4096 ? 64 : 16);
    if (zpb > nz) zpb = (uint32_t)nz;
    size_t chunks = ceil_div(nz, (size_t)zpb);
    zpb = (uint32_t)ceil_div(nz, chunks);  // chunks of one size
    a.zPerBlock = zpb;
    chunks = ceil_div(nz, (size_t)zpb);
    FA_REQUIRE(chunks <= 65535, "too many z chunks for one launch");
    const dim3 grid((uint32_t)ceil_div(a.nOut, kBlock), (uint32_t)chunks, 1);
    // the lane kernels map workgroup -> block of targets through the XCD it runs on: their grid is a multiple of 8
    const dim3 gridLane((uint32_t)(ceil_div(blocks, (size_t)kXcds) * kXcds), (uint32_t)chunks, 1);
    // mean bucket length decides: long buckets are reduced by a whole wave
    const size_t nonEmpty = a.nOut - plan.info.undefinedCells;
    const double meanBucket = nonEmpty ? (double)plan.info.mappedSourceCells / (double)nonEmpty : 0.0;
    int waveMode = tuning("FWD_WAVE", -1);
    // wave per bucket only where the targets are too few to fill the chip with lanes (fewer than 64 K (target, z chunk) lanes): a
    // dense mapping with many targets runs 1.7-1.8 ms through the lane kernels against 6.5 (mean: the ordered fold) and 2.0 ms (max)
    // through the wave kernels (0.1-degree global -> 1-degree global, 100 cells per bucket, 100 slices)
    const bool wavePath = (waveMode < 0) ? (meanBucket >= 32.0 && (size_t)a.nOut * chunks < 65536) : (waveMode != 0);
    const bool longBuckets = meanBucket > 4.0;  // the lane kernels with eight cells of look-ahead
    const bool u = plan.undefAggr;
    switch (plan.aggregate) {
    case Aggregate::Sum: u ? launch_kind<0, true>(a, gridLane, wavePath, longBuckets, stream) : launch_kind<0, false>(a, gridLane, wavePath, longBuckets, stream); break;
    case Aggregate::Mean: u ? launch_kind<1, true>(a, gridLane, wavePath, longBuckets, stream) : launch_kind<1, false>(a, gridLane, wavePath, longBuckets, stream); break;
    case Aggregate::Max: u ? launch_kind<3, true>(a, gridLane, wavePath, longBuckets, stream) : launch_kind<3, false>(a, gridLane, wavePath, longBuckets, stream); break;
    case Aggregate::Min: u ? launch_kind<4, true>(a, gridLane, wavePath, longBuckets, stream) : launch_kind<4, false>(a, gridLane, wavePath, longBuckets, stream); break;
    case Aggregate::Median:
        // buckets of at most two cells: eight slices in flight, no rank counting; longer ones are ranked slice by slice by the same kernel
        // (four targets per lane with 16-byte stores were measured as well: 8 % slower on configs[3], the gathers lose parallelism)
        if (plan.info.maxBucket <= 1 && a.rankAll == 0) {
            // no bucket holds more than one cell (a source grid coarser than the target, configs[3]): the median of one value is
            // the value, which is what the max kernel returns for it bit for bit -- without the median's per-slice state
            u ? launch_kind<3, true>(a, gridLane, false, false, stream) : launch_kind<3, false>(a, gridLane, false, false, stream);
        } else if (a.rankAll == 0 && plan.info.maxBucket > 2 && plan.info.maxBucket <= 256 && meanBucket >= (double)tuning("FWD_MEDIAN_WAVE_MIN", 12) && tuning("FWD_MEDIAN_WAVE", 1) != 0) {
            // medium buckets throughout (a source finer than the target): one wave per target, the median by selection on keys.  Its
            // cost does not depend on the bucket's length (25 ms per 100 slices of a million targets), the lane kernel's rank counting
            // grows with its square (6.3 ms at 4-9 cells, about 32 at 25): a wave per target from a mean length of twelve
            const dim3 g((uint32_t)ceil_div(a.nOut, kBlock / kWave), gridLane.y, 1);
            const bool select = tuning("FWD_MEDIAN_SELECT", 1) != 0;  // 0: ranks by broadcast (the form before it)
            if (plan.info.maxBucket <= 64 && select) {
                if (u) forward_apply_median_select<true, 1><<<g, kBlock, 0, stream>>>(a);
                else forward_apply_median_select<false, 1><<<g, kBlock, 0, stream>>>(a);
            } else if (plan.info.maxBucket <= 128) {
                if (select) { if (u) forward_apply_median_select<true, 2><<<g, kBlock, 0, stream>>>(a); else forward_apply_median_select<false, 2><<<g, kBlock, 0, stream>>>(a); }
                else if (u) forward_apply_median_wave<true, 2><<<g, kBlock, 0, stream>>>(a);
                else forward_apply_median_wave<false, 2><<<g, kBlock, 0, stream>>>(a);
            } else {
                if (select) { if (u) forward_apply_median_select<true, 4><<<g, kBlock, 0, stream>>>(a); else forward_apply_median_select<false, 4><<<g, kBlock, 0, stream>>>(a); }
                else if (u) forward_apply_median_wave<true, 4><<<g, kBlock, 0, stream>>>(a);
                else forward_apply_median_wave<false, 4><<<g, kBlock, 0, stream>>>(a);
            }
        } else if (plan.info.maxBucket > 2 || a.rankAll != 0) {
            if (u) forward_apply_lane<5, true, kLaneZc, true><<<gridLane, kBlock, 0, stream>>>(a);
            else forward_apply_lane<5, false, kLaneZc, true><<<gridLane, kBlock, 0, stream>>>(a);
        } else if (tuning("FWD_MEDIAN_ZC", 8) == 4) {
            if (u) forward_apply_lane<5, true, 4><<<gridLane, kBlock, 0, stream>>>(a);
            else forward_apply_lane<5, false, 4><<<gridLane, kBlock, 0, stream>>>(a);
        } else {
            if (u) forward_apply_lane<5, true, kLaneZc><<<gridLane, kBlock, 0, stream>>>(a);
            else forward_apply_lane<5, false, kLaneZc><<<gridLane, kBlock, 0, stream>>>(a);
        }
        break;
    }
    FA_HIP(hipGetLastError());
}

}  // namespace fimex_amd
