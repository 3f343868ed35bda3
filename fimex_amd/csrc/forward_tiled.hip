// Forward-mapping regrid of DENSE mappings (a source finer than the target: many source cells per bucket), LDS-staged.
//
// The lane kernels of forward.hip gather a bucket's values straight from memory: neighbouring lanes (neighbouring targets) read
// cells one bucket width apart, every 128-byte line of the source is asked for by as many successive loads as a bucket is wide
// and the lines do not survive in L2 between them (0.1-degree global -> 1-degree global, 100 cells per bucket, 100 slices: 8.8 GB
// of fabric traffic for 2.6 GB of source, profiles/r03_forwarddense_mean_pmc.json).  Here the targets are cut into tiles of 64
// (one wave each); at plan creation every tile gets
//   * the list of 16-byte chunks of a source slice that hold its buckets' cells (per source row one run from the leftmost to the
//     rightmost cell any of its buckets holds in that row),
//   * a step table: for step s and lane l the LDS position of the s-th cell of lane l's bucket (source scan order, the reference's
//     push_back order, src/CachedForwardInterpolation.cc:103-112).
// The apply kernel streams the chunks of one slice into LDS with buffer_load_dwordx4 ... lds (whole lines, every byte once), the
// next slice's while the lanes walk their buckets through the step table (itself in LDS for the whole z chunk) and reduce in the
// reference's order: the same additions / comparisons as forward.hip's reduce_bucket, so results stay bit-identical.
// Tiles whose buckets are spread too far for LDS (a bucket across the date line, a pole) read from memory as the lane kernels do.
#include "staged_common.hpp"
#include "typed_convert.hpp"

#include <vector>

namespace fimex_amd {

namespace {

constexpr int kFtMaxRows = 512;           // source rows one tile's buckets may span
constexpr uint32_t kFtKmax = 32;          // 16-byte chunks a lane stages per slice
constexpr uint32_t kFtMaxChunks = kFtKmax * kWave;  // 32 KiB of LDS per slice
constexpr uint32_t kFtMaxLen = 256;       // cells per bucket (step table: 128 bytes per step)
constexpr uint32_t kFtDirect = 0xFFFFFFFFu;
constexpr uint32_t kFtMaxZChunks = 96;

struct FtGeom {
    uint32_t inX, outX, outY, tw, th, tilesX, nTiles;
    uint32_t cpc;        // cells per 16-byte chunk: 4 (float), 8 or 16 (slices in a 2- or 1-byte stored type)
    uint32_t cellBytes;  // 16 / cpc
};

using FtTile = ForwardTile;
constexpr uint32_t kFtGroup = 8;  // steps whose positions a lane reads with one 16-byte LDS load

// entry of (step s, lane l) in a tile's step table: groups of eight steps, lane-major inside a group
__host__ __device__ inline size_t step_entry(uint32_t s, uint32_t lane) { return ((size_t)(s / kFtGroup) * kWave + lane) * kFtGroup + s % kFtGroup; }

__device__ __forceinline__ uint32_t wave_max_u(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, kWave));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, kWave));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, kWave));
    return v;
}

// One wave per tile.  EMIT false: counts (chunks, longest bucket) into tiles[]; true: the chunk list and the step table at the
// bases the host has put into tiles[] in between.
template <bool EMIT>
__global__ void __launch_bounds__(kWave) forward_tile_scan(FtGeom g, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ src,
                                                           FtTile* __restrict__ tiles, uint32_t* __restrict__ chunkOff, unsigned short* __restrict__ steps, uint32_t padBytes)
{
    __shared__ int rowMin[kFtMaxRows], rowMax[kFtMaxRows];  // absolute source cell index of the leftmost / rightmost cell of a row
    __shared__ uint32_t rowChunk[kFtMaxRows + 1];
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    const uint32_t tx = (tile % g.tilesX) * g.tw + lane % g.tw, ty = (tile / g.tilesX) * g.th + lane / g.tw;
    const bool mine = tx < g.outX && ty < g.outY;
    uint32_t b = 0, e = 0;
    if (mine) { const uint32_t t = ty * g.outX + tx; b = offsets[t]; e = offsets[t + 1]; }
    const uint32_t maxLen = wave_max_u(e - b);
    FtTile T = EMIT ? tiles[tile] : FtTile{0, 0, 0, maxLen};
    if (maxLen == 0) {
        if (!EMIT && lane == 0) tiles[tile] = FtTile{0, 0, 0, 0};
        return;
    }
    if (EMIT && T.nChunks == kFtDirect) return;
    int rmin = 0x7FFFFFFF, rmax = -1;
    for (uint32_t j = b; j < e; ++j) {
        const int row = (int)(src[j] / g.inX);
        rmin = min(rmin, row);
        rmax = max(rmax, row);
    }
    rmin = wave_min_i(rmin);
    rmax = wave_max_i(rmax);
    const int nr = rmax - rmin + 1;
    if (nr > kFtMaxRows || maxLen > kFtMaxLen) {  // wave-uniform
        if (!EMIT && lane == 0) tiles[tile] = FtTile{0, kFtDirect, 0, maxLen};
        return;
    }
    for (int i = lane; i < nr; i += kWave) { rowMin[i] = 0x7FFFFFFF; rowMax[i] = -1; }
    __syncthreads();
    for (uint32_t j = b; j < e; ++j) {
        const int c = (int)src[j];
        const int i = c / (int)g.inX - rmin;
        atomicMin(&rowMin[i], c);
        atomicMax(&rowMax[i], c);
    }
    __syncthreads();
    const int cmask = (int)g.cpc - 1;
    if (lane == 0) {
        uint32_t acc = 0;
        for (int i = 0; i < nr; ++i) {
            rowChunk[i] = acc;
            if (rowMax[i] >= 0) acc += (uint32_t)((rowMax[i] - (rowMin[i] & ~cmask)) / (int)g.cpc + 1);  // chunks start on multiples of cpc cells of the slice
        }
        rowChunk[nr] = acc;
    }
    __syncthreads();
    const uint32_t nChunks = rowChunk[nr];
    if (!EMIT) {
        if (lane == 0) tiles[tile] = FtTile{0, nChunks > kFtMaxChunks ? kFtDirect : nChunks, 0, maxLen};
        return;
    }
    for (int i = 0; i < nr; ++i) {  // the chunk list: source cell index of every chunk's first cell
        if (rowMax[i] < 0) continue;
        const uint32_t first = (uint32_t)(rowMin[i] & ~cmask), n = rowChunk[i + 1] - rowChunk[i];
        for (uint32_t k = lane; k < n; k += kWave) chunkOff[T.chunkBase + rowChunk[i] + k] = first + g.cpc * k;
    }
    const uint32_t padded = (T.maxLen + kFtGroup - 1) / kFtGroup * kFtGroup;
    for (uint32_t s = 0; s < padded; ++s) {  // the step table: LDS position (in bytes) of step s of every lane's bucket; past its end: the padding cell
        uint32_t v = padBytes;
        if (b + s < e) {
            const int c = (int)src[b + s];
            const int i = c / (int)g.inX - rmin;
            v = (rowChunk[i] * g.cpc + (uint32_t)(c - (rowMin[i] & ~cmask))) * g.cellBytes;
        }
        steps[(size_t)T.stepBase + step_entry(s, lane)] = (unsigned short)v;
    }
}

struct FtArgs {
    const void* in;   // elements of T: float, or the variable's 1- or 2-byte stored type
    void* out;
    float bad;        // stored types: the fill value narrowed to float (mifi_bad2nanf's parameter), whether there is one,
    int hasBad;       // and the fill value as interpolationArray2Data receives it
    double fillOut;
    const uint32_t* offsets;
    const uint32_t* src;
    const FtTile* tiles;
    const uint32_t* chunkOff;
    const unsigned short* steps;
    FtGeom g;
    uint32_t nOut, inBytes, nz, slotChunks, slots;
    uint32_t zStart[kFtMaxZChunks + 1];  // z chunk c = slices zStart[c] .. zStart[c + 1]
    uint32_t loadAux;  // tuning build (FWD_TILED_AUX): cache policy of the staging loads
    uint32_t ablate;  // tuning build (FWD_TILED_ABLATE): 1 = no staging, 2 = no walk
    size_t inLayer;
};

// One value of a bucket in scan order.  The arithmetic of forward.hip's reduce_bucket (sum / mean / max / min = KIND 0 / 1 / 3 / 4) with
// nothing to decide per step beyond it: steps past the end of a lane's bucket read the slot's padding cell, whose value leaves the
// result alone (NaN where NaNs are skipped or never win a comparison; -0.0 for the sums that keep NaNs: x + -0.0 == x for every x,
// signed zeros included), and the kinds that keep undefined values know their count (the bucket's length) without counting.
//   max / min start from -inf / +inf instead of "the first kept value": a comparison with the first value then takes it, or leaves
//   an equal infinity -- the same value; a NaN in first place (kept by the "undef" kinds: std::max_element returns it) is looked at
//   separately (firstNan).
template <int KIND, bool UNDEF>
__device__ __forceinline__ void take(float v, float& acc, uint32_t& cnt)
{
    if (KIND == 0 || KIND == 1) {
        if (UNDEF) acc = acc + v;                           // std::accumulate(.., 0.f), NaNs included
        else { const bool ok = v == v; acc = ok ? acc + v : acc; cnt += ok ? 1u : 0u; }
    } else {
        if (KIND == 3) acc = (acc < v) ? v : acc;            // std::max_element: the first of equal elements stays
        else acc = (v < acc) ? v : acc;                      // std::min_element
        if (!UNDEF) cnt += (v == v) ? 1u : 0u;
    }
}
template <int KIND, bool UNDEF>
__device__ __forceinline__ float start_value() { return (KIND == 3) ? -INFINITY : (KIND == 4) ? INFINITY : 0.f; }
template <int KIND, bool UNDEF>
__device__ __forceinline__ float padding_value() { return ((KIND == 0 || KIND == 1) && UNDEF) ? -0.f : undefined_f(); }
template <int KIND, bool UNDEF>
__device__ __forceinline__ float finish(float acc, uint32_t cnt, uint32_t len, bool firstNan)
{
    const uint32_t n = UNDEF ? len : cnt;
    if (n == 0) return undefined_f();                        // empty bucket, src/CachedForwardInterpolation.cc:123-124
    if ((KIND == 3 || KIND == 4) && UNDEF && firstNan) return undefined_f();
    return KIND == 1 ? acc / (float)n : acc;                 // aggrMean: sum / size()
}
// the lane kernels' step (forward.hip reduce_bucket), for tiles that are not staged
template <int KIND, bool UNDEF>
__device__ __forceinline__ void take_plain(float v, float& acc, uint32_t& cnt)
{
    if (UNDEF || !isnan(v)) {
        if (KIND == 0 || KIND == 1) acc = acc + v;
        else if (KIND == 3) { if (cnt == 0 || acc < v) acc = v; }
        else { if (cnt == 0 || v < acc) acc = v; }
        cnt++;
    }
}

// element of a slice as float: Data::asFloat + mifi_bad2nanf for the stored types (typed_convert.hpp)
template <typename T>
__device__ __forceinline__ float cell_value(T raw, float bad, bool hasBad)
{
    if constexpr (std::is_same<T, float>::value) return raw;
    else return as_float_nan(raw, bad, hasBad);
}
// a result in the slices' element type: interpolationArray2Data (NaN -> fill value, rounded) for the stored types
template <typename T>
__device__ __forceinline__ void store_result(T* p, float r, T fill)
{
    if constexpr (std::is_same<T, float>::value) __builtin_nontemporal_store(r, p);
    else *p = from_float_fill<T>(r, fill);
}

// Slices z0 .. z1 of one tile, by one wave.  E: the slices' element type.  Stored types have no element that stands for "leaves
// the result alone" (see take), so their steps past a bucket's end are replaced by that value after the conversion.
template <int KIND, bool UNDEF, int G, int KMAX, typename E>
__device__ __forceinline__ void tile_slices(const FtArgs& a, uint32_t tile, uint32_t z0, uint32_t z1)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];  // the workgroup's dynamic LDS
    constexpr bool kTyped = !std::is_same<E, float>::value;
    const char* const inBase = reinterpret_cast<const char*>(a.in);
    E* const outBase = reinterpret_cast<E*>(a.out);
    const size_t sliceBytes = a.inLayer * sizeof(E);
    const bool hasBad = a.hasBad != 0;
    const E fill = kTyped ? static_cast<E>(a.fillOut) : E();  // ScaleValue's newFill_ (include/fimex/Utils.h:456)
    const uint32_t lane = threadIdx.x;
    const FtTile T = a.tiles[tile];
    const uint32_t tx = (tile % a.g.tilesX) * a.g.tw + lane % a.g.tw, ty = (tile / a.g.tilesX) * a.g.th + lane / a.g.tw;
    const bool mine = tx < a.g.outX && ty < a.g.outY;
    const uint32_t t = mine ? ty * a.g.outX + tx : 0u;
    uint32_t b = 0, e = 0;
    if (mine) { b = a.offsets[t]; e = a.offsets[t + 1]; }
    const uint32_t len = e - b;
    if (T.maxLen == 0) {
        if (mine) for (uint32_t z = z0; z < z1; ++z) store_result<E>(outBase + (size_t)z * a.nOut + t, undefined_f(), fill);
        return;
    }
    if (T.nChunks == kFtDirect) {  // buckets spread too far for LDS: from memory, slice by slice
        if (mine) for (uint32_t z = z0; z < z1; ++z) {
            const E* s = reinterpret_cast<const E*>(inBase + (size_t)z * sliceBytes);
            float acc = 0.f;
            uint32_t cnt = 0;
            for (uint32_t j = b; j < e; ++j) take_plain<KIND, UNDEF>(cell_value<E>(s[a.src[j]], a.bad, hasBad), acc, cnt);
            store_result<E>(outBase + (size_t)z * a.nOut + t, cnt == 0 ? undefined_f() : (KIND == 1 ? acc / (float)cnt : acc), fill);
        }
        return;
    }
    // The step table of the tile -- 2 bytes per step and lane, the eight steps of a group in 16 bytes per lane -- is this lane's own
    // and the same for every slice: it lives in REGISTERS for the whole z chunk (4 * G of them: G = 8 / 16 / 32 groups by the plan's
    // longest bucket; the waves a CU holds are bounded by LDS, not by registers).  In LDS it took a third of what a wave needs
    // there (13 KB next to 27 KB of slice at 100 cells per bucket: four waves per CU instead of six) and a 16-byte read per group.
    const uint32_t groups = (T.maxLen + kFtGroup - 1) / kFtGroup;
    uint32_t tr[4 * G];
    {
        const uint4* g128 = reinterpret_cast<const uint4*>(a.steps + T.stepBase);
        const uint32_t padPair = a.slotChunks * 16u * 0x10001u;  // groups past the tile's last: the padding cell
#pragma unroll
        for (int k = 0; k < G; ++k) {
            uint4 q = make_uint4(padPair, padPair, padPair, padPair);
            if ((uint32_t)k < groups) q = g128[k * kWave + lane];  // wave-uniform
            tr[4 * k] = q.x; tr[4 * k + 1] = q.y; tr[4 * k + 2] = q.z; tr[4 * k + 3] = q.w;
        }
    }
    const uint32_t slotFloats = (a.slotChunks + 1u) * 4u;  // the chunks of a slice and the padding cell behind them
    // staging list: chunk c = lane + 64 * j of the tile (byte offset of its 16 bytes inside a source slice); lanes beyond the list
    // repeat its last chunk into the unused tail of the slot (no selection here: the compiler turns one into a branch and a
    // wait around every one of these loads)
    uint32_t gOff[KMAX];
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)KMAX; ++j) {
        const uint32_t c = lane + j * kWave;
        gOff[j] = a.chunkOff[T.chunkBase + min(c, T.nChunks - 1)] * (uint32_t)sizeof(E);
    }
    const uint32_t un = (T.nChunks + kWave - 1) / kWave;
    auto dma = [&](uint32_t slot, uint32_t z) __attribute__((always_inline)) {
        const rsrc_t rs = make_rsrc(inBase + (size_t)z * sliceBytes, a.inBytes);
        float* const dst = &smem[slot * slotFloats];
        uint32_t n = (kTuningBuild && (a.ablate & 1)) ? 0u : un;
        asm volatile("" : "+s"(n));  // compared afresh: hoisted out of the slice loop the 32 conditions cost 64 scalar registers
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)KMAX; ++j)
            if (j < n) dma16(rs, dst + j * kWave * 4u, gOff[j], kTuningBuild ? a.loadAux : 0u);
    };
    // a.slots slices per pass: all of them are asked for, then waited for, then reduced one after the other.  One by default: with
    // long buckets LDS bounds the waves of a CU (a second slot for a prefetch halved them and lost), with short ones
    // several slices per wave in flight gained nothing
    if (lane < a.slots) smem[lane * slotFloats + a.slotChunks * 4u] = padding_value<KIND, UNDEF>();
    // every load so far has returned (the builtin, not inline assembly: the compiler's own counter tracking must see these waits, or
    // it waits for the loads of before the loop, behind the DMAs, inside the walk)
    __builtin_amdgcn_s_waitcnt(0x0070);
    asm volatile("" ::: "memory");
    for (uint32_t zp = z0; zp < z1; zp += a.slots) {
        const uint32_t here = min(a.slots, z1 - zp);
        for (uint32_t sl = 0; sl < here; ++sl) dma(sl, zp + sl);
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): the slices of this pass have landed
        asm volatile("" ::: "memory");
        for (uint32_t sl = 0; sl < here; ++sl) {
            const uint32_t z = zp + sl;
            const char* cur = reinterpret_cast<const char*>(&smem[sl * slotFloats]);
            float acc = start_value<KIND, UNDEF>();
            uint32_t cnt = 0;
            // the walk, eight steps at a time: the values of group k + 1 are on their way while group k is reduced
            auto values = [&](int k, float (&v)[kFtGroup]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[2 * i] = cell_value<E>(*reinterpret_cast<const E*>(cur + (tr[4 * k + i] & 0xFFFFu)), a.bad, hasBad);
                    v[2 * i + 1] = cell_value<E>(*reinterpret_cast<const E*>(cur + (tr[4 * k + i] >> 16)), a.bad, hasBad);
                }
                if constexpr (kTyped) {
#pragma unroll
                    for (int i = 0; i < (int)kFtGroup; ++i)
                        v[i] = ((uint32_t)(k * (int)kFtGroup + i) < len) ? v[i] : padding_value<KIND, UNDEF>();
                }
            };
            float v[kFtGroup], vn[kFtGroup];
            values(0, v);
            const bool firstNan = len != 0 && v[0] != v[0];
            const uint32_t walk = (kTuningBuild && (a.ablate & 2)) ? 0u : groups;
#pragma unroll
            for (int k = 0; k < G; ++k) {
                if ((uint32_t)k < walk) {  // wave-uniform
                    if (k + 1 < G) values(k + 1, vn);
#pragma unroll
                    for (uint32_t i = 0; i < kFtGroup; ++i) take<KIND, UNDEF>(v[i], acc, cnt);
#pragma unroll
                    for (uint32_t i = 0; i < kFtGroup; ++i) v[i] = vn[i];
                }
            }
            if (mine) store_result<E>(outBase + (size_t)z * a.nOut + t, finish<KIND, UNDEF>(acc, cnt, len, firstNan), fill);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the reads of this pass are done before the next one overwrites the slots
        asm volatile("" ::: "memory");
    }
    wait_vmcnt<0>();
}

// One wave per (tile, z chunk).  Workgroups are dealt round-robin over the XCDs; XCD x takes the x-th eighth of the tiles, so
// that neighbouring tiles -- which share the source lines at their edges -- use one L2.
// (Measured against it: as many waves as the chip holds, each with an equal run of (tile, slice) pairs -- no last round for a few
// tiles, but all waves of a CU then stage and walk in step, 0.66 ms against 0.57 for the dense case of DESIGN.md 6; workgroups that
// start whenever one ends keep the phases apart.)
template <int KIND, bool UNDEF, int G, int KMAX, typename T = float>
__global__ void __launch_bounds__(kWave) forward_apply_tiled(FtArgs a)
{
    const uint32_t perXcd = gridDim.x / kXcds;
    const uint32_t tile = (blockIdx.x % kXcds) * perXcd + blockIdx.x / kXcds;
    if (tile >= a.g.nTiles) return;
    tile_slices<KIND, UNDEF, G, KMAX, T>(a, tile, a.zStart[blockIdx.y], a.zStart[blockIdx.y + 1]);
}

// G groups of eight steps in registers, KMAX chunks per lane and slice: the smallest form that holds the plan's longest bucket and
// largest tile -- short buckets take few registers and many waves per SIMD, which is what hides the wait for a slice there
template <int KIND, bool UNDEF>
void launch_tiled(const FtArgs& a, uint32_t groups, dim3 grid, size_t lds, hipStream_t stream)
{
    const uint32_t kmax = a.slotChunks / kWave;
    if (groups <= 2 && kmax <= 4) forward_apply_tiled<KIND, UNDEF, 2, 4><<<grid, kWave, lds, stream>>>(a);
    else if (groups <= 4 && kmax <= 8) forward_apply_tiled<KIND, UNDEF, 4, 8><<<grid, kWave, lds, stream>>>(a);
    else if (groups <= 8 && kmax <= 16) forward_apply_tiled<KIND, UNDEF, 8, 16><<<grid, kWave, lds, stream>>>(a);
    else if (groups <= 16) forward_apply_tiled<KIND, UNDEF, 16, 32><<<grid, kWave, lds, stream>>>(a);
    else forward_apply_tiled<KIND, UNDEF, 32, 32><<<grid, kWave, lds, stream>>>(a);
}
// stored types: three forms
template <int KIND, bool UNDEF, typename T>
void launch_tiled_typed(const FtArgs& a, uint32_t groups, dim3 grid, size_t lds, hipStream_t stream)
{
    const uint32_t kmax = a.slotChunks / kWave;
    if (groups <= 4 && kmax <= 8) forward_apply_tiled<KIND, UNDEF, 4, 8, T><<<grid, kWave, lds, stream>>>(a);
    else if (groups <= 16) forward_apply_tiled<KIND, UNDEF, 16, 32, T><<<grid, kWave, lds, stream>>>(a);
    else forward_apply_tiled<KIND, UNDEF, 32, 32, T><<<grid, kWave, lds, stream>>>(a);
}
template <typename T>
bool launch_tiled_typed_kind(const fimex_amd_regrid_plan& plan, const FtArgs& a, uint32_t groups, dim3 grid, size_t lds, hipStream_t stream)
{
    const bool u = plan.undefAggr;
    switch (plan.aggregate) {
    case Aggregate::Sum: u ? launch_tiled_typed<0, true, T>(a, groups, grid, lds, stream) : launch_tiled_typed<0, false, T>(a, groups, grid, lds, stream); break;
    case Aggregate::Mean: u ? launch_tiled_typed<1, true, T>(a, groups, grid, lds, stream) : launch_tiled_typed<1, false, T>(a, groups, grid, lds, stream); break;
    case Aggregate::Max: u ? launch_tiled_typed<3, true, T>(a, groups, grid, lds, stream) : launch_tiled_typed<3, false, T>(a, groups, grid, lds, stream); break;
    case Aggregate::Min: u ? launch_tiled_typed<4, true, T>(a, groups, grid, lds, stream) : launch_tiled_typed<4, false, T>(a, groups, grid, lds, stream); break;
    default: return false;
    }
    return true;
}

}  // namespace

namespace {

// The tiled form of a forward plan (after the CSR) for slices of cellBytes-byte elements: built where the buckets are long on average.
void build_tiles(const fimex_amd_regrid_plan& plan, ForwardTiles& ft, uint32_t cellBytes, size_t& planBytes, hipStream_t stream)
{
    ft.valid = false;
    ft.cellBytes = cellBytes;
    planBytes = 0;
    const size_t nOut = plan.outX * plan.outY;
    const size_t nonEmpty = nOut - plan.info.undefinedCells;
    const double meanBucket = nonEmpty ? (double)plan.info.mappedSourceCells / (double)nonEmpty : 0.0;
    if (plan.aggregate == Aggregate::Median) return;  // the median ranks a bucket's values in registers (forward.hip)
    // from 2.5 cells per bucket on average: 0.1-degree source onto 1/6 degree (2.8 cells) 0.83 ms staged against 0.94 through the lane
    // kernels, onto 1/8 degree (1.6 cells) 1.20 against 0.99, configs[3] (sparse) 0.66 against 0.22
    if (meanBucket * 10.0 < (double)tuning("FWD_TILED_MIN_TENTHS", 25) || tuning("FWD_TILED", 1) == 0) return;
    // Tile shape: the wider the tile, the longer the runs of a source row it stages (what the staging loads like: 1/4-degree targets
    // from 0.1 degree, 4-9 cells per bucket, 100 slices: 0.66 ms with 16 x 4 tiles, 0.58 with 32 x 2, 0.63 with 64 x 1, 0.97 with
    // 8 x 8; 100 cells per bucket: no difference between 16, 32 and 64) -- 32 x 2 for short buckets, 16 x 4 for long ones, whose
    // tiles fill a slot sooner; where that shape cannot be staged (a mapping that turns the grids against each other makes wide tiles
    // span many source rows) the other shapes are tried and the one that stages the fewest chunks is kept.
    const int forced = tuning("FWD_TILE_W", 0);
    const uint32_t first = meanBucket >= 50.0 ? 16u : 32u;
    std::vector<uint32_t> widths = {first, first == 16u ? 32u : 16u, 8, 4};
    if (forced == 4 || forced == 8 || forced == 16 || forced == 32 || forced == 64) widths = {(uint32_t)forced};
    FtGeom g{};
    std::vector<FtTile> h;
    size_t chunks = 0, stepEntries = 0, direct = 0, staged = 0, nTiles = 0;
    uint32_t maxChunks = 0, maxLen = 0;
    bool found = false;
    for (const uint32_t tw : widths) {
        FtGeom c{};
        c.inX = (uint32_t)plan.inX;
        c.outX = (uint32_t)plan.outX;
        c.outY = (uint32_t)plan.outY;
        c.tw = tw;
        c.th = kWave / tw;
        c.cpc = 16u / cellBytes;
        c.cellBytes = cellBytes;
        c.tilesX = (uint32_t)ceil_div(plan.outX, (size_t)c.tw);
        const size_t n = (size_t)c.tilesX * ceil_div(plan.outY, (size_t)c.th);
        if (n > 0x7FFFFFFFu / 64) continue;
        c.nTiles = (uint32_t)n;
        DeviceArray<FtTile> d_tiles(n);
        forward_tile_scan<false><<<dim3(c.nTiles), kWave, 0, stream>>>(c, plan.offsets.get(), plan.src.get(), d_tiles.get(), nullptr, nullptr, 0u);
        FA_HIP(hipGetLastError());
        std::vector<FtTile> hc(n);
        FA_HIP(hipMemcpyAsync(hc.data(), d_tiles.get(), n * sizeof(FtTile), hipMemcpyDeviceToHost, stream));
        FA_HIP(hipStreamSynchronize(stream));
        size_t cChunks = 0, cSteps = 0, cDirect = 0, cStaged = 0;
        uint32_t cMaxChunks = 0, cMaxLen = 0;
        for (auto& T : hc) {
            if (T.maxLen == 0) continue;
            if (T.nChunks == kFtDirect) { cDirect++; continue; }
            cStaged++;
            T.chunkBase = (uint32_t)cChunks;
            T.stepBase = (uint32_t)cSteps;
            cChunks += T.nChunks;
            cSteps += ceil_div((size_t)T.maxLen, (size_t)kFtGroup) * kFtGroup * kWave;
            cMaxChunks = std::max(cMaxChunks, T.nChunks);
            cMaxLen = std::max(cMaxLen, T.maxLen);
        }
        // worth it only where (nearly) every tile can be staged
        if (cStaged == 0 || cDirect * 8 > cStaged || cChunks > 0xFFFFFFFFu || cSteps > 0xFFFFFFFFu) continue;
        if (found && cChunks >= chunks) continue;
        found = true;
        g = c; h.swap(hc); nTiles = n;
        chunks = cChunks; stepEntries = cSteps; direct = cDirect; staged = cStaged; maxChunks = cMaxChunks; maxLen = cMaxLen;
        if (tw == first) break;  // the preferred shape where it works
    }
    if (!found) return;
    ft.tiles.allocate(nTiles);
    FA_HIP(hipMemcpyAsync(ft.tiles.get(), h.data(), nTiles * sizeof(FtTile), hipMemcpyHostToDevice, stream));
    ft.chunkOff.allocate(chunks);
    ft.steps.allocate(stepEntries);
    forward_tile_scan<true><<<dim3(g.nTiles), kWave, 0, stream>>>(g, plan.offsets.get(), plan.src.get(), ft.tiles.get(), ft.chunkOff.get(), ft.steps.get(),
                                                             (uint32_t)ceil_div((size_t)maxChunks, (size_t)kWave) * kWave * 16u);
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(stream));  // h is released on return
    ft.tw = g.tw;
    ft.th = g.th;
    ft.tilesX = g.tilesX;
    ft.nTiles = g.nTiles;
    ft.slotChunks = (uint32_t)ceil_div((size_t)maxChunks, (size_t)kWave) * kWave;
    ft.groups = (uint32_t)ceil_div((size_t)maxLen, (size_t)kFtGroup);
    ft.stagedTiles = staged;
    ft.directTiles = direct;
    ft.stagedChunks = chunks;
    ft.valid = true;
    planBytes = nTiles * sizeof(FtTile) + chunks * sizeof(uint32_t) + stepEntries * sizeof(unsigned short);
}

// everything of a launch but the element type
bool prepare_launch(const fimex_amd_regrid_plan& plan, const ForwardTiles& ft, const void* d_in, size_t nz, void* d_out, FtArgs& a, dim3& grid,
                    size_t& lds)
{
    if (!ft.valid || plan.aggregate == Aggregate::Median || tuning("FWD_TILED", 1) == 0) return false;
    a.in = d_in;
    a.out = d_out;
    a.offsets = plan.offsets.get();
    a.src = plan.src.get();
    a.tiles = ft.tiles.get();
    a.chunkOff = ft.chunkOff.get();
    a.steps = ft.steps.get();
    a.g = FtGeom{(uint32_t)plan.inX, (uint32_t)plan.outX, (uint32_t)plan.outY, ft.tw, ft.th, ft.tilesX, ft.nTiles, 16u / ft.cellBytes, ft.cellBytes};
    a.nOut = (uint32_t)(plan.outX * plan.outY);
    a.inLayer = plan.inX * plan.inY;
    a.inBytes = (uint32_t)(a.inLayer * ft.cellBytes);
    a.nz = (uint32_t)nz;
    a.slotChunks = ft.slotChunks;
    // slices per pass (see tile_slices): one.  Several (tuning build) were no faster where a slice of a tile is small (1/4-degree
    // targets, 2 KB per slice: 0.69 ms with one or two, 0.71 with four, 0.73 with eight): the staging loads' run length is what
    // bounds those launches, not a wave's wait
    const size_t slotBytes = ((size_t)ft.slotChunks + 1) * 16;  // + the padding cell's chunk
    int slots = tuning("FWD_TILED_SLOTS", 0);
    if (slots < 1 || slots > 8) slots = 1;
    slots = (int)std::min<size_t>((size_t)slots, std::max<size_t>(nz, 1));
    a.slots = (uint32_t)slots;
    a.ablate = (uint32_t)tuning("FWD_TILED_ABLATE", 0);
    a.loadAux = (uint32_t)tuning("FWD_TILED_AUX", 0);
    lds = (size_t)slots * slotBytes;
    // z chunks: step table and chunk list are loaded once per (tile, chunk) -- about as many bytes as one slice of the tile -- so
    // chunks are long, many more workgroups than the chip holds at once (1536 with six per CU) all the same, and the LAST chunks
    // (workgroups start in the order of their indices) are short, so that the last round, which few workgroups run, is short too:
    // three quarters of the slices in chunks of `big`, then halves of it down to four slices
    size_t big = std::max<size_t>(4, ceil_div(nz * (size_t)ft.nTiles, (size_t)tuning("FWD_TILED_WAVES", 6144)));
    big = std::max(big, ceil_div(nz, (size_t)(kFtMaxZChunks - 16)));  // the tapered tail adds at most log2(big) chunks
    big = std::min<size_t>(big, std::max<size_t>(nz, 1));
    size_t chunks = 0;
    {
        size_t z = 0, len = big;
        a.zStart[0] = 0;
        while (z < nz) {
            if (tuning("FWD_TILED_TAPER", 1) != 0 && len > 4 && (nz - z) * 4 <= nz + 3 && (nz - z) <= 2 * len) len = std::max<size_t>(4, len / 2);
            z = std::min(nz, z + len);
            if (chunks + 1 >= kFtMaxZChunks) z = nz;  // the last chunk takes what is left
            a.zStart[++chunks] = (uint32_t)z;
        }
    }
    grid = dim3((uint32_t)(ceil_div((size_t)ft.nTiles, (size_t)kXcds) * kXcds), (uint32_t)chunks, 1);
    return true;
}

}  // namespace

void build_forward_tiles(fimex_amd_regrid_plan& plan, hipStream_t stream)
{
    size_t bytes = 0;
    build_tiles(plan, plan.fwdTiles, 4, bytes, stream);
    if (!plan.fwdTiles.valid) return;
    plan.info.stagedCells = plan.fwdTiles.stagedChunks * 4;
    plan.info.tileW = plan.fwdTiles.tw;
    plan.info.tileH = plan.fwdTiles.th;
    plan.info.planBytes += bytes;
}

bool launch_forward_tiled(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream)
{
    const ForwardTiles& ft = plan.fwdTiles;
    FtArgs a{};
    dim3 grid;
    size_t lds = 0;
    if (!prepare_launch(plan, ft, d_in, nz, d_out, a, grid, lds)) return false;
    const uint32_t groups = ft.groups;  // of the plan's longest bucket
    const bool u = plan.undefAggr;
    switch (plan.aggregate) {
    case Aggregate::Sum: u ? launch_tiled<0, true>(a, groups, grid, lds, stream) : launch_tiled<0, false>(a, groups, grid, lds, stream); break;
    case Aggregate::Mean: u ? launch_tiled<1, true>(a, groups, grid, lds, stream) : launch_tiled<1, false>(a, groups, grid, lds, stream); break;
    case Aggregate::Max: u ? launch_tiled<3, true>(a, groups, grid, lds, stream) : launch_tiled<3, false>(a, groups, grid, lds, stream); break;
    case Aggregate::Min: u ? launch_tiled<4, true>(a, groups, grid, lds, stream) : launch_tiled<4, false>(a, groups, grid, lds, stream); break;
    default: return false;
    }
    FA_HIP(hipGetLastError());
    return true;
}

// Slices in a 1- or 2-byte stored type through the same kernel: half or a quarter of the bytes are staged, elements
// become floats as they are read from LDS (data2InterpolationArray) and results elements as they are stored
// (interpolationArray2Data).  The plan's tiled form for that element size is built on the first such call, under the mutex
// (plans are shared by threads).  false: not this path (the caller converts, applies, converts back).
bool launch_forward_tiled_typed(const fimex_amd_regrid_plan& plan, const void* d_in, int cdmType, size_t nz, double badValue, void* d_out,
                                hipStream_t stream)
{
    if (plan.kind != PlanKind::Forward || plan.aggregate == Aggregate::Median || !plan.fwdTiles.valid || tuning("TYPED_FORWARD", 1) == 0) return false;
    uint32_t eb = 0;
    switch (cdmType) {
    case FIMEX_AMD_CDM_CHAR: case FIMEX_AMD_CDM_UCHAR: eb = 1; break;
    case FIMEX_AMD_CDM_SHORT: case FIMEX_AMD_CDM_USHORT: eb = 2; break;
    default: return false;
    }
    if (nz == 0) return true;
    auto& forms = plan.fwdTyped;
    ForwardTiles* ft = nullptr;
    {
        std::lock_guard<std::mutex> lock(forms.mtx);
        const int k = eb == 2 ? 0 : 1;
        if (!forms.tried[k]) {
            size_t bytes = 0;
            build_tiles(plan, forms.form[k], eb, bytes, stream);
            forms.tried[k] = true;
        }
        ft = &forms.form[k];
    }
    FtArgs a{};
    dim3 grid;
    size_t lds = 0;
    if (!prepare_launch(plan, *ft, d_in, nz, d_out, a, grid, lds)) return false;
    a.bad = (float)badValue;
    a.hasBad = !(a.bad != a.bad);
    a.fillOut = badValue;
    bool ok = false;
    switch (cdmType) {
    case FIMEX_AMD_CDM_CHAR: ok = launch_tiled_typed_kind<signed char>(plan, a, ft->groups, grid, lds, stream); break;
    case FIMEX_AMD_CDM_UCHAR: ok = launch_tiled_typed_kind<unsigned char>(plan, a, ft->groups, grid, lds, stream); break;
    case FIMEX_AMD_CDM_SHORT: ok = launch_tiled_typed_kind<short>(plan, a, ft->groups, grid, lds, stream); break;
    default: ok = launch_tiled_typed_kind<unsigned short>(plan, a, ft->groups, grid, lds, stream); break;
    }
    FA_HIP(hipGetLastError());
    return ok;
}

}  // namespace fimex_amd
