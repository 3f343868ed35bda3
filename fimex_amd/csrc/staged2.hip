// LDS-staged backward regrid, second form: the bandwidth path of the headline metric from round 2 on.
//
// Same arithmetic as staged.hip / regrid.hip (src/interpolation.c:862-1028) and the same idea -- a workgroup owns a tile of
// OUTPUT cells, streams the source row segments that tile needs HBM -> LDS by LDS-DMA for slice z + d while slice z is
// interpolated out of LDS -- with the structure that the access calibration of round 2 asks for
// (scripts/calib/stream_pattern.hip, profiles/calib/r02_stream_pattern_*.jsonl):
//   * workgroups of 256, 512 or 1024 threads on tiles 2-4 times as large: the lines at the ends of a row segment and the
//     halo rows are shared with the neighbouring tile, and only tiles that run in lockstep share them for certain;
//   * tiles of one height and VARYING width: an output cell of the benchmark plan covers 1.6 .. 3.3 source columns, so a
//     uniform tile grid sizes every LDS slot for the worst tile and leaves 40 % of it unused.  Here the plan narrows the
//     tiles of a tile row until every tile fits the same budget;
//   * one copy of the slice loop per number of DMA instructions a lane issues per slice (1 .. KMAX, chosen per tile), so that
//     no instruction is issued for chunks a tile does not have and every s_waitcnt keeps an immediate count;
//   * the plan stores the source offset of every 16-byte chunk of a tile, so the workgroup prologue is a coalesced load
//     instead of a binary search per chunk, and row segments start on 16-byte boundaries of the slice for any row length
//     (inX % 4 != 0 included: a reduced domain, src/CachedInterpolation.cc:159-200, crops to arbitrary widths);
//   * tile rows are dealt to the XCDs in stripes (neighbours in x share an L2) through a workgroup -> tile table.
#include "plan.hpp"
#include "staged_common.hpp"
#include "typed_convert.hpp"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace fimex_amd {

namespace {

// One workgroup per tile.  emit == 0: counts the 16-byte chunks of the tile's row segments (tiles[t].nChunks, ~0u = does not
// fit).  emit == 1: writes the chunk list and every output cell's LDS offsets (16 bits per stencil row, in floats).
// Where the stencil of an output cell lies: from the caller's positions (plan creation), or -- for the forms of stored types,
// which are built on first use, long after the positions are gone -- from the gather plan the positions were turned into
// (regrid.hip: pos = first cell of the stencil, the sign bits of xf / yf = "one column" / "one row", src/interpolation.c:903-948).
struct NeedSource {
    const double* px = nullptr;
    const double* py = nullptr;
    const uint32_t* pos = nullptr;
    const float* xf = nullptr;
    const float* yf = nullptr;
};

template <int STENCIL>
__device__ __forceinline__ CellNeed need_of(const NeedSource& n, size_t cell, int64_t ix, int64_t iy)
{
    if (n.px != nullptr) return classify<STENCIL>(n.px[cell], n.py[cell], ix, iy);
    CellNeed c{};
    const uint32_t p = n.pos[cell];
    c.valid = p != kInvalidPos;
    if (!c.valid) return c;
    c.ya = (int64_t)(p / (uint32_t)ix);
    c.xa = (int64_t)p - c.ya * ix;
    if (STENCIL == 1) { c.xb = c.xa; c.yb = c.ya; }
    else if (STENCIL == 2) {
        c.xb = c.xa + ((__float_as_uint(n.xf[cell]) >> 31) ? 0 : 1);
        c.yb = c.ya + ((__float_as_uint(n.yf[cell]) >> 31) ? 0 : 1);
    } else { c.xb = c.xa + 3; c.yb = c.ya + 3; }
    return c;
}

// cpc: source cells per 16-byte chunk (4 for float slices, 8 / 16 for slices of 2- / 1-byte elements); LDS offsets count elements.
template <int STENCIL>
__global__ void __launch_bounds__(kBlock) tile_scan(NeedSource need, int64_t ix, int64_t iy,
                                                    uint32_t outX, uint32_t outY, uint32_t tileH, StagedTile* __restrict__ tiles,
                                                    uint32_t capChunks, int emit, uint32_t* __restrict__ chunkOff,
                                                    uint32_t* __restrict__ ldsA, uint32_t* __restrict__ ldsB, uint32_t cpc)
{
    __shared__ int shRmin, shRmax, shFail;
    __shared__ int rowMin[kMaxRows], rowMax[kMaxRows];
    __shared__ uint32_t rowChunk[kMaxRows + 1];
    const uint32_t t = blockIdx.x;
    const StagedTile T = tiles[t];
    if (T.rsv[0] != 0) return;  // not staged (see build_shape)
    const uint32_t nCells = T.w * tileH;
    if (threadIdx.x == 0) { shRmin = 0x7FFFFFFF; shRmax = -1; shFail = 0; }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nCells; e += kBlock) {
        const uint32_t y = T.y0 + e / T.w, x = T.x0 + e % T.w;
        if (y >= outY) continue;
        const size_t cell = (size_t)y * outX + x;
        const CellNeed c = need_of<STENCIL>(need, cell, ix, iy);
        if (c.valid) { atomicMin(&shRmin, (int)c.ya); atomicMax(&shRmax, (int)c.yb); }
    }
    __syncthreads();
    const int rmin = shRmin;
    const int nr = (shRmax >= rmin) ? shRmax - rmin + 1 : 0;
    if (nr > kMaxRows) {
        if (threadIdx.x == 0 && !emit) tiles[t].nChunks = 0xFFFFFFFFu;
        return;
    }
    for (int i = threadIdx.x; i < nr; i += kBlock) { rowMin[i] = 0x7FFFFFFF; rowMax[i] = -0x7FFFFFFF; }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nCells; e += kBlock) {
        const uint32_t y = T.y0 + e / T.w, x = T.x0 + e % T.w;
        if (y >= outY) continue;
        const size_t cell = (size_t)y * outX + x;
        const CellNeed c = need_of<STENCIL>(need, cell, ix, iy);
        if (c.valid)
            for (int64_t r = c.ya; r <= c.yb; ++r) {
                atomicMin(&rowMin[r - rmin], (int)c.xa);
                atomicMax(&rowMax[r - rmin], (int)c.xb);
            }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int64_t layer = ix * iy;
        uint32_t acc = 0;
        for (int i = 0; i < nr; ++i) {
            rowChunk[i] = acc;
            if (rowMax[i] >= rowMin[i]) {
                // the segment starts on a 16-byte boundary of the SLICE (the DMA moves 16 bytes per lane; aligned pieces stay
                // inside one line): it may begin up to 3 cells before the first cell needed, in the row above for x < 0
                const int64_t first = (int64_t)(rmin + i) * ix + rowMin[i];
                int64_t start = first & ~(int64_t)(cpc - 1);
                const uint32_t nch = (uint32_t)((first - start + (rowMax[i] - rowMin[i])) / cpc + 1);
                // a last chunk that would cross the end of the slice is moved back instead (unaligned, still whole; slices of
                // stored types hold a multiple of 4 bytes, so the chunk still starts on a 4-byte boundary)
                if (start + cpc * (int64_t)nch > layer) start = layer - cpc * (int64_t)nch;
                if (start < 0) shFail = 1;
                rowMin[i] = (int)(start - (int64_t)(rmin + i) * ix);  // column of the segment's first cell, may be negative
                acc += nch;
            }
        }
        rowChunk[nr] = acc;
        if (acc > capChunks) shFail = 1;
    }
    __syncthreads();
    if (!emit) {
        if (threadIdx.x == 0) tiles[t].nChunks = shFail ? 0xFFFFFFFFu : rowChunk[nr];
        return;
    }
    const uint32_t total = rowChunk[nr];
    for (uint32_t c = threadIdx.x; c < total; c += kBlock) {
        uint32_t lo = 0, hi = (uint32_t)nr - 1;  // last row whose first chunk <= c and that holds chunks
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (rowChunk[mid] <= c) lo = mid; else hi = mid - 1;
        }
        chunkOff[T.chunkBase + c] = (uint32_t)((int64_t)(rmin + (int)lo) * ix + rowMin[lo] + cpc * (int64_t)(c - rowChunk[lo]));
    }
    for (uint32_t e = threadIdx.x; e < nCells; e += kBlock) {
        const uint32_t y = T.y0 + e / T.w, x = T.x0 + e % T.w;
        if (y >= outY) continue;
        const size_t cell = (size_t)y * outX + x;
        const CellNeed c = need_of<STENCIL>(need, cell, ix, iy);
        uint32_t a = kInvalidPos, b = kInvalidPos;
        if (c.valid) {
            uint32_t off[4];
            for (int r = 0; r < 4; ++r) {
                const int64_t row = (c.ya + r <= c.yb) ? c.ya + r : c.yb;  // missing rows repeat the last one
                const int i = (int)(row - rmin);
                off[r] = rowChunk[i] * cpc + (uint32_t)(c.xa - rowMin[i]);
            }
            a = off[0] | (off[1] << 16);
            b = off[2] | (off[3] << 16);
        }
        ldsA[cell] = a;
        if (STENCIL == 4) ldsB[cell] = b;
    }
}

constexpr int kMaxZChunks = 31;

struct Staged2Args {
    const float* in;
    float* out;
    const StagedTile* tiles;
    const uint32_t* order;
    const uint32_t* chunkOff;
    const uint32_t* ldsA;
    const uint32_t* ldsB;
    const uint32_t* pos;  // gather plan (regrid.hip): source cell of the stencil's corner, for the tiles that are not staged
    const float* xf;
    const float* yf;
    const double* xfd;
    const double* yfd;
    uint32_t outX, outY, tileH;
    uint32_t inX;
    uint32_t inBytes;    // one source slice
    uint32_t nOut;
    uint32_t nz;
    uint32_t zStart[kMaxZChunks + 1];  // slices [zStart[c], zStart[c + 1]) belong to z chunk c = blockIdx.y
    uint32_t nZChunks;    // > 0: flat grid, the z chunks of a tile are consecutive workgroups of one XCD (see launch_staged2_apply)
    uint32_t slotChunks;  // 16-byte chunks of one slot of the slice ring (a multiple of 64: whole wave instructions)
    uint32_t flags;      // tuning build only: 1 no source loads, 2 no result stores
};

// Result stores of the slice loop: non-temporal and written through (sc1 nt): 0.4-1 % faster than nt alone in six placements of the
// output out of six (profiles/calib/r02_store_policy.jsonl), plain stores 2-6 % slower.  Tuning build: flags 8 plain, 16 nt, 32 sc0 nt.
__device__ __forceinline__ void store_result(uint32_t bits, rsrc_t ro, uint32_t off, uint32_t flags)
{
    if (kTuningBuild && (flags & 56u)) {
        if (flags & 8u) __builtin_amdgcn_raw_buffer_store_b32(bits, ro, off, 0, 0);
        else if (flags & 16u) __builtin_amdgcn_raw_buffer_store_b32(bits, ro, off, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b32(bits, ro, off, 0, 3);
        return;
    }
    __builtin_amdgcn_raw_buffer_store_b32(bits, ro, off, 0, 18);
}

// STENCIL: 1 nearest, 2 bilinear, 4 bicubic; NT: threads of the workgroup; PER: outputs per lane (tile = NT * PER outputs);
// KMAX: most 16-byte chunks a lane stages per slice.
// FAST (bicubic only): the weights rounded to float and float fused multiply-adds instead of the reference's double products
// accumulated into a float (interpolation.c:1005-1019) -- not bit-identical, within 1e-5 of the stencil's magnitude (the
// tolerance BASELINE.json states), chosen per plan (FIMEX_AMD_BICUBIC_FAST).
template <int STENCIL, int NT, int PER, int KMAX, bool FAST = false, int DEPTH = 2>
__global__ void __launch_bounds__(NT) staged_apply2(Staged2Args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint32_t slot0 = blockIdx.x, zc = blockIdx.y;
    if (a.nZChunks != 0) {  // workgroup s runs on XCD s % 8; the k-th workgroup of an XCD is z chunk k % n of the XCD's tile k / n
        const uint32_t k = blockIdx.x / kXcds;
        zc = k % a.nZChunks;
        slot0 = (k / a.nZChunks) * kXcds + blockIdx.x % kXcds;
    }
    const uint32_t tile = a.order[slot0];
    if (tile == 0xFFFFFFFFu) return;
    const StagedTile T = a.tiles[tile];
    const uint32_t z0 = a.zStart[zc];
    const uint32_t z1 = a.zStart[zc + 1];
    const uint32_t nzl = z1 - z0;

    const uint32_t outBytes = a.nOut * 4u;
    const char* inBase = reinterpret_cast<const char*>(a.in);
    char* outBase = reinterpret_cast<char*>(a.out);
    const uint32_t outRecords = (kTuningBuild && (a.flags & 2)) ? 0u : outBytes;

    // ---- the staging list and the first DMAs come before everything else: the per-output plan below loads while they fly
    // per-lane staging list: chunk c = threadIdx.x + j * NT of the tile's list, un = DMA instructions per lane and slice
    const uint32_t waveChunk = (threadIdx.x / kWave) * kWave;
    uint32_t gOff[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        const uint32_t c = threadIdx.x + j * NT;
        gOff[j] = (c < T.nChunks) ? a.chunkOff[T.chunkBase + c] * 4u : 0xFFFFFFFFu;  // ~0u: dropped by the bounds check (zeros)
    }
    const uint32_t un = (T.nChunks + NT - 1) / NT;
    // DEPTH slots, each holds the largest tile of the plan, then 1 KiB that is never read.  A lane without a chunk carries an
    // offset beyond the slice and the DMA writes ZEROS for it (scripts/calib/dma_oob.hip): inside the slot that is unused
    // space, but a slot is not a whole number of NT chunks, and the last wave instructions of a full tile would run past
    // its end into the next slot -- the slice being interpolated.  Such an instruction (whole: slots are multiples of 64
    // chunks) is pointed at the spare KiB instead of being left out, so every wave issues the same number of them.
    const uint32_t slotFloats = a.slotChunks * 4u;
    float* const spare = smem + DEPTH * slotFloats;
    auto dma_dst = [&](uint32_t sl, int j) {
        const uint32_t c = waveChunk + (uint32_t)j * NT;
        return c < a.slotChunks ? smem + sl * slotFloats + c * 4u : spare;
    };
    const uint32_t inRecords = (kTuningBuild && (a.flags & 1)) ? 0u : a.inBytes;

    // prologue: DEPTH - 1 slices in flight (issued here, before the per-output plan is loaded)
    for (uint32_t i = 0; i + 1 < (uint32_t)DEPTH && i < nzl; ++i) {
        const rsrc_t rs = make_rsrc(inBase + (size_t)(z0 + i) * a.inBytes, inRecords);
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if ((uint32_t)j < un) dma16(rs, dma_dst(i, j), gOff[j]);
    }
    // ---- per-lane plan: outputs e = threadIdx.x + k * NT of the tile (a wave covers 64 consecutive cells of one row)
    uint32_t cellOff[PER];
    uint32_t row[PER][STENCIL];
    float xf[PER], yf[PER];
    double XM[PER][4], MY[PER][4];
    float XMf[PER][4], MYf[PER][4];
    bool undef[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t e = threadIdx.x + k * NT;
        const uint32_t ly = e / T.w, lx = e - ly * T.w;
        const uint32_t y = T.y0 + ly;
        cellOff[k] = 0xFFFFFFFFu;
        uint32_t pa = kInvalidPos, pb = kInvalidPos;
        xf[k] = yf[k] = 0.f;
        double fx = 0, fy = 0;
        if (ly < a.tileH && y < a.outY) {
            const uint32_t cell = y * a.outX + T.x0 + lx;
            cellOff[k] = cell * 4u;
            pa = a.ldsA[cell];
            if (STENCIL == 4) { pb = a.ldsB[cell]; fx = a.xfd[cell]; fy = a.yfd[cell]; }
            else if (STENCIL == 2) { xf[k] = a.xf[cell]; yf[k] = a.yf[cell]; }
        }
        undef[k] = pa == kInvalidPos;  // undefined cells read LDS offset 0 and discard it
        row[k][0] = undef[k] ? 0u : (pa & 0xFFFFu) * 4u;
        if (STENCIL >= 2) row[k][STENCIL >= 2 ? 1 : 0] = undef[k] ? 0u : (pa >> 16) * 4u;
        if (STENCIL == 4) {
            row[k][2] = undef[k] ? 0u : (pb & 0xFFFFu) * 4u;
            row[k][3] = undef[k] ? 0u : (pb >> 16) * 4u;
            cubic_weights(fx, XM[k]);
            cubic_weights(fy, MY[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { XMf[k][j] = (float)XM[k][j]; MYf[k][j] = (float)MY[k][j]; }
        }
    }
    if (T.rsv[0] != 0) {
        // A tile whose footprint does not fit a slot even at the smallest width (an outlier among its positions: a grid
        // that wraps around the date line, isolated special points) reads its stencils straight from memory, like the
        // gather kernels of regrid.hip; every other tile of the plan stays staged.
        uint32_t p[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            p[k] = (cellOff[k] != 0xFFFFFFFFu) ? a.pos[cellOff[k] / 4u] : kInvalidPos;
            undef[k] = p[k] == kInvalidPos;
            if (undef[k]) p[k] = 0;
        }
        const uint32_t inRec = (kTuningBuild && (a.flags & 1)) ? 0u : a.inBytes;
        // (the slice pointers are wave-uniform; said explicitly, or the compiler loops over the lanes' descriptors)
        auto uniform = [](const char* ptr) {
            const uint64_t v = reinterpret_cast<uint64_t>(ptr);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        };
        for (uint32_t z = z0; z < z1; ++z) {
            const rsrc_t rs = make_rsrc(uniform(inBase + (size_t)z * a.inBytes), inRec);
            const rsrc_t ro = make_rsrc(uniform(outBase + (size_t)z * outBytes), outRecords);
            auto ld = [&](uint32_t cell) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, cell * 4u, 0, 0)); };
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                float r;
                if constexpr (STENCIL == 1) {
                    r = ld(p[k]);
                } else if constexpr (STENCIL == 2) {
                    const bool nnx = (__float_as_uint(xf[k]) >> 31) != 0, nny = (__float_as_uint(yf[k]) >> 31) != 0;
                    const uint32_t dx = nnx ? 0u : 1u, dy = nny ? 0u : a.inX;  // a missing neighbour repeats the cell itself
                    const float s00 = ld(p[k]), s01 = ld(p[k] + dx), s10 = ld(p[k] + dy), s11 = ld(p[k] + dx + dy);
                    const float top = (1.f - xf[k]) * s00 + xf[k] * s01;
                    const float bot = (1.f - xf[k]) * s10 + xf[k] * s11;
                    const float inter = (1.f - yf[k]) * top + yf[k] * bot;
                    const float liny = (1 - yf[k]) * s00 + (yf[k] * s10);
                    r = nnx ? (nny ? s00 : liny) : (nny ? top : inter);
                } else {
                    float f[4][4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int j = 0; j < 4; ++j) f[q][j] = ld(p[k] + q * a.inX + j);
                    float acc = 0;
                    if constexpr (FAST) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float xmf = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j) xmf = __builtin_fmaf(XMf[k][j], f[q][j], xmf);
                            acc = __builtin_fmaf(xmf, MYf[k][q], acc);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            double xmf = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j) xmf += XM[k][j] * (double)f[q][j];
                            acc = (float)((double)acc + xmf * MY[k][q]);
                        }
                    }
                    r = acc;
                }
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(undef[k] ? undefined_f() : r), ro, cellOff[k], 0, 2);
            }
        }
        return;
    }
    if (T.nChunks == 0) {  // nothing of the source is needed: every output of the tile is undefined
        for (uint32_t z = z0; z < z1; ++z) {
            const rsrc_t ro = make_rsrc(outBase + (size_t)z * outBytes, outRecords);
#pragma unroll
            for (int k = 0; k < PER; ++k) __builtin_amdgcn_raw_buffer_store_b32(0x7fc00000u, ro, cellOff[k], 0, 2);
        }
        return;
    }

    // Main loop with the number of DMA instructions per slice as a compile-time constant (one copy of the loop per value), so
    // that every wait is an immediate: results come back in issue order, and behind the DMA of slice i + 1 the DMAs of the
    // slices i + 2 .. i + DEPTH - 1 and the stores of the last DEPTH - 1 iterations may stay in flight.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first DEPTH - 1 slices (a workgroup's first wait only)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // A wave whose outputs are all interior cells of the source (nothing undefined, no border branch: nearly every wave)
    // runs a copy of the loop without the selections between the border forms.
    bool plainWave = true;
#pragma unroll
    for (int k = 0; k < PER; ++k)
        plainWave = plainWave && !undef[k] && (STENCIL != 2 || ((__float_as_uint(xf[k]) | __float_as_uint(yf[k])) >> 31) == 0);
    plainWave = __all(plainWave) != 0;
    // selection masks of the border forms (interpolation.c:903-948) for the other copy: all ones / all zeros per output
    uint32_t mNnx[PER], mNny[PER], mUndef[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        mNnx[k] = (uint32_t)((int32_t)__float_as_uint(xf[k]) >> 31);
        mNny[k] = (uint32_t)((int32_t)__float_as_uint(yf[k]) >> 31);
        mUndef[k] = undef[k] ? 0xFFFFFFFFu : 0u;
    }
    auto pick = [](uint32_t mask, float a, float b) { return __uint_as_float((__float_as_uint(a) & mask) | (__float_as_uint(b) & ~mask)); };
    auto run = [&](auto unTag, auto plainTag) __attribute__((always_inline)) {
        constexpr int UN = decltype(unTag)::value;
        constexpr bool PLAIN = decltype(plainTag)::value;
        uint32_t slot = 0;
        for (uint32_t i = 0; i < nzl; ++i) {
            const uint32_t z = z0 + i;
            const bool more = i + DEPTH - 1 < nzl;
            if (more) {  // into the slot slice i - 1 has left
                const uint32_t sl = (slot + DEPTH - 1 >= (uint32_t)DEPTH) ? slot - 1 : slot + DEPTH - 1;
                const rsrc_t rs = make_rsrc(inBase + (size_t)(z + DEPTH - 1) * a.inBytes, inRecords);
#pragma unroll
                for (int j = 0; j < UN; ++j) dma16(rs, dma_dst(sl, j), gOff[j]);
            }
            const char* curb = reinterpret_cast<const char*>(smem + slot * slotFloats);
            const rsrc_t ro = make_rsrc(outBase + (size_t)z * outBytes, outRecords);
            if constexpr (STENCIL == 1) {
                float v[PER];
    #pragma unroll
                for (int k = 0; k < PER; ++k) v[k] = *reinterpret_cast<const float*>(curb + row[k][0]);
    #pragma unroll
                for (int k = 0; k < PER; ++k)  // src/interpolation.c:869-876
                    store_result(__float_as_uint(PLAIN ? v[k] : pick(mUndef[k], undefined_f(), v[k])), ro, cellOff[k], a.flags);
            } else if constexpr (STENCIL == 2) {
                float s00[PER], s01[PER], s10[PER], s11[PER];
    #pragma unroll
                for (int k = 0; k < PER; ++k) {  // all stencil reads first: 2 x ds_read2_b32 per output, no waits in between
                    const float* pa = reinterpret_cast<const float*>(curb + row[k][0]);
                    const float* pb = reinterpret_cast<const float*>(curb + row[k][1]);
                    s00[k] = pa[0]; s01[k] = pa[1]; s10[k] = pb[0]; s11[k] = pb[1];
                }
    #pragma unroll
                for (int k = 0; k < PER; ++k) {
                    // interior (interpolation.c:899-900); its upper row is the "linear in x, nearest in y" value (:911)
                    const float top = (1.f - xf[k]) * s00[k] + xf[k] * s01[k];
                    const float bot = (1.f - xf[k]) * s10[k] + xf[k] * s11[k];
                    const float inter = (1.f - yf[k]) * top + yf[k] * bot;
                    float r = inter;
                    if constexpr (!PLAIN) {  // every form is computed, bit masks pick one: no divergent branches in the loop
                        const float liny = (1 - yf[k]) * s00[k] + (yf[k] * s10[k]);  // nearest in x, linear in y (:931)
                        r = pick(mNnx[k], pick(mNny[k], s00[k], liny), pick(mNny[k], top, inter));
                        r = pick(mUndef[k], undefined_f(), r);
                    }
                    store_result(__float_as_uint(r), ro, cellOff[k], a.flags);
                }
            } else if constexpr (FAST) {
                // float arithmetic: the stencil's columns are summed first, two at a time in packed FMAs -- the two floats a
                // ds_read2_b32 delivers are one operand, the row's weight is the other (both halves) -- then the four column sums
                // meet the x weights: 11 packed / scalar instructions per output instead of 20 FMAs.  The launch is bound by its
                // instructions (LDS reads and arithmetic: 1.9 of 2.5 ms with the memory instructions switched off,
                // profiles/r03_ablate_bicubic_fast.log), so this is where its time is.
                using v2f = float __attribute__((ext_vector_type(2)));
                // (the four floats of a stencil row are read as two ds_read2_b32: 8-byte LDS reads at 4-byte addresses work on this
                // hardware but take five times as long, scripts/calib/lds_unaligned.hip, profiles/calib/r03_lds_unaligned.jsonl)
    #pragma unroll
                for (int k = 0; k < PER; ++k) {
                    v2f c01 = {0.f, 0.f}, c23 = {0.f, 0.f};
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* fr = reinterpret_cast<const float*>(curb + row[k][r]);
                        const v2f f01 = {fr[0], fr[1]};
                        const v2f f23 = {fr[2], fr[3]};
                        const v2f myPair = {MYf[k][r & 2], MYf[k][(r & 2) + 1]};  // (a register pair; the packed FMA takes one half twice)
                        const v2f my = (r & 1) ? __builtin_shufflevector(myPair, myPair, 1, 1) : __builtin_shufflevector(myPair, myPair, 0, 0);
                        c01 = __builtin_elementwise_fma(f01, my, c01);
                        c23 = __builtin_elementwise_fma(f23, my, c23);
                    }
                    const v2f x01 = {XMf[k][0], XMf[k][1]}, x23 = {XMf[k][2], XMf[k][3]};
                    const v2f p = __builtin_elementwise_fma(x23, c23, x01 * c01);
                    const float acc = p.x + p.y;
                    store_result(__float_as_uint(PLAIN ? acc : pick(mUndef[k], undefined_f(), acc)), ro, cellOff[k], a.flags);
                }
            } else {
    #pragma unroll
                for (int k = 0; k < PER; ++k) {
                    float f[4][4];
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
    #pragma unroll
                        for (int j = 0; j < 4; ++j) f[r][j] = *reinterpret_cast<const float*>(curb + row[k][r] + 4 * j);
                    }
                    float acc = 0;  // interpolation.c:1005: accumulates into the float output
                    {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            double xmf = 0;
    #pragma unroll
                            for (int j = 0; j < 4; ++j) xmf += XM[k][j] * (double)f[r][j];  // :1015
                            acc = (float)((double)acc + xmf * MY[k][r]);                    // :1019
                        }
                    }
                    store_result(__float_as_uint(PLAIN ? acc : pick(mUndef[k], undefined_f(), acc)), ro, cellOff[k], a.flags);
                }
            }
            if (more) wait_vmcnt<(DEPTH - 2) * UN + (DEPTH - 1) * PER>();
            else wait_vmcnt<PER>();  // the tail of the z chunk: everything but this slice's stores
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            slot = (slot + 1 == (uint32_t)DEPTH) ? 0 : slot + 1;
        }
    };
    static_assert(KMAX <= 8, "one copy of the loop per DMA count");
    auto run_un = [&](auto unTag) __attribute__((always_inline)) {
        if (plainWave) run(unTag, std::true_type());
        else run(unTag, std::false_type());
    };
    switch (un) {
    case 1: run_un(std::integral_constant<int, 1>()); break;
    case 2: run_un(std::integral_constant<int, KMAX >= 2 ? 2 : 1>()); break;
    case 3: run_un(std::integral_constant<int, KMAX >= 3 ? 3 : 1>()); break;
    case 4: run_un(std::integral_constant<int, KMAX >= 4 ? 4 : 1>()); break;
    case 5: run_un(std::integral_constant<int, KMAX >= 5 ? 5 : 1>()); break;
    case 6: run_un(std::integral_constant<int, KMAX >= 6 ? 6 : 1>()); break;
    case 7: run_un(std::integral_constant<int, KMAX >= 7 ? 7 : 1>()); break;
    default: run_un(std::integral_constant<int, KMAX >= 8 ? 8 : 1>()); break;
    }
}

// ---- the same scheme on a variable's STORED type (data2InterpolationArray + interpolateValues +
// interpolationArray2Data of src/CDMInterpolator.cc:115-124, 251-285 in one kernel): slices of 1- or 2-byte integers.
// The plan form is its own (built on first use, staged2_typed_form): a 16-byte chunk holds 8 or 16 source cells, LDS offsets
// count elements.  Differences to the float kernel: a lane owns two PAIRS of neighbouring outputs (cells 2 * t, 2 * t + 1 of
// the tile, and the same NT * 2 cells further on), so that two results leave in one 4-byte (2-byte elements) or 2-byte store
// and a wave still writes 256 (128) contiguous bytes; the two source elements of a stencil row arrive in one ds_read2_b32 and
// are shifted apart; elements become float / NaN as Data::asFloat + mifi_bad2nanf do, results go back through ScaleValue's
// rounding (typed_convert.hpp).
struct TypedEdge {
    float bad;          // the variable's fill value narrowed to float (mifi_bad2nanf's argument)
    uint32_t hasBad;
    double fillOut;     // NaN -> this (interpolationArray2Data)
    uint32_t pairStore; // outX even: the two results of a pair share one store
};

// two neighbouring 1- or 2-byte elements at element offset `byteOff / sizeof(T)` of the staged image
// (alignedOff = byteOff & ~3; shift = byteOff * 8: v_alignbit_b32 takes the low five bits, (byteOff & 3) * 8 -- both are
// computed once per lane, outside the slice loop)
template <typename T>
__device__ __forceinline__ void lds_pair2(const char* buf, uint32_t alignedOff, uint32_t shift, float bad, bool hasBad, float& first, float& second)
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(buf + alignedOff);
    const uint32_t both = __builtin_amdgcn_alignbit(w[1], w[0], shift);
    if constexpr (sizeof(T) == 2) {
        first = as_float_nan((T)(unsigned short)(both & 0xffffu), bad, hasBad);
        second = as_float_nan((T)(unsigned short)(both >> 16), bad, hasBad);
    } else {
        first = as_float_nan((T)(unsigned char)(both & 0xffu), bad, hasBad);
        second = as_float_nan((T)(unsigned char)((both >> 8) & 0xffu), bad, hasBad);
    }
}
template <typename T>
__device__ __forceinline__ float lds_one(const char* buf, uint32_t byteOff, float bad, bool hasBad)
{
    return as_float_nan(*reinterpret_cast<const T*>(buf + byteOff), bad, hasBad);
}
// the two elements as they are stored, converted but not yet compared with the fill value (the interior form tests all four
// stencil values at once: any fill value among them makes the result undefined, whatever its weight -- 0 * NaN is NaN)
template <typename T>
__device__ __forceinline__ void lds_pair2_raw(const char* buf, uint32_t alignedOff, uint32_t shift, float& first, float& second)
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(buf + alignedOff);
    const uint32_t both = __builtin_amdgcn_alignbit(w[1], w[0], shift);
    if constexpr (sizeof(T) == 2) {
        first = (float)(T)(unsigned short)(both & 0xffffu);
        second = (float)(T)(unsigned short)(both >> 16);
    } else {
        first = (float)(T)(unsigned char)(both & 0xffu);
        second = (float)(T)(unsigned char)((both >> 8) & 0xffu);
    }
}
// interpolationArray2Data for results of THIS kernel: NaN -> fill, else MetNoFimex::round (lround) and the reference's casts
// long -> int -> T (typed_convert.hpp: from_float_fill).  The results here are stored elements or convex combinations of four
// of them, so |v| < 2^17: the branch of from_float_fill for values beyond the int range cannot be taken and is left out, the
// rest is the same arithmetic without branches (the fraction v - trunc(v) is exact in float).
template <typename T>
__device__ __forceinline__ uint32_t round_bits(float v, T fill)
{
    const float t = truncf(v);
    const float r = t + ((fabsf(v - t) >= 0.5f) ? copysignf(1.f, v) : 0.f);
    const int i = (v != v) ? (int)fill : (int)r;
    return (uint32_t)i;
}
// results r0 (cell c) and r1 (cell c + 1) of one pair; offsets in BYTES of the typed slice, ~0u = not mine
constexpr int kTypedStoreAux = 2;  // non-temporal (written through as well -- sc1 nt, the float kernel's policy -- these 4-byte-per-lane stores of half as many bytes lose 9 %: 1.48 against 1.35 ms)
template <typename T, bool PAIR>
__device__ __forceinline__ void store_pair(rsrc_t ro, uint32_t off0, uint32_t off1, float r0, float r1, T fill)
{
    constexpr uint32_t kMask = sizeof(T) == 2 ? 0xffffu : 0xffu;
    const uint32_t b0 = round_bits<T>(r0, fill) & kMask, b1 = round_bits<T>(r1, fill) & kMask;
    if constexpr (PAIR) {  // both cells exist or neither (even row length, even tile widths)
        if constexpr (sizeof(T) == 2) __builtin_amdgcn_raw_buffer_store_b32(b0 | (b1 << 16), ro, off0, 0, kTypedStoreAux);
        else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(b0 | (b1 << 8)), ro, off0, 0, 2);
        (void)off1;
    } else if constexpr (sizeof(T) == 2) {
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)b0, ro, off0, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)b1, ro, off1, 0, 2);
    } else {
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b0, ro, off0, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b1, ro, off1, 0, 2);
    }
}

// STENCIL 1 (nearest) or 2 (bilinear); NT threads, 4 outputs per lane (two pairs); KMAX 16-byte chunks per lane and slice
// PAIR: the row length and the slice start allow aligned stores of two results
template <int STENCIL, int NT, int KMAX, typename T, bool PAIR, int DEPTH = 2>
__global__ void __launch_bounds__(NT) staged_apply2_typed(Staged2Args a, TypedEdge te)
{
    static_assert(STENCIL == 1 || STENCIL == 2, "stored types: nearest and bilinear");
    constexpr uint32_t EB = sizeof(T);
    constexpr int PER = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint32_t slot0 = blockIdx.x, zc = blockIdx.y;
    if (a.nZChunks != 0) {
        const uint32_t k = blockIdx.x / kXcds;
        zc = k % a.nZChunks;
        slot0 = (k / a.nZChunks) * kXcds + blockIdx.x % kXcds;
    }
    const uint32_t tile = a.order[slot0];
    if (tile == 0xFFFFFFFFu) return;
    const StagedTile T_ = a.tiles[tile];
    const uint32_t z0 = a.zStart[zc], z1 = a.zStart[zc + 1];
    const uint32_t nzl = z1 - z0;
    const bool hasBad = te.hasBad != 0;
    const T fillT = static_cast<T>(te.fillOut);  // ScaleValue's newFill_ (include/fimex/Utils.h:456)
    const uint32_t outBytes = a.nOut * EB;
    const char* inBase = reinterpret_cast<const char*>(a.in);
    char* outBase = reinterpret_cast<char*>(a.out);
    const uint32_t outRecords = (kTuningBuild && (a.flags & 2)) ? 0u : outBytes;
    const uint32_t inRecords = (kTuningBuild && (a.flags & 1)) ? 0u : a.inBytes;

    // staging list: chunk c = threadIdx.x + j * NT of the tile (byte offset of its 16 bytes inside a source slice)
    const uint32_t waveChunk = (threadIdx.x / kWave) * kWave;
    uint32_t gOff[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        const uint32_t c = threadIdx.x + j * NT;
        gOff[j] = (c < T_.nChunks) ? a.chunkOff[T_.chunkBase + c] * EB : 0xFFFFFFFFu;
    }
    const uint32_t un = (T_.nChunks + NT - 1) / NT;
    const uint32_t slotFloats = a.slotChunks * 4u;
    float* const spare = smem + DEPTH * slotFloats;
    auto dma_dst = [&](uint32_t sl, int j) {
        const uint32_t c = waveChunk + (uint32_t)j * NT;
        return c < a.slotChunks ? smem + sl * slotFloats + c * 4u : spare;
    };
    for (uint32_t i = 0; i + 1 < (uint32_t)DEPTH && i < nzl; ++i) {
        const rsrc_t rs = make_rsrc(inBase + (size_t)(z0 + i) * a.inBytes, inRecords);
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if ((uint32_t)j < un) dma16(rs, dma_dst(i, j), gOff[j]);
    }
    // per-lane plan: output q = 2 * p + h is cell 2 * threadIdx.x + h + p * 2 * NT of the tile (row-major over the tile's width)
    uint32_t cellOff[PER];       // byte offset inside a typed output slice, ~0u = not mine
    uint32_t row[PER][STENCIL];  // byte offsets of the stencil rows in the staged image
    uint32_t cellIdx[PER];
    float xf[PER], yf[PER];
    bool undef[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const uint32_t e = 2u * threadIdx.x + (uint32_t)(q & 1) + (uint32_t)(q >> 1) * 2u * NT;
        const uint32_t ly = e / T_.w, lx = e - ly * T_.w;
        const uint32_t y = T_.y0 + ly;
        cellOff[q] = 0xFFFFFFFFu;
        cellIdx[q] = 0xFFFFFFFFu;
        uint32_t pa = kInvalidPos;
        xf[q] = yf[q] = 0.f;
        if (ly < a.tileH && y < a.outY) {
            const uint32_t cell = y * a.outX + T_.x0 + lx;
            cellIdx[q] = cell;
            cellOff[q] = cell * EB;
            pa = a.ldsA[cell];
            if (STENCIL == 2) { xf[q] = a.xf[cell]; yf[q] = a.yf[cell]; }
        }
        undef[q] = pa == kInvalidPos;
        row[q][0] = undef[q] ? 0u : (pa & 0xFFFFu) * EB;
        if (STENCIL == 2) row[q][STENCIL - 1] = undef[q] ? 0u : (pa >> 16) * EB;
    }
    if (T_.rsv[0] != 0) {
        // gather tile (see staged_apply2): stencils straight from memory, element by element
        uint32_t p[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            p[q] = (cellIdx[q] != 0xFFFFFFFFu) ? a.pos[cellIdx[q]] : kInvalidPos;
            undef[q] = p[q] == kInvalidPos;
            if (undef[q]) p[q] = 0;
        }
        auto uniform = [](const char* ptr) {
            const uint64_t v = reinterpret_cast<uint64_t>(ptr);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
        };
        for (uint32_t z = z0; z < z1; ++z) {
            const rsrc_t rs = make_rsrc(uniform(inBase + (size_t)z * a.inBytes), inRecords);
            const rsrc_t ro = make_rsrc(uniform(outBase + (size_t)z * outBytes), outRecords);
            auto ld = [&](uint32_t cell) {
                if constexpr (EB == 2) return as_float_nan((T)__builtin_amdgcn_raw_buffer_load_b16(rs, cell * 2u, 0, 0), te.bad, hasBad);
                else return as_float_nan((T)__builtin_amdgcn_raw_buffer_load_b8(rs, cell, 0, 0), te.bad, hasBad);
            };
            float r[PER];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                if constexpr (STENCIL == 1) {
                    r[q] = ld(p[q]);
                } else {
                    const bool nnx = (__float_as_uint(xf[q]) >> 31) != 0, nny = (__float_as_uint(yf[q]) >> 31) != 0;
                    const uint32_t dx = nnx ? 0u : 1u, dy = nny ? 0u : a.inX;
                    const float s00 = ld(p[q]), s01 = ld(p[q] + dx), s10 = ld(p[q] + dy), s11 = ld(p[q] + dx + dy);
                    const float top = (1.f - xf[q]) * s00 + xf[q] * s01;
                    const float bot = (1.f - xf[q]) * s10 + xf[q] * s11;
                    const float inter = (1.f - yf[q]) * top + yf[q] * bot;
                    const float liny = (1 - yf[q]) * s00 + (yf[q] * s10);
                    r[q] = nnx ? (nny ? s00 : liny) : (nny ? top : inter);
                }
                if (undef[q]) r[q] = undefined_f();
            }
            store_pair<T, PAIR>(ro, cellOff[0], cellOff[1], r[0], r[1], fillT);
            store_pair<T, PAIR>(ro, cellOff[2], cellOff[3], r[2], r[3], fillT);
        }
        return;
    }
    if (T_.nChunks == 0) {  // every output of the tile is undefined
        for (uint32_t z = z0; z < z1; ++z) {
            const rsrc_t ro = make_rsrc(outBase + (size_t)z * outBytes, outRecords);
            store_pair<T, PAIR>(ro, cellOff[0], cellOff[1], undefined_f(), undefined_f(), fillT);
            store_pair<T, PAIR>(ro, cellOff[2], cellOff[3], undefined_f(), undefined_f(), fillT);
        }
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    constexpr int NST = PAIR ? 2 : 4;  // store instructions a lane issues per slice
    bool plainWave = true;
#pragma unroll
    for (int q = 0; q < PER; ++q)
        plainWave = plainWave && !undef[q] && (STENCIL != 2 || ((__float_as_uint(xf[q]) | __float_as_uint(yf[q])) >> 31) == 0);
    plainWave = __all(plainWave) != 0;
    uint32_t mNnx[PER], mNny[PER], mUndef[PER];
    uint32_t rowA[PER][STENCIL], rowS[PER][STENCIL];  // bilinear: aligned byte offset of a stencil row's pair, and its shift operand
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        mNnx[q] = (uint32_t)((int32_t)__float_as_uint(xf[q]) >> 31);
        mNny[q] = (uint32_t)((int32_t)__float_as_uint(yf[q]) >> 31);
        mUndef[q] = undef[q] ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int i = 0; i < STENCIL; ++i) { rowA[q][i] = row[q][i] & ~3u; rowS[q][i] = row[q][i] << 3; }
    }
    // no fill value: comparisons with NaN never hold, the loops need no separate test
    const float badCmp = hasBad ? te.bad : undefined_f();
    auto pick = [](uint32_t mask, float x, float y) { return __uint_as_float((__float_as_uint(x) & mask) | (__float_as_uint(y) & ~mask)); };
    auto run = [&](auto unTag, auto plainTag) __attribute__((always_inline)) {
        constexpr int UN = decltype(unTag)::value;
        constexpr bool PLAIN = decltype(plainTag)::value;
        uint32_t slot = 0;
        for (uint32_t i = 0; i < nzl; ++i) {
            const uint32_t z = z0 + i;
            const bool more = i + DEPTH - 1 < nzl;
            if (more) {
                const uint32_t sl = (slot + DEPTH - 1 >= (uint32_t)DEPTH) ? slot - 1 : slot + DEPTH - 1;
                const rsrc_t rs = make_rsrc(inBase + (size_t)(z + DEPTH - 1) * a.inBytes, inRecords);
#pragma unroll
                for (int j = 0; j < UN; ++j) dma16(rs, dma_dst(sl, j), gOff[j]);
            }
            const char* curb = reinterpret_cast<const char*>(smem + slot * slotFloats);
            const rsrc_t ro = make_rsrc(outBase + (size_t)z * outBytes, outRecords);
            float r[PER];
            if constexpr (STENCIL == 1) {
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const float v = lds_one<T>(curb, row[q][0], te.bad, hasBad);
                    r[q] = PLAIN ? v : pick(mUndef[q], undefined_f(), v);
                }
            } else {
                float s00[PER], s01[PER], s10[PER], s11[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    if constexpr (PLAIN) {
                        lds_pair2_raw<T>(curb, rowA[q][0], rowS[q][0], s00[q], s01[q]);
                        lds_pair2_raw<T>(curb, rowA[q][STENCIL - 1], rowS[q][STENCIL - 1], s10[q], s11[q]);
                    } else {
                        lds_pair2<T>(curb, rowA[q][0], rowS[q][0], te.bad, hasBad, s00[q], s01[q]);
                        lds_pair2<T>(curb, rowA[q][STENCIL - 1], rowS[q][STENCIL - 1], te.bad, hasBad, s10[q], s11[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const float top = (1.f - xf[q]) * s00[q] + xf[q] * s01[q];
                    const float bot = (1.f - xf[q]) * s10[q] + xf[q] * s11[q];
                    const float inter = (1.f - yf[q]) * top + yf[q] * bot;
                    r[q] = inter;
                    if constexpr (PLAIN) {  // interior cell: undefined iff one of the four is the fill value (mifi_bad2nanf, then NaN spreads)
                        // (the stored elements are integers, exact in float, and so is a fill value that can occur among them: the
                        // product of the four differences is zero iff one element is the fill value -- one comparison and one
                        // selection per output instead of four of each; without a fill value the product is NaN and never zero)
                        const float anyBad = ((s00[q] - badCmp) * (s01[q] - badCmp)) * ((s10[q] - badCmp) * (s11[q] - badCmp));
                        r[q] = (anyBad == 0.f) ? undefined_f() : inter;
                    }
                    if constexpr (!PLAIN) {
                        const float liny = (1 - yf[q]) * s00[q] + (yf[q] * s10[q]);
                        r[q] = pick(mNnx[q], pick(mNny[q], s00[q], liny), pick(mNny[q], top, inter));
                        r[q] = pick(mUndef[q], undefined_f(), r[q]);
                    }
                }
            }
            store_pair<T, PAIR>(ro, cellOff[0], cellOff[1], r[0], r[1], fillT);
            store_pair<T, PAIR>(ro, cellOff[2], cellOff[3], r[2], r[3], fillT);
            if (more) wait_vmcnt<(DEPTH - 2) * UN + (DEPTH - 1) * NST>();
            else wait_vmcnt<NST>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            slot = (slot + 1 == (uint32_t)DEPTH) ? 0 : slot + 1;
        }
    };
    auto run_un = [&](auto unTag) __attribute__((always_inline)) {
        if (plainWave) run(unTag, std::true_type());
        else run(unTag, std::false_type());
    };
    switch (un) {
    case 1: run_un(std::integral_constant<int, 1>()); break;
    case 2: run_un(std::integral_constant<int, KMAX >= 2 ? 2 : 1>()); break;
    case 3: run_un(std::integral_constant<int, KMAX >= 3 ? 3 : 1>()); break;
    case 4: run_un(std::integral_constant<int, KMAX >= 4 ? 4 : 1>()); break;
    case 5: run_un(std::integral_constant<int, KMAX >= 5 ? 5 : 1>()); break;
    default: run_un(std::integral_constant<int, KMAX >= 6 ? 6 : 1>()); break;
    }
}

constexpr uint32_t kSpareBytes = 1024;  // one wave instruction of LDS-DMA behind the ring (see staged_apply2)

// chunks of one slot: the workgroup's LDS less the spare KiB, in `depth` equal slots of whole wave instructions
inline uint32_t slot_chunks(uint32_t ldsBytes, uint32_t depth)
{
    return ((ldsBytes - kSpareBytes) / depth / 16u) & ~63u;
}

struct Shape2 {
    int nt, per, kmax;
    uint32_t depth;
    uint32_t tileW, tileH;  // widest tile
    uint32_t ldsBytes;
};

template <int STENCIL, int NT, int PER, int KMAX, bool FAST = false>
void launch_one(const Staged2Args& a, dim3 grid, size_t ldsBytes, uint32_t depth, hipStream_t stream)
{
    if (depth == 3) {
        allow_dynamic_lds(reinterpret_cast<const void*>(&staged_apply2<STENCIL, NT, PER, KMAX, FAST, 3>), ldsBytes);
        staged_apply2<STENCIL, NT, PER, KMAX, FAST, 3><<<grid, NT, ldsBytes, stream>>>(a);
    } else {
        allow_dynamic_lds(reinterpret_cast<const void*>(&staged_apply2<STENCIL, NT, PER, KMAX, FAST, 2>), ldsBytes);
        staged_apply2<STENCIL, NT, PER, KMAX, FAST, 2><<<grid, NT, ldsBytes, stream>>>(a);
    }
}

template <int STENCIL, bool FAST = false>
void launch_shape(const Staged2Plan& s, const Staged2Args& a, dim3 grid, hipStream_t stream)
{
    const uint32_t key = s.nt * 10000 + s.per * 100 + s.kmax;
    switch (key) {
    case 2560406: launch_one<STENCIL, 256, 4, 6, FAST>(a, grid, s.ldsBytes, s.depth, stream); break;
    case 5120406: launch_one<STENCIL, 512, 4, 6, FAST>(a, grid, s.ldsBytes, s.depth, stream); break;
    case 10240405: launch_one<STENCIL, 1024, 4, 5, FAST>(a, grid, s.ldsBytes, s.depth, stream); break;
    case 5120203: launch_one<STENCIL, 512, 2, 3, FAST>(a, grid, s.ldsBytes, s.depth, stream); break;
    case 10240203: launch_one<STENCIL, 1024, 2, 3, FAST>(a, grid, s.ldsBytes, s.depth, stream); break;
    case 2560204: launch_one<STENCIL, 256, 2, 4, FAST>(a, grid, s.ldsBytes, s.depth, stream); break;
    default: throw Error("staged2: unexpected workgroup shape");
    }
}

template <int STENCIL>
bool build_shape(const fimex_amd_regrid_plan& plan, Staged2Plan& s, const NeedSource& need, hipStream_t stream, const Shape2& sh, uint32_t stripe,
                 uint32_t cpc = 4)
{
    const uint32_t outX = (uint32_t)plan.outX, outY = (uint32_t)plan.outY;
    const uint32_t tileH = sh.tileH;
    const uint32_t nBands = (uint32_t)ceil_div(outY, tileH);
    // the ring holds `depth` slots, each large enough for any tile (chunks rounded up to whole wave instructions)
    uint32_t cap = std::min<uint32_t>(slot_chunks(sh.ldsBytes, sh.depth), (uint32_t)sh.kmax * sh.nt);
    cap = std::min<uint32_t>(cap, 65535u / cpc);  // 16-bit LDS offsets, in elements
    const uint32_t step = sh.tileW >= 128 ? 64u : 32u;  // tile widths are multiples of this (a wave stores 64 consecutive cells)
    // Tiles: every tile row starts as tiles of the widest shape.  A tile that does not fit (too many chunks for a slot, too
    // many source rows) makes its row narrower when most tiles of the row fail (the row's cells cover more source: rows near
    // the pole of the benchmark plan), otherwise it is split in two; at the narrowest width it becomes a gather tile
    // (rsv[0] = 1: the kernel reads its stencils from memory).  More than 1/8 of the cells that way: no staged plan.
    const uint32_t widest = std::min(sh.tileW, (outX + step - 1) / step * step);
    std::vector<StagedTile> tiles;
    std::vector<uint32_t> bandOf;
    auto uniform_row = [&](uint32_t b, uint32_t w, std::vector<StagedTile>& out) {
        for (uint32_t x0 = 0; x0 < outX; x0 += w) {
            StagedTile t{};
            t.x0 = x0;
            t.y0 = b * tileH;
            t.w = std::min(w, outX - x0);
            out.push_back(t);
        }
    };
    std::vector<std::vector<StagedTile>> rows(nBands);
    std::vector<uint32_t> rowW(nBands, widest);
    for (uint32_t b = 0; b < nBands; ++b) uniform_row(b, widest, rows[b]);
    DeviceArray<StagedTile> dTiles;
    for (int pass = 0;; ++pass) {
        tiles.clear();
        bandOf.clear();
        for (uint32_t b = 0; b < nBands; ++b)
            for (const StagedTile& t : rows[b]) { tiles.push_back(t); bandOf.push_back(b); }
        if (tiles.size() > 0x7FFFFFFFu / 8) return false;
        dTiles.allocate(tiles.size());
        FA_HIP(hipMemcpyAsync(dTiles.get(), tiles.data(), tiles.size() * sizeof(StagedTile), hipMemcpyHostToDevice, stream));
        tile_scan<STENCIL><<<(uint32_t)tiles.size(), kBlock, 0, stream>>>(need, (int64_t)plan.inX, (int64_t)plan.inY, outX, outY, tileH,
                                                                       dTiles.get(), cap, 0, nullptr, nullptr, nullptr, cpc);
        FA_HIP(hipGetLastError());
        FA_HIP(hipMemcpyAsync(tiles.data(), dTiles.get(), tiles.size() * sizeof(StagedTile), hipMemcpyDeviceToHost, stream));
        FA_HIP(hipStreamSynchronize(stream));
        bool again = false;
        size_t i = 0;
        for (uint32_t b = 0; b < nBands; ++b) {
            const size_t n = rows[b].size();
            size_t failed = 0;
            for (size_t k = 0; k < n; ++k) {
                rows[b][k] = tiles[i + k];
                if (tiles[i + k].rsv[0] == 0 && tiles[i + k].nChunks == 0xFFFFFFFFu) ++failed;
            }
            i += n;
            if (failed == 0) continue;
            again = true;
            if (2 * failed > n && rowW[b] > step) {  // the whole row is too heavy: narrower tiles throughout
                rowW[b] -= step;
                rows[b].clear();
                uniform_row(b, rowW[b], rows[b]);
                continue;
            }
            std::vector<StagedTile> next;
            for (const StagedTile& t : rows[b]) {
                if (t.rsv[0] != 0 || t.nChunks != 0xFFFFFFFFu) { next.push_back(t); continue; }
                if (t.w <= step) {  // cannot be split any further
                    StagedTile g = t;
                    g.nChunks = 0;
                    g.rsv[0] = 1;
                    next.push_back(g);
                    continue;
                }
                StagedTile l = t, r = t;
                l.w = (t.w / 2 + step - 1) / step * step;
                r.x0 = t.x0 + l.w;
                r.w = t.w - l.w;
                l.nChunks = r.nChunks = 0;
                next.push_back(l);
                next.push_back(r);
            }
            rows[b].swap(next);
        }
        if (!again) break;
        if (pass > 64) return false;
    }
    size_t gatherCells = 0, liveCells = 0;
    for (const StagedTile& t : tiles) {
        if (t.rsv[0] != 0) gatherCells += t.w;
        if (t.rsv[0] != 0 || t.nChunks != 0) liveCells += t.w;
    }
    if (gatherCells * 8 > liveCells) return false;  // positions without spatial coherence: the gather kernels serve them better
    size_t total = 0;
    for (auto& t : tiles) {
        FA_REQUIRE(total <= 0xFFFFFFFFu, "staged plan: too many chunks");
        t.chunkBase = (uint32_t)total;
        total += t.nChunks;
    }
    if (total > 0xFFFFFFFFull) return false;
    FA_HIP(hipMemcpyAsync(dTiles.get(), tiles.data(), tiles.size() * sizeof(StagedTile), hipMemcpyHostToDevice, stream));
    const size_t n = plan.outX * plan.outY;
    s.chunkOff.allocate(std::max<size_t>(total, 1));
    s.ldsA.allocate(n);
    s.ldsB.allocate(STENCIL == 4 ? n : 0);
    tile_scan<STENCIL><<<(uint32_t)tiles.size(), kBlock, 0, stream>>>(need, (int64_t)plan.inX, (int64_t)plan.inY, outX, outY, tileH,
                                                                   dTiles.get(), cap, 1, s.chunkOff.get(), s.ldsA.get(), s.ldsB.get(), cpc);
    FA_HIP(hipGetLastError());
    // workgroup -> tile: workgroups are dealt round-robin over the XCDs (b % 8 shares an L2); tile rows go to the XCDs in
    // stripes of `stripe` rows, so that neighbours in x (and, inside a stripe, in y) run on the same XCD and meet in its L2
    // (stripes of about `stripe` rows, their number a multiple of the XCD count so that every XCD gets equally many)
    std::vector<std::vector<uint32_t>> perXcd(kXcds);
    uint32_t nStripes = (uint32_t)((nBands + stripe * kXcds / 2) / (stripe * kXcds)) * kXcds;
    if (nStripes < (uint32_t)kXcds) nStripes = kXcds;
    if (nStripes > nBands) nStripes = std::max<uint32_t>(nBands / kXcds * kXcds, 1);
    // (fewer tile rows than XCDs -- wide, short targets such as cross-sections: the tiles themselves are dealt round-robin)
    for (size_t i = 0; i < tiles.size(); ++i)
        perXcd[nBands < (uint32_t)kXcds ? i % kXcds : ((uint64_t)bandOf[i] * nStripes / nBands) % kXcds].push_back((uint32_t)i);
    size_t longest = 0;
    for (auto& l : perXcd) longest = std::max(longest, l.size());
    std::vector<uint32_t> order(longest * kXcds, 0xFFFFFFFFu);
    for (int x = 0; x < kXcds; ++x)
        for (size_t k = 0; k < perXcd[x].size(); ++k) order[k * kXcds + x] = perXcd[x][k];
    s.order.allocate(order.size());
    FA_HIP(hipMemcpyAsync(s.order.get(), order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    FA_HIP(hipStreamSynchronize(stream));
    s.tiles = std::move(dTiles);
    s.nt = (uint32_t)sh.nt;
    s.per = (uint32_t)sh.per;
    s.kmax = (uint32_t)sh.kmax;
    s.tileH = tileH;
    s.tileWMax = sh.tileW;
    s.nTiles = (uint32_t)tiles.size();
    s.gridX = (uint32_t)order.size();
    s.ldsBytes = sh.ldsBytes;
    s.depth = sh.depth;
    s.totalChunks = total;
    s.stagedCells = total * cpc;
    s.valid = true;
    return true;
}

}  // namespace

// Workgroup shape by stencil (tuning: STAGE2_NT / STAGE2_TW / STAGE2_LDS / STAGE2_STRIPE); false: no staged form for this
// plan (positions without spatial coherence), the caller keeps the gather kernels.
namespace {
bool build_staged2_shape(fimex_amd_regrid_plan& plan, Staged2Plan& target, int ntWanted, const double* d_px, const double* d_py, hipStream_t stream)
{
    if (plan.outX * plan.outY == 0) return false;
    // the float form of the bicubic stencil is as light as the bilinear one: it takes the bilinear shapes
    const bool cubic = plan.kind == PlanKind::Bicubic && !plan.bicubicFast;
    // measured on the benchmark plan (round 2's sweeps, profiles/LAB_NOTES_r01_r02.md): 1024 threads on 512 x 8 tiles (one workgroup per CU) for the
    // 1 x 1 and 2 x 2 stencils.  With the tile-major launch order 512 threads on 256 x 8 tiles (two workgroups per CU) run the
    // bilinear launch 2.5-3.5 % faster on two boxes (2.19 against 2.27 ms) and 4 % slower on two others (2.41 against 2.31 ms), and
    // lose on the 1 x 1 stencil and on short batches everywhere: the shape that behaves the same on every box is kept.  The 4 x 4
    // stencil in float arithmetic takes 256 x 8 tiles on 512 threads (its halo makes taller or wider tiles stage more), in the
    // reference's arithmetic it is FP64-bound and prefers 128 x 8 tiles on 512 threads.
    const int nt = ntWanted > 0 ? ntWanted : tuning("STAGE2_NT", (cubic || plan.bicubicFast) ? 512 : 1024);
    if (!(nt == 256 || nt == 512 || nt == 1024)) return false;
    Shape2 sh{};
    sh.nt = nt;
    sh.per = cubic ? 2 : 4;
    // chunks per lane: 1024 threads hold one slot of at most 80 KB
    sh.kmax = cubic ? (nt == 256 ? 4 : 3) : (nt == 1024 ? 5 : 6);
    const uint32_t outputs = (uint32_t)(sh.nt * sh.per);
    sh.tileW = (uint32_t)tuning("STAGE2_TW", cubic ? 64 * (nt / 256) : nt / 2);
    if (sh.tileW < 32 || sh.tileW % 32 != 0 || outputs % sh.tileW != 0) return false;
    sh.tileH = outputs / sh.tileW;
    // LDS of one workgroup: 3 / 2 / 1 workgroups per CU (160 KB)
    const int ldsDefault = nt == 256 ? 52 : (nt == 512 ? 79 : 159);
    sh.ldsBytes = (uint32_t)tuning("STAGE2_LDS_KB", ldsDefault) * 1024u;
    if (sh.ldsBytes > 160u * 1024u - 64u) sh.ldsBytes = 160u * 1024u - 64u;
    if (sh.ldsBytes < 16u * 1024u) return false;
    sh.depth = tuning("STAGE2_DEPTH", 2) == 3 ? 3u : 2u;  // slices of the ring: one or two in flight while one is interpolated
    // tile rows go to the XCDs one by one (row r to XCD r % 8): with the tile-major launch order the eight XCDs then work on eight
    // neighbouring tile rows at any time.  (With the chunk-major order stripes of 8 rows per XCD fetched 10.2 instead of 11.1 GB for
    // the bilinear launch at the same time; with the tile-major order stripes of 2, 4 or 8 rows lose 3-6 %.)
    const uint32_t stripe = (uint32_t)std::max(1, tuning("STAGE2_STRIPE", 1));
    NeedSource need;
    need.px = d_px;
    need.py = d_py;
    switch (plan.kind) {
    case PlanKind::Nearest: return build_shape<1>(plan, target, need, stream, sh, stripe);
    case PlanKind::Bilinear: return build_shape<2>(plan, target, need, stream, sh, stripe);
    case PlanKind::Bicubic: return build_shape<4>(plan, target, need, stream, sh, stripe);
    default: return false;
    }
}
}  // namespace

bool build_staged2_plan(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream)
{
    if (!build_staged2_shape(plan, plan.staged2, 0, d_px, d_py, stream)) return false;
    // the bilinear plan also holds the 512-thread shape (two workgroups per CU): faster on some devices for long batches,
    // slower on others and for short ones -- fimex_amd_regrid_plan_tune_device decides on the spot, the default stays
    if (plan.kind == PlanKind::Bilinear && plan.staged2.nt == 1024 && tuning("STAGE2_ALT", 1) != 0)
        build_staged2_shape(plan, plan.staged2Alt, 512, d_px, d_py, stream);
    // likewise the 4 x 4 stencil in float arithmetic: 256 threads on 128 x 8 tiles (three workgroups per CU) beside 512 threads on
    // 256 x 8 (2.38 against 2.41 ms in one process, round 2, profiles/LAB_NOTES_r01_r02.md)
    if (plan.kind == PlanKind::Bicubic && plan.bicubicFast && plan.staged2.nt == 512 && tuning("STAGE2_ALT", 1) != 0)
        build_staged2_shape(plan, plan.staged2Alt, 256, d_px, d_py, stream);
    return true;
}


void launch_staged2_apply(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream)
{
    const int forced = tuning("STAGE2_USE_ALT", -1);  // tuning build: 0 / 1 overrides the plan's choice
    const bool alt = plan.staged2Alt.valid && (forced >= 0 ? forced == 1 : plan.useAlt != 0);
    const Staged2Plan& s = alt ? plan.staged2Alt : plan.staged2;
    Staged2Args a{};
    a.in = d_in;
    a.out = d_out;
    a.tiles = s.tiles.get();
    a.order = s.order.get();
    a.chunkOff = s.chunkOff.get();
    a.ldsA = s.ldsA.get();
    a.ldsB = s.ldsB.get();
    a.pos = plan.pos.get();
    a.inX = (uint32_t)plan.inX;
    a.xf = plan.xf.get();
    a.yf = plan.yf.get();
    a.xfd = plan.xfd.get();
    a.yfd = plan.yfd.get();
    a.outX = (uint32_t)plan.outX;
    a.outY = (uint32_t)plan.outY;
    a.tileH = s.tileH;
    a.inBytes = (uint32_t)(plan.inX * plan.inY * 4);
    a.nOut = (uint32_t)(plan.outX * plan.outY);
    a.nz = (uint32_t)nz;
    // z chunks.  Tile-major order (the default, STAGE2_ORDER 1): the chunks of a tile are consecutive workgroups of one XCD, so
    // they start together and fetch the tile's per-output plan -- 12 bytes per cell, 0.05 GB per chunk of the benchmark
    // launch -- once from memory instead of once per chunk, and the chip as a whole works on eight neighbouring tile rows;
    // chunks of one size (about STAGE2_ZPB slices).  Measured on the benchmark plan (round 2's sweeps, profiles/LAB_NOTES_r01_r02.md): bilinear
    // 2.40 -> 2.32 ms, nearest 2.32 -> 2.21 ms, 25 slices 0.322 -> 0.303 ms.
    // Chunk-major order (STAGE2_ORDER 0, round 1 and early round 2): all tiles of chunk 0, then chunk 1, ...; the chunks shrink
    // towards the end of the launch so that the last workgroups are short ones (STAGE2_ZTAIL).
    const bool tileMajor = tuning("STAGE2_ORDER", 1) == 1;
    uint32_t zpb = (uint32_t)tuning("STAGE2_ZPB", tileMajor ? 25 : 50);
    const uint32_t ztail = tileMajor ? 0u : (uint32_t)tuning("STAGE2_ZTAIL", 10);  // 0: chunks of one size
    if (zpb < 1) zpb = 1;
    const size_t mostChunks = tileMajor ? 16 : (size_t)kMaxZChunks - 6;
    if (ceil_div(nz, (size_t)zpb) > mostChunks) zpb = (uint32_t)ceil_div(nz, mostChunks);
    uint32_t nChunks = 0;
    if (tileMajor) {
        const uint32_t n = (uint32_t)ceil_div(nz, (size_t)zpb);
        for (uint32_t c = 0, z = 0; c < n; ++c) {
            a.zStart[nChunks++] = z;
            z += (uint32_t)nz / n + (c < (uint32_t)nz % n ? 1u : 0u);
        }
    }
    for (uint32_t z = 0; !tileMajor && z < nz;) {
        const uint32_t rem = (uint32_t)nz - z;
        uint32_t size = zpb;
        if (ztail > 0 && rem <= 2 * zpb) size = std::max(ztail, (rem + 1) / 2);
        if (size > rem || rem - size < (ztail + 1) / 2) size = rem;
        FA_REQUIRE(nChunks < (uint32_t)kMaxZChunks, "too many z chunks for one launch");
        a.zStart[nChunks++] = z;
        z += size;
    }
    a.zStart[nChunks] = (uint32_t)nz;
    a.slotChunks = slot_chunks(s.ldsBytes, s.depth);
    a.flags = (uint32_t)tuning("STAGE2_ABLATE", 0) | ((uint32_t)tuning("STAGE2_STORE", 0) << 3);  // STORE 1 plain, 2 nt, 4 sc0 nt (default: sc1 nt)
    a.nZChunks = tileMajor ? nChunks : 0u;
    const dim3 grid(tileMajor ? s.gridX * nChunks : s.gridX, tileMajor ? 1u : nChunks, 1);
    switch (plan.kind) {
    case PlanKind::Nearest: launch_shape<1>(s, a, grid, stream); break;
    case PlanKind::Bilinear: launch_shape<2>(s, a, grid, stream); break;
    default:
        if (plan.bicubicFast) launch_shape<4, true>(s, a, grid, stream);
        else launch_shape<4>(s, a, grid, stream);
        break;
    }
    FA_HIP(hipGetLastError());
}

// ---- stored types
namespace {

// the plan's staged form for slices of elemBytes-byte elements (2 or 1), built on first use; nullptr: none (the caller takes
// the first staged form or the gather kernels)
const Staged2Plan* staged2_typed_form(const fimex_amd_regrid_plan& plan, uint32_t elemBytes, hipStream_t stream)
{
    if (plan.kind != PlanKind::Nearest && plan.kind != PlanKind::Bilinear) return nullptr;
    if ((plan.inX * plan.inY * elemBytes) % 4 != 0) return nullptr;  // slices start on 4-byte boundaries (LDS-DMA)
    const int idx = elemBytes == 2 ? 0 : 1;
    std::lock_guard<std::mutex> lock(plan.typed2.mtx);
    Staged2Plan& form = plan.typed2.form[idx];
    if (!plan.typed2.tried[idx]) {
        plan.typed2.tried[idx] = true;
        // 512 threads on 256 x 8 tiles, two workgroups per CU: the kernel converts every element it touches and is bound by its
        // instructions as much as by memory, so occupancy counts for more than tile size here
        Shape2 sh{};
        sh.nt = tuning("STAGE2T_NT", 512);
        if (!(sh.nt == 256 || sh.nt == 512 || sh.nt == 1024)) return nullptr;
        sh.per = 4;
        sh.kmax = sh.nt == 1024 ? 5 : 6;
        sh.tileW = (uint32_t)tuning("STAGE2T_TW", sh.nt / 2);
        const uint32_t outputs = (uint32_t)sh.nt * 4u;
        if (sh.tileW < 64 || sh.tileW % 64 != 0 || outputs % sh.tileW != 0) return nullptr;
        sh.tileH = outputs / sh.tileW;
        sh.ldsBytes = (uint32_t)tuning("STAGE2T_LDS_KB", sh.nt == 256 ? 39 : (sh.nt == 512 ? 79 : 159)) * 1024u;
        if (sh.ldsBytes > 160u * 1024u - 64u) sh.ldsBytes = 160u * 1024u - 64u;
        if (sh.ldsBytes < 16u * 1024u) return nullptr;
        sh.depth = (sh.nt == 512 && tuning("STAGE2T_DEPTH", 2) == 3) ? 3u : 2u;
        NeedSource need;
        need.pos = plan.pos.get();
        need.xf = plan.xf.get();
        need.yf = plan.yf.get();
        const uint32_t cpc = 16u / elemBytes;
        try {
            if (plan.kind == PlanKind::Nearest) build_shape<1>(plan, form, need, stream, sh, 1, cpc);
            else build_shape<2>(plan, form, need, stream, sh, 1, cpc);
        } catch (...) {
            form.valid = false;
            throw;
        }
    }
    return form.valid ? &form : nullptr;
}

template <int STENCIL, typename T>
void launch_typed_shape(const Staged2Plan& s, const Staged2Args& a, const TypedEdge& te, dim3 grid, hipStream_t stream)
{
    auto go = [&](auto kernel, int nt) {
        allow_dynamic_lds(reinterpret_cast<const void*>(kernel), s.ldsBytes);
        kernel<<<grid, nt, s.ldsBytes, stream>>>(a, te);
    };
    const bool pair = te.pairStore != 0;
    if (s.nt == 512 && s.depth == 3) {
        pair ? go(&staged_apply2_typed<STENCIL, 512, 6, T, true, 3>, 512) : go(&staged_apply2_typed<STENCIL, 512, 6, T, false, 3>, 512);
        return;
    }
    switch (s.nt) {
    case 256: pair ? go(&staged_apply2_typed<STENCIL, 256, 6, T, true>, 256) : go(&staged_apply2_typed<STENCIL, 256, 6, T, false>, 256); break;
    case 512: pair ? go(&staged_apply2_typed<STENCIL, 512, 6, T, true>, 512) : go(&staged_apply2_typed<STENCIL, 512, 6, T, false>, 512); break;
    case 1024: pair ? go(&staged_apply2_typed<STENCIL, 1024, 5, T, true>, 1024) : go(&staged_apply2_typed<STENCIL, 1024, 5, T, false>, 1024); break;
    default: throw Error("staged2 typed: unexpected workgroup shape");
    }
}

template <typename T>
void launch_typed_t(const fimex_amd_regrid_plan& plan, const Staged2Plan& s, const Staged2Args& a, const TypedEdge& te, dim3 grid, hipStream_t stream)
{
    if (plan.kind == PlanKind::Nearest) launch_typed_shape<1, T>(s, a, te, grid, stream);
    else launch_typed_shape<2, T>(s, a, te, grid, stream);
}

}  // namespace

// data2InterpolationArray + interpolateValues + interpolationArray2Data (src/CDMInterpolator.cc:115-124, 251-285) on slices of
// 1- and 2-byte integers, nearest and bilinear, through the second staged form.  false: not applicable, the caller goes on.
bool launch_staged2_apply_typed(const fimex_amd_regrid_plan& plan, const void* d_in, int cdmType, size_t nz, double badValue, void* d_out,
                                hipStream_t stream)
{
    if (!(cdmType == FIMEX_AMD_CDM_CHAR || cdmType == FIMEX_AMD_CDM_UCHAR || cdmType == FIMEX_AMD_CDM_SHORT || cdmType == FIMEX_AMD_CDM_USHORT))
        return false;
    const uint32_t eb = (cdmType == FIMEX_AMD_CDM_SHORT || cdmType == FIMEX_AMD_CDM_USHORT) ? 2u : 1u;
    if (reinterpret_cast<uintptr_t>(d_in) % 4 != 0) return false;
    if (nz == 0) return true;
    const Staged2Plan* form = staged2_typed_form(plan, eb, stream);
    if (!form) return false;
    const Staged2Plan& s = *form;
    Staged2Args a{};
    a.in = static_cast<const float*>(d_in);
    a.out = static_cast<float*>(d_out);
    a.tiles = s.tiles.get();
    a.order = s.order.get();
    a.chunkOff = s.chunkOff.get();
    a.ldsA = s.ldsA.get();
    a.ldsB = s.ldsB.get();
    a.pos = plan.pos.get();
    a.inX = (uint32_t)plan.inX;
    a.xf = plan.xf.get();
    a.yf = plan.yf.get();
    a.outX = (uint32_t)plan.outX;
    a.outY = (uint32_t)plan.outY;
    a.tileH = s.tileH;
    a.inBytes = (uint32_t)(plan.inX * plan.inY * eb);
    a.nOut = (uint32_t)(plan.outX * plan.outY);
    a.nz = (uint32_t)nz;
    // z chunks of about 50 slices (200 slices: 1.354 against 1.364 ms with 25, profiles/r03_sweep_typed*.log), at least four
    // where the batch allows, so that short batches still fill the chip
    uint32_t zpb = (uint32_t)tuning("STAGE2T_ZPB", 50);
    if (zpb < 1) zpb = 1;
    if (ceil_div(nz, (size_t)zpb) > 16) zpb = (uint32_t)ceil_div(nz, (size_t)16);
    uint32_t n = (uint32_t)ceil_div(nz, (size_t)zpb);
    n = std::max<uint32_t>(n, (uint32_t)std::min<size_t>(4, nz / 6));
    if (n < 1) n = 1;
    for (uint32_t c = 0, z = 0; c < n; ++c) {
        a.zStart[c] = z;
        z += (uint32_t)nz / n + (c < (uint32_t)nz % n ? 1u : 0u);
    }
    a.zStart[n] = (uint32_t)nz;
    a.nZChunks = n;
    a.slotChunks = slot_chunks(s.ldsBytes, s.depth);
    a.flags = (uint32_t)tuning("STAGE2_ABLATE", 0);
    TypedEdge te{};
    te.bad = (float)badValue;
    te.hasBad = !(te.bad != te.bad);
    te.fillOut = badValue;
    // two results per store where both the row length and the slice start allow aligned 4-byte (2-byte) stores
    te.pairStore = (plan.outX % 2 == 0 && reinterpret_cast<uintptr_t>(d_out) % 4 == 0 && tuning("STAGE2T_PAIR", 1) != 0) ? 1u : 0u;
    const dim3 grid(s.gridX * n, 1, 1);
    switch (cdmType) {
    case FIMEX_AMD_CDM_CHAR: launch_typed_t<signed char>(plan, s, a, te, grid, stream); break;
    case FIMEX_AMD_CDM_UCHAR: launch_typed_t<unsigned char>(plan, s, a, te, grid, stream); break;
    case FIMEX_AMD_CDM_SHORT: launch_typed_t<short>(plan, s, a, te, grid, stream); break;
    default: launch_typed_t<unsigned short>(plan, s, a, te, grid, stream); break;
    }
    FA_HIP(hipGetLastError());
    return true;
}

}  // namespace fimex_amd
