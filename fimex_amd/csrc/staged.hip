// LDS-staged backward regrid (bilinear and bicubic) for gfx950: the bandwidth path of the headline metric.
//
// Same arithmetic as bilinear_apply / bicubic_apply in regrid.hip (src/interpolation.c:881-1028), different data
// movement.  A per-lane gather of the stencil straight from global memory asks the memory system for the same
// 128-byte lines several times (measured on the 4000x3000 -> 2000x2000 rotated-pole case, profiles/LAB_NOTES_r01_r02.md, round 1: 3x the
// ideal L2 requests and 1.6x the ideal fabric reads for the 2x2 stencil; the 4x4 stencil ran at 17 % of the HBM
// roofline), because a wave's lanes stride through the source and neighbouring workgroups share lines but not
// their timing.  Here a workgroup owns a TW x TH tile of OUTPUT cells; the SOURCE cells that tile needs form a
// sheared band (the target grid is rotated against the source grid): per source row one contiguous x range.  The
// plan stores, per tile, that list of row segments (16-byte aligned, so the staged image is a little larger than
// the cells actually read).  For every z slice the workgroup
//   1. streams the segments HBM -> LDS with buffer_load_dwordx4 ... lds (LDS-DMA: 1 KiB per wave instruction, whole
//      lines, each requested once per tile, no VGPR round trip) into the other LDS buffer while
//   2. the current buffer is interpolated: stencils come from LDS (ds_read2_b32 pairs), and every wave writes
//      256 contiguous bytes of output per store, non-temporal.
// The per-output plan entry is two (bilinear) or four (bicubic) 16-bit LDS row offsets; the fractions are the ones of
// the gather plan.  Measured and kept out (round 1's sweeps, profiles/LAB_NOTES_r01_r02.md): register staging instead of LDS-DMA (equal),
// 2-4 slices of prefetch depth (equal), non-temporal loads (-15 %), plain stores (-6 %), XCD-contiguous or striped
// tile orders (equal or worse than plain round-robin).
#include "plan.hpp"
#include "staged_common.hpp"
#include "typed_convert.hpp"

#include <type_traits>

namespace fimex_amd {

namespace {

struct TileGeom {
    uint32_t outX, outY;
    uint32_t tileW, tileH;  // output cells per tile (tileW * tileH = 256 * outputs per lane)
    uint32_t tilesX, nTiles;
    uint32_t capChunks;     // 16-byte chunks one LDS buffer holds
};

struct BuildCounters {
    unsigned long long overflow, stagedChunks;
};

// One workgroup per tile: finds the row segments the tile reads and the LDS offsets of every output's stencil rows.
// tileRows[tile][2*i] = global cell offset of segment i, [2*i+1] = first chunk of segment i (prefix sum);
// tileHdr[tile] = {rows, chunks}; ldsA[cell] = offsets (in floats) of stencil rows 0 | 1 << 16, ldsB: rows 2 | 3 << 16.
template <int STENCIL>
__global__ void __launch_bounds__(kBlock) build_tiles(const double* __restrict__ px, const double* __restrict__ py, int64_t ix,
                                                      int64_t iy, TileGeom g, uint32_t* __restrict__ tileRows,
                                                      uint2* __restrict__ tileHdr, uint32_t* __restrict__ ldsA,
                                                      uint32_t* __restrict__ ldsB, BuildCounters* counters)
{
    __shared__ int shRmin, shRmax;
    __shared__ int rowMin[kMaxRows], rowMax[kMaxRows];
    __shared__ uint32_t rowChunk[kMaxRows + 1];
    __shared__ int shOverflow;
    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % g.tilesX, ty = tile / g.tilesX;
    const uint32_t perLane = (g.tileW * g.tileH) / kBlock;
    if (threadIdx.x == 0) { shRmin = 0x7FFFFFFF; shRmax = -1; shOverflow = 0; }
    __syncthreads();
    // lane owns column lx of rows ly0, ly0 + rowsPerPass, ... of the tile (a wave covers 64 consecutive x)
    const uint32_t lx = threadIdx.x % g.tileW;
    const uint32_t rowsPerPass = kBlock / g.tileW;
    const uint32_t ly0 = threadIdx.x / g.tileW;
    for (uint32_t k = 0; k < perLane; ++k) {
        const uint32_t x = tx * g.tileW + lx, y = ty * g.tileH + ly0 + k * rowsPerPass;
        if (x < g.outX && y < g.outY) {
            const size_t cell = (size_t)y * g.outX + x;
            const CellNeed c = classify<STENCIL>(px[cell], py[cell], ix, iy);
            if (c.valid) { atomicMin(&shRmin, (int)c.ya); atomicMax(&shRmax, (int)c.yb); }
        }
    }
    __syncthreads();
    const int rmin = shRmin;
    const int nr = (shRmax >= rmin) ? shRmax - rmin + 1 : 0;
    if (nr > kMaxRows) {
        if (threadIdx.x == 0) { atomicAdd(&counters->overflow, 1ull); tileHdr[tile] = make_uint2(0, 0); }
        return;
    }
    for (int i = threadIdx.x; i < nr; i += kBlock) { rowMin[i] = 0x7FFFFFFF; rowMax[i] = -1; }
    __syncthreads();
    for (uint32_t k = 0; k < perLane; ++k) {
        const uint32_t x = tx * g.tileW + lx, y = ty * g.tileH + ly0 + k * rowsPerPass;
        if (x < g.outX && y < g.outY) {
            const size_t cell = (size_t)y * g.outX + x;
            const CellNeed c = classify<STENCIL>(px[cell], py[cell], ix, iy);
            if (c.valid)
                for (int64_t r = c.ya; r <= c.yb; ++r) {
                    atomicMin(&rowMin[r - rmin], (int)c.xa);
                    atomicMax(&rowMax[r - rmin], (int)c.xb);
                }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int i = 0; i < nr; ++i) {
            rowChunk[i] = acc;
            if (rowMax[i] >= 0) {
                const int xs = rowMin[i] & ~3;  // 16-byte aligned start (ix % 4 == 0)
                rowMin[i] = xs;
                acc += (uint32_t)((rowMax[i] - xs) / 4 + 1);
            }
        }
        rowChunk[nr] = acc;
        if (acc > g.capChunks) shOverflow = 1;
    }
    __syncthreads();
    if (shOverflow) {
        if (threadIdx.x == 0) { atomicAdd(&counters->overflow, 1ull); tileHdr[tile] = make_uint2(0, 0); }
        return;
    }
    uint32_t* rows = tileRows + (size_t)tile * 2 * kMaxRows;
    for (int i = threadIdx.x; i < nr; i += kBlock) {
        rows[2 * i] = (rowMax[i] >= 0) ? (uint32_t)((int64_t)(rmin + i) * ix + rowMin[i]) : 0u;
        rows[2 * i + 1] = rowChunk[i];
    }
    if (threadIdx.x == 0) {
        tileHdr[tile] = make_uint2((uint32_t)nr, rowChunk[nr]);
        atomicAdd(&counters->stagedChunks, (unsigned long long)rowChunk[nr]);
    }
    for (uint32_t k = 0; k < perLane; ++k) {
        const uint32_t x = tx * g.tileW + lx, y = ty * g.tileH + ly0 + k * rowsPerPass;
        if (x < g.outX && y < g.outY) {
            const size_t cell = (size_t)y * g.outX + x;
            const CellNeed c = classify<STENCIL>(px[cell], py[cell], ix, iy);
            uint32_t a = kInvalidPos, b = kInvalidPos;
            if (c.valid) {
                uint32_t off[4];
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = (c.ya + r <= c.yb) ? c.ya + r : c.yb;  // missing rows repeat the last one
                    const int i = (int)(row - rmin);
                    off[r] = rowChunk[i] * 4 + (uint32_t)(c.xa - rowMin[i]);
                }
                a = off[0] | (off[1] << 16);
                b = off[2] | (off[3] << 16);
            }
            ldsA[cell] = a;
            if (STENCIL == 4) ldsB[cell] = b;
        }
    }
}

struct StagedArgs {
    const void* in;     // elements of T: float, or the variable's stored type
    void* out;
    float bad;          // stored types: the fill value narrowed to float (mifi_bad2nanf's parameter), whether there is one,
    int hasBad;         // and the fill value as interpolationArray2Data receives it
    double fillOut;
    const uint32_t* tileRows;
    const uint2* tileHdr;
    const uint32_t* ldsA;
    const uint32_t* ldsB;
    const float* xf;    // bilinear fractions of the gather plan (sign bit = nearest neighbour in that direction)
    const float* yf;
    const double* xfd;  // bicubic fractions
    const double* yfd;
    TileGeom g;
    size_t inLayer;
    uint32_t nOut;
    uint32_t nz, zPerBlock;
    uint32_t tilesPerXcd, xcdRemap, storeAux, loadAux;
    uint32_t nZChunks;  // > 0: flat grid in tile-major order (the z chunks of a tile are consecutive workgroups of one XCD)
    uint32_t ablate;  // tuning build only (FIMEX_AMD_ABLATE): 1 = no source loads, 2 = no output stores
};

// The same with 4 bytes per lane (256 bytes per wave instruction): sources of 1- and 2-byte elements, whose row segments
// start on 4-byte but not on 16-byte boundaries.
__device__ __forceinline__ void dma4(rsrc_t rs, void* ldsBase, uint32_t voff)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using lds_ptr = __attribute__((address_space(3))) void*;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)ldsBase, 4, voff, 0, 0, 0);
#else
    (void)rs; (void)ldsBase; (void)voff;
#endif
}

// one element of the staged image as float (stored types: Data::asFloat + mifi_bad2nanf, typed_convert.hpp)
template <typename T>
__device__ __forceinline__ float lds_value(const char* buf, uint32_t floatByteOff, float bad, bool hasBad)
{
    if constexpr (std::is_same<T, float>::value) return *reinterpret_cast<const float*>(buf + floatByteOff);
    else return as_float_nan(*reinterpret_cast<const T*>(buf + floatByteOff / 4 * sizeof(T)), bad, hasBad);
}

// two neighbouring 1- or 2-byte elements: the two dwords that hold them in one ds_read2_b32, shifted into place
template <typename T>
__device__ __forceinline__ void lds_pair(const char* buf, uint32_t floatByteOff, float bad, bool hasBad, float& first, float& second)
{
    const uint32_t byteOff = floatByteOff / 4 * sizeof(T);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(buf + (byteOff & ~3u));
    const uint32_t both = __builtin_amdgcn_alignbit(w[1], w[0], (byteOff & 3u) * 8u);
    if constexpr (sizeof(T) == 2) {
        first = as_float_nan((T)(unsigned short)(both & 0xffffu), bad, hasBad);
        second = as_float_nan((T)(unsigned short)(both >> 16), bad, hasBad);
    } else {
        first = as_float_nan((T)(unsigned char)(both & 0xffu), bad, hasBad);
        second = as_float_nan((T)(unsigned char)((both >> 8) & 0xffu), bad, hasBad);
    }
}

template <typename T>
__device__ __forceinline__ void store_result(rsrc_t ro, uint32_t cellByteOff, float r, T fill)
{
    if constexpr (std::is_same<T, float>::value) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), ro, cellByteOff, 0, 2);
    else {
        const T v = from_float_fill<T>(r, fill);  // interpolationArray2Data: NaN -> fill value, integers rounded
        if constexpr (sizeof(T) == 1) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, ro, cellByteOff / 4, 0, 2);
        else if constexpr (sizeof(T) == 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, ro, cellByteOff / 2, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b32((unsigned int)v, ro, cellByteOff, 0, 2);
    }
}

// NBUF: LDS buffers of the slice ring; NBUF - 1 slices are in flight while one is interpolated.  What a workgroup waits
// for per slice is the latency of its DMA, so the bytes in flight per CU (LDS capacity x (NBUF - 1) / NBUF) set the rate.
// T: float, or the variable's stored type -- the same tiles and LDS offsets (in elements), the staged image holds
// elements of T (1- and 2-byte types arrive through 4-byte DMA pieces), values become float on the LDS read and go back to
// T on the store.
template <int STENCIL, int PER, int KMAX, int NBUF, typename T = float>
__global__ void __launch_bounds__(kBlock) staged_apply(StagedArgs a)
{
    constexpr bool kFloat = std::is_same<T, float>::value;
    constexpr uint32_t EB = sizeof(T);
    // DMA pieces per lane and slice: 16-byte chunks for 4-byte elements, 4-byte pieces otherwise (a chunk of the plan is
    // 4 elements = 4 * EB bytes = EB pieces)
    // 2-byte elements: 16-byte pieces of 8 elements = two chunks; every row segment is padded to an even number of chunks in
    // the LDS image (kWide), so that a piece never spans two segments -- a quarter of the DMA instructions of 4-byte pieces,
    // which matter more than the bytes (each vector-memory wave instruction costs the address unit ~16 clocks).
    // 1-byte elements: 4-byte pieces (a chunk is one piece).
    constexpr bool kWide = (EB == 2);
    constexpr int UN = (EB == 4) ? KMAX : (kWide ? KMAX / 2 + 1 : KMAX);
    const bool hasBad = a.hasBad != 0;
    const T fillT = kFloat ? T() : static_cast<T>(a.fillOut);  // ScaleValue's newFill_ (Utils.h:456)
    extern __shared__ __attribute__((aligned(16))) float smem[];  // NBUF buffers of KMAX*256*4 floats (+ slack), then the row table
    constexpr uint32_t kBufFloats = KMAX * kBlock * 4 + 4;
    uint32_t* shRows = reinterpret_cast<uint32_t*>(smem + NBUF * kBufFloats);  // [2 * nr]
    uint32_t* shPiece = shRows + 2 * kMaxRows;                                   // kWide: first piece of every segment, [nr + 1]

    // workgroup -> tile.  Workgroups are dealt round-robin over the 8 XCDs (b % 8 shares an L2):
    //   xcdRemap 0: tiles in dispatch order (neighbours on different XCDs) -- the default, measured as good as any;
    //   xcdRemap 1: each XCD owns one contiguous band of tile rows;
    //   xcdRemap >= 2: tile rows dealt to the XCDs in stripes of (xcdRemap - 1) rows.
    uint32_t b = blockIdx.x, zc = blockIdx.y;
    if (a.nZChunks != 0) {  // workgroup s runs on XCD s % 8; the k-th workgroup of an XCD is z chunk k % n of the XCD's tile k / n
        const uint32_t k = blockIdx.x / kXcds;
        zc = k % a.nZChunks;
        b = (k / a.nZChunks) * kXcds + blockIdx.x % kXcds;
    }
    uint32_t tile = b;
    if (a.xcdRemap == 1) {
        tile = (b % kXcds) * a.tilesPerXcd + b / kXcds;
    } else if (a.xcdRemap >= 2) {
        const uint32_t stripe = (a.xcdRemap - 1) * a.g.tilesX;
        const uint32_t idx = b / kXcds;
        tile = ((idx / stripe) * kXcds + b % kXcds) * stripe + idx % stripe;
    }
    if (tile >= a.g.nTiles) return;
    const uint32_t z0 = zc * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const uint2 hdr = a.tileHdr[tile];
    const uint32_t nr = hdr.x, totalChunks = hdr.y;

    // ---- per-lane plan: outputs, LDS byte offsets of their stencil rows, weights
    const uint32_t tx = tile % a.g.tilesX, ty = tile / a.g.tilesX;
    const uint32_t lx = threadIdx.x % a.g.tileW;
    const uint32_t rowsPerPass = kBlock / a.g.tileW;
    const uint32_t ly0 = threadIdx.x / a.g.tileW;
    uint32_t cellOff[PER];        // byte offset of the output cell inside a slice; ~0u (not mine) is dropped by the bounds check
    uint32_t row[PER][STENCIL];   // LDS byte offsets of the stencil rows
    float xf[PER], yf[PER];       // bilinear
    double XM[PER][4], MY[PER][4];  // bicubic (unused and eliminated for bilinear)
    bool undef[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t x = tx * a.g.tileW + lx, y = ty * a.g.tileH + ly0 + k * rowsPerPass;
        cellOff[k] = 0xFFFFFFFFu;
        uint32_t pa = kInvalidPos, pb = kInvalidPos;
        xf[k] = yf[k] = 0.f;
        double fx = 0, fy = 0;
        if (x < a.g.outX && y < a.g.outY) {
            const uint32_t cell = y * a.g.outX + x;
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
        }
    }
    const uint32_t outBytes = a.nOut * EB;
    const char* inBase = static_cast<const char*>(a.in);
    char* outBase = static_cast<char*>(a.out);

    if (totalChunks == 0) {  // nothing of the source is needed: every output of the tile is undefined
        for (uint32_t z = z0; z < z1; ++z) {
            const rsrc_t ro = make_rsrc(outBase + (size_t)z * outBytes, outBytes);
#pragma unroll
            for (int k = 0; k < PER; ++k) store_result<T>(ro, cellOff[k], undefined_f(), fillT);
        }
        return;
    }

    // ---- per-lane staging list: chunk c = threadIdx.x + j*256 of the tile's row segments
    const uint32_t* rows = a.tileRows + (size_t)tile * 2 * kMaxRows;
    for (uint32_t i = threadIdx.x; i < 2 * nr; i += kBlock) shRows[i] = rows[i];
    __syncthreads();
    auto segment_of_chunk = [&](uint32_t c) {  // last segment whose first chunk <= c
        uint32_t lo = 0, hi = nr - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (shRows[2 * mid + 1] <= c) lo = mid; else hi = mid - 1;
        }
        return lo;
    };
    if constexpr (kWide) {
        if (threadIdx.x == 0) {
            uint32_t acc = 0;
            for (uint32_t r = 0; r < nr; ++r) {
                shPiece[r] = acc;
                const uint32_t nch = (r + 1 < nr ? shRows[2 * (r + 1) + 1] : totalChunks) - shRows[2 * r + 1];
                acc += (nch + 1) / 2;
            }
            shPiece[nr] = acc;
        }
        __syncthreads();
        // the plan's LDS offsets count elements of an image without padding: move them to the padded one
#pragma unroll
        for (int k = 0; k < PER; ++k) {
#pragma unroll
            for (int i = 0; i < STENCIL; ++i) {
                if (undef[k]) continue;
                const uint32_t o = row[k][i] / 4u;                     // element offset in the unpadded image
                const uint32_t r = segment_of_chunk(o / 4u);
                row[k][i] = (shPiece[r] * 8u + (o - shRows[2 * r + 1] * 4u)) * 4u;
            }
        }
    }
    // piece u = threadIdx.x + j*256: 16-byte chunk u (4-byte elements), 16-byte piece u of the padded image (2-byte
    // elements), or 4-byte chunk u (1-byte elements)
    uint32_t gOff[UN];  // byte offset of the piece inside a source slice, ~0u = none (dropped by the bounds check: zeros)
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const uint32_t u = threadIdx.x + j * kBlock;
        gOff[j] = 0xFFFFFFFFu;
        if constexpr (kWide) {
            if (u < shPiece[nr]) {
                uint32_t lo = 0, hi = nr - 1;  // last segment whose first piece <= u
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    if (shPiece[mid] <= u) lo = mid; else hi = mid - 1;
                }
                // (the last piece of a segment with an odd number of chunks reads four elements past it: never used)
                gOff[j] = (shRows[2 * lo] + (u - shPiece[lo]) * 8u) * EB;
            }
        } else if (u < totalChunks) {
            const uint32_t lo = segment_of_chunk(u);
            gOff[j] = (shRows[2 * lo] + (u - shRows[2 * lo + 1]) * 4u) * EB;  // first element of the chunk
        }
    }

    const uint32_t inBytes = (uint32_t)a.inLayer * EB;
    // one wave instruction moves 64 pieces (1 KiB or 256 bytes): LDS destination = wave-uniform base + lane * piece size
    const uint32_t waveChunk = (threadIdx.x / kWave) * kWave;
    auto dma = [&](float* dst, uint32_t z) {
        const rsrc_t rs = make_rsrc(inBase + (size_t)z * inBytes, (kTuningBuild && (a.ablate & 1)) ? 0u : inBytes);
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            if constexpr (EB == 4 || kWide) dma16(rs, dst + (waveChunk + j * kBlock) * 4, gOff[j], a.loadAux);
            else dma4(rs, dst + (waveChunk + j * kBlock), gOff[j]);
        }
    };

    // prologue: NBUF - 1 slices in flight, the first one landed (NBUF == 1: no prefetch, the slice is fetched, then used)
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (z0 + i < z1) dma(smem + i * kBufFloats, z0 + i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    uint32_t slot = 0;
    for (uint32_t z = z0; z < z1; ++z) {
        const bool more = z + (NBUF - 1) < z1;
        if (NBUF == 1) {
            dma(smem, z);
            wait_vmcnt<0>();  // results return in issue order: the DMA is the most recent, everything before it has to land too
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else if (more) {
            dma(smem + ((slot + NBUF - 1) % NBUF) * kBufFloats, z + (NBUF - 1));  // into the buffer slice z - 1 has left
        }
        const float* cur = smem + slot * kBufFloats;
        const rsrc_t ro = make_rsrc(outBase + (size_t)z * outBytes, (kTuningBuild && (a.ablate & 2)) ? 0u : outBytes);
        const char* curb = reinterpret_cast<const char*>(cur);
        if constexpr (STENCIL == 1) {
            float v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) v[k] = lds_value<T>(curb, row[k][0], a.bad, hasBad);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                if constexpr (kFloat) __builtin_amdgcn_raw_buffer_store_b32(undef[k] ? 0x7fc00000u : __float_as_uint(v[k]), ro, cellOff[k], 0, 2);  // :869-876
                else store_result<T>(ro, cellOff[k], undef[k] ? undefined_f() : v[k], fillT);
            }
        } else if constexpr (STENCIL == 2) {
            float s00[PER], s01[PER], s10[PER], s11[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {  // all stencil reads first: 2 x ds_read2_b32 per output, no waits in between
                if constexpr (kFloat) {
                    const float* pa = reinterpret_cast<const float*>(curb + row[k][0]);
                    const float* pb = reinterpret_cast<const float*>(curb + row[k][1]);
                    s00[k] = pa[0]; s01[k] = pa[1]; s10[k] = pb[0]; s11[k] = pb[1];
                } else if constexpr (EB == 4) {
                    s00[k] = lds_value<T>(curb, row[k][0], a.bad, hasBad); s01[k] = lds_value<T>(curb, row[k][0] + 4, a.bad, hasBad);
                    s10[k] = lds_value<T>(curb, row[k][1], a.bad, hasBad); s11[k] = lds_value<T>(curb, row[k][1] + 4, a.bad, hasBad);
                } else {
                    lds_pair<T>(curb, row[k][0], a.bad, hasBad, s00[k], s01[k]);
                    lds_pair<T>(curb, row[k][1], a.bad, hasBad, s10[k], s11[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const bool nnx = (__float_as_uint(xf[k]) >> 31) != 0, nny = (__float_as_uint(yf[k]) >> 31) != 0;
                // interior (interpolation.c:899-900); its upper row is the "linear in x, nearest in y" value (:911)
                const float top = (1.f - xf[k]) * s00[k] + xf[k] * s01[k];
                const float bot = (1.f - xf[k]) * s10[k] + xf[k] * s11[k];
                const float inter = (1.f - yf[k]) * top + yf[k] * bot;
                const float liny = (1 - yf[k]) * s00[k] + (yf[k] * s10[k]);  // nearest in x, linear in y (:931)
                float r = nnx ? (nny ? s00[k] : liny) : (nny ? top : inter);
                r = undef[k] ? undefined_f() : r;
                if constexpr (!kFloat) { store_result<T>(ro, cellOff[k], r, fillT); continue; }
                switch (a.storeAux) {  // cache policy of the result stores (wave-uniform): 2 = non-temporal is the default
                case 0: __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), ro, cellOff[k], 0, 0); break;
                case 1: __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), ro, cellOff[k], 0, 1); break;
                case 3: __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), ro, cellOff[k], 0, 3); break;
                case 16: __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), ro, cellOff[k], 0, 16); break;
                case 17: __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), ro, cellOff[k], 0, 17); break;
                case 18: __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), ro, cellOff[k], 0, 18); break;
                case 19: __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), ro, cellOff[k], 0, 19); break;
                default: __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), ro, cellOff[k], 0, 2); break;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                float f[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[i][j] = lds_value<T>(curb, row[k][i] + 4 * j, a.bad, hasBad);
                }
                float acc = 0;  // interpolation.c:1005: accumulates into the float output
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double xmf = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xmf += XM[k][j] * (double)f[i][j];  // :1015
                    acc = (float)((double)acc + xmf * MY[k][i]);                    // :1019
                }
                const float r = undef[k] ? undefined_f() : acc;
                store_result<T>(ro, cellOff[k], r, fillT);
            }
        }
        // Slice z + 1 must have landed.  Results come back in issue order: behind its DMA are the DMAs of slices
        // z + 2 .. z + NBUF - 1 and the stores of NBUF - 1 slices, which may all stay in flight.  At the end of the run
        // (no new DMA issued) only this slice's stores may.
        if (NBUF == 1) { /* the buffer is rewritten after the barrier below */ }
        else if (more) wait_vmcnt<(NBUF >= 2 ? (NBUF - 2) * UN + (NBUF - 1) * PER : 0)>();
        else wait_vmcnt<PER>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot = (slot + 1 == NBUF) ? 0 : slot + 1;
    }
}

template <int STENCIL, int PER, int KMAX, int NBUF, typename T>
void launch_staged_n(const StagedArgs& a, dim3 grid, hipStream_t stream)
{
    constexpr size_t ldsBytes = (size_t)NBUF * (KMAX * kBlock * 4 + 4) * sizeof(float) + (3 * kMaxRows + 1) * sizeof(uint32_t);
    static_assert(ldsBytes <= 160 * 1024, "slice ring does not fit the CU's LDS");
    allow_dynamic_lds(reinterpret_cast<const void*>(&staged_apply<STENCIL, PER, KMAX, NBUF, T>), ldsBytes);
    staged_apply<STENCIL, PER, KMAX, NBUF, T><<<grid, kBlock, ldsBytes, stream>>>(a);
}

template <int STENCIL, int PER, int KMAX, typename T>
void launch_staged(const StagedArgs& a, dim3 grid, hipStream_t stream)
{
    constexpr bool kFloat = std::is_same<T, float>::value;
    constexpr size_t buf = (size_t)(KMAX * kBlock * 4 + 4) * sizeof(float);
    constexpr int UN = sizeof(T) == 4 ? KMAX : KMAX * (int)sizeof(T);
    // a ring of 3 was measured for the stored types as well (256-byte DMA instructions): slower than 2 (2.11 against 1.87 ms)
    const int nbuf = kFloat ? tuning("STAGE_NBUF", 2) : tuning("TYPED_NBUF", 2);
    if constexpr (kFloat) {  // 1, 4 and 6 are experiment switches of the float kernel
        if (nbuf == 1) { launch_staged_n<STENCIL, PER, KMAX, 1, T>(a, grid, stream); return; }
        if constexpr (4 * buf + 2 * kMaxRows * sizeof(uint32_t) <= 160 * 1024) {
            if (nbuf == 4) { launch_staged_n<STENCIL, PER, KMAX, 4, T>(a, grid, stream); return; }
        }
        if constexpr (6 * buf + 2 * kMaxRows * sizeof(uint32_t) <= 160 * 1024 && (4 * KMAX + 5 * PER) < 64) {
            if (nbuf == 6) { launch_staged_n<STENCIL, PER, KMAX, 6, T>(a, grid, stream); return; }
        }
    }
    if constexpr (3 * buf + 2 * kMaxRows * sizeof(uint32_t) <= 160 * 1024 && (UN + 2 * PER) < 64) {
        if (nbuf == 3) { launch_staged_n<STENCIL, PER, KMAX, 3, T>(a, grid, stream); return; }
    }
    launch_staged_n<STENCIL, PER, KMAX, 2, T>(a, grid, stream);
}

template <int STENCIL>
bool try_build_staged(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream, uint32_t tileW,
                      uint32_t per, uint32_t kmax)
{
    TileGeom g{};
    g.outX = (uint32_t)plan.outX;
    g.outY = (uint32_t)plan.outY;
    g.tileW = tileW;
    g.tileH = per * kBlock / tileW;
    g.tilesX = (uint32_t)ceil_div(plan.outX, tileW);
    g.nTiles = g.tilesX * (uint32_t)ceil_div(plan.outY, g.tileH);
    g.capChunks = kmax * kBlock;
    const size_t n = plan.outX * plan.outY;
    DeviceArray<uint32_t> tileRows((size_t)g.nTiles * 2 * kMaxRows);
    DeviceArray<uint2> tileHdr(g.nTiles);
    DeviceArray<uint32_t> ldsA(n), ldsB(STENCIL == 4 ? n : 0);
    DeviceArray<BuildCounters> counters(1);
    FA_HIP(hipMemsetAsync(counters.get(), 0, sizeof(BuildCounters), stream));
    build_tiles<STENCIL><<<g.nTiles, kBlock, 0, stream>>>(d_px, d_py, (int64_t)plan.inX, (int64_t)plan.inY, g, tileRows.get(),
                                                         tileHdr.get(), ldsA.get(), ldsB.get(), counters.get());
    FA_HIP(hipGetLastError());
    BuildCounters h{};
    FA_HIP(hipMemcpyAsync(&h, counters.get(), sizeof(h), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    if (h.overflow != 0) return false;  // some tile's footprint does not fit this shape
    plan.staged.tileW = g.tileW;
    plan.staged.tileH = g.tileH;
    plan.staged.per = per;
    plan.staged.kmax = kmax;
    plan.staged.tilesX = g.tilesX;
    plan.staged.nTiles = g.nTiles;
    plan.staged.stagedCells = (size_t)h.stagedChunks * 4;
    plan.staged.tileRows = std::move(tileRows);
    plan.staged.tileHdr = std::move(tileHdr);
    plan.staged.ldsA = std::move(ldsA);
    plan.staged.ldsB = std::move(ldsB);
    plan.staged.valid = true;
    return true;
}

}  // namespace

// Chooses the smallest LDS budget whose tiles all fit; plans without spatial coherence (or with a source row length
// that breaks the 16-byte alignment of row starts) keep only the gather kernel.
bool build_staged_plan(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream)
{
    if (plan.inX % 4 != 0) return false;
    // output columns per tile: wide tiles fetch fewer lines twice (bilinear on the benchmark plan: 14.0 / 11.7 / 10.6 GB per
    // launch at 32 / 64 / 128 columns) but are only 1-2 % faster -- the duplicates are served by the memory-side cache;
    // the 4x4 stencil prefers squarer tiles (fewer halo rows)
    const uint32_t tw = (uint32_t)tuning("STAGE_TW", plan.kind == PlanKind::Bicubic ? 32 : 128);
    if (!(tw == 32 || tw == 64 || tw == 128 || tw == 256)) return false;
    const int forcedPer = tuning("STAGE_PER", 0), forcedK = tuning("STAGE_K", 0);
    if (plan.kind == PlanKind::Bilinear) {
        const uint32_t shapes[4][2] = {{4, 4}, {4, 6}, {8, 8}, {8, 12}};  // {outputs per lane, chunks per lane}
        for (const auto& sh : shapes) {
            if (forcedPer && (uint32_t)forcedPer != sh[0]) continue;
            if (forcedK && (uint32_t)forcedK != sh[1]) continue;
            if (try_build_staged<2>(plan, d_px, d_py, stream, tw, sh[0], sh[1])) return true;
        }
    } else if (plan.kind == PlanKind::Nearest) {
        const uint32_t shapes[3][2] = {{4, 4}, {4, 6}, {8, 8}};
        for (const auto& sh : shapes) {
            if (forcedPer && (uint32_t)forcedPer != sh[0]) continue;
            if (forcedK && (uint32_t)forcedK != sh[1]) continue;
            if (try_build_staged<1>(plan, d_px, d_py, stream, tw, sh[0], sh[1])) return true;
        }
    } else if (plan.kind == PlanKind::Bicubic) {
        const uint32_t shapes[4][2] = {{2, 3}, {2, 4}, {4, 6}, {4, 8}};
        for (const auto& sh : shapes) {
            if (forcedPer && (uint32_t)forcedPer != sh[0]) continue;
            if (forcedK && (uint32_t)forcedK != sh[1]) continue;
            if (try_build_staged<4>(plan, d_px, d_py, stream, tw, sh[0], sh[1])) return true;
        }
    }
    return false;
}

namespace {

template <typename T>
void launch_staged_t(const fimex_amd_regrid_plan& plan, StagedArgs& a, size_t nz, hipStream_t stream)
{
    const auto& s = plan.staged;
    a.tileRows = s.tileRows.get();
    a.tileHdr = s.tileHdr.get();
    a.ldsA = s.ldsA.get();
    a.ldsB = s.ldsB.get();
    a.xf = plan.xf.get();
    a.yf = plan.yf.get();
    a.xfd = plan.xfd.get();
    a.yfd = plan.yfd.get();
    a.g.outX = (uint32_t)plan.outX;
    a.g.outY = (uint32_t)plan.outY;
    a.g.tileW = s.tileW;
    a.g.tileH = s.tileH;
    a.g.tilesX = s.tilesX;
    a.g.nTiles = s.nTiles;
    a.g.capChunks = s.kmax * kBlock;
    a.inLayer = plan.inX * plan.inY;
    a.nOut = (uint32_t)(plan.outX * plan.outY);
    a.nz = (uint32_t)nz;
    // STAGE_ORDER 1: tile-major launch order as in staged2.hip (the z chunks of a tile start together on one XCD, tile rows dealt
    // to the XCDs one by one); 0: all tiles of z chunk 0, then chunk 1, ...
    const bool tileMajor = tuning("STAGE_ORDER", 0) == 1;
    uint32_t zpb = (uint32_t)tuning("STAGE_ZPB", tileMajor ? 25 : 50);
    if (zpb < 1) zpb = 1;
    if (zpb > nz) zpb = (uint32_t)nz;
    a.zPerBlock = zpb;
    // tile -> XCD map: stored types (1 and 2 bytes) with the 1 x 1 and 2 x 2 stencils run 3.5-4.3 % faster with tile rows dealt to the
    // XCDs one by one (packed shorts, 200 slices: bilinear 1.71 -> 1.65 ms, nearest 1.56 -> 1.49 ms in three processes,
    // profiles/r02_typed_order*.jsonl); floats and the 4 x 4 stencil keep the dispatch order
    const bool rowsToXcds = tileMajor || (!std::is_same<T, float>::value && plan.kind != PlanKind::Bicubic);
    a.xcdRemap = (uint32_t)tuning("XCD", rowsToXcds ? 2 : 0);
    a.ablate = (uint32_t)tuning("ABLATE", 0);
    a.storeAux = (uint32_t)tuning("STORE_AUX", 2);
    a.loadAux = (uint32_t)tuning("LOAD_AUX", 0);
    a.tilesPerXcd = (uint32_t)ceil_div(s.nTiles, kXcds);
    uint32_t gridX = a.tilesPerXcd * kXcds;
    if (a.xcdRemap >= 2) {  // whole stripes per XCD
        const uint32_t stripe = (a.xcdRemap - 1) * s.tilesX;
        gridX = (uint32_t)ceil_div(s.nTiles, (size_t)stripe * kXcds) * stripe * kXcds;
    }
    const size_t chunks = ceil_div(nz, (size_t)zpb);
    FA_REQUIRE(chunks <= 65535 && (size_t)gridX * chunks <= 0x7FFFFFFFu, "too many z chunks for one launch");
    a.nZChunks = tileMajor ? (uint32_t)chunks : 0u;
    const dim3 grid(tileMajor ? gridX * (uint32_t)chunks : gridX, tileMajor ? 1u : (uint32_t)chunks, 1);
    const uint32_t key = s.per * 100 + s.kmax;
    if (plan.kind == PlanKind::Nearest) {
        switch (key) {
        case 404: launch_staged<1, 4, 4, T>(a, grid, stream); break;
        case 406: launch_staged<1, 4, 6, T>(a, grid, stream); break;
        case 808: launch_staged<1, 8, 8, T>(a, grid, stream); break;
        default: throw Error("staged nearest: unexpected tile shape");
        }
    } else if (plan.kind == PlanKind::Bilinear) {
        switch (key) {
        case 404: launch_staged<2, 4, 4, T>(a, grid, stream); break;
        case 406: launch_staged<2, 4, 6, T>(a, grid, stream); break;
        case 808: launch_staged<2, 8, 8, T>(a, grid, stream); break;
        case 812: launch_staged<2, 8, 12, T>(a, grid, stream); break;
        default: throw Error("staged bilinear: unexpected tile shape");
        }
    } else {
        switch (key) {
        case 203: launch_staged<4, 2, 3, T>(a, grid, stream); break;
        case 204: launch_staged<4, 2, 4, T>(a, grid, stream); break;
        case 406: launch_staged<4, 4, 6, T>(a, grid, stream); break;
        case 408: launch_staged<4, 4, 8, T>(a, grid, stream); break;
        default: throw Error("staged bicubic: unexpected tile shape");
        }
    }
    FA_HIP(hipGetLastError());
}

}  // namespace

void launch_staged_apply(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream)
{
    StagedArgs a{};
    a.in = d_in;
    a.out = d_out;
    launch_staged_t<float>(plan, a, nz, stream);
}

// The staged kernels on a variable's stored type (1- and 2-byte integers): what typed_apply (regrid.hip) does with
// gathers, through the LDS image.  false: not applicable (type, alignment), the caller takes another path.
bool launch_staged_apply_typed(const fimex_amd_regrid_plan& plan, const void* d_in, int cdmType, size_t nz, double badValue, void* d_out,
                               hipStream_t stream)
{
    if (!plan.staged.valid) return false;
    if (!(cdmType == FIMEX_AMD_CDM_CHAR || cdmType == FIMEX_AMD_CDM_UCHAR || cdmType == FIMEX_AMD_CDM_SHORT || cdmType == FIMEX_AMD_CDM_USHORT))
        return false;
    // 4-byte DMA pieces: slices and row segments start on 4-byte boundaries (inX % 4 == 0 holds for every staged plan)
    if (reinterpret_cast<uintptr_t>(d_in) % 4 != 0) return false;
    StagedArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.bad = (float)badValue;
    a.hasBad = !(a.bad != a.bad);
    a.fillOut = badValue;
    switch (cdmType) {
    case FIMEX_AMD_CDM_CHAR: launch_staged_t<signed char>(plan, a, nz, stream); break;
    case FIMEX_AMD_CDM_UCHAR: launch_staged_t<unsigned char>(plan, a, nz, stream); break;
    case FIMEX_AMD_CDM_SHORT: launch_staged_t<short>(plan, a, nz, stream); break;
    default: launch_staged_t<unsigned short>(plan, a, nz, stream); break;
    }
    return true;
}

}  // namespace fimex_amd
