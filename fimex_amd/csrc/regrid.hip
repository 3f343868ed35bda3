// Backward-mapping regrid for gfx950: plan classification and the nearest / bilinear /
// bicubic apply kernels.
//
// Replaces the per-cell function-pointer loop of CachedInterpolation::interpolateValues
// (src/CachedInterpolation.cc:118-147) around mifi_get_values_f / _bilinear_f / _bicubic_f
// (src/interpolation.c:862-1028).
//
// Mapping to the machine (HBM-bound gather, no MFMA):
//  * one lane per output cell, 256-lane workgroups over 256 consecutive output cells, so every
//    store instruction of a wave writes 256 contiguous bytes and neighbouring lanes gather from
//    neighbouring source cells (the same few 128-B lines);
//  * the z loop (time x level slices) runs inside the lane: the plan entry is read once and kept
//    in registers, and ZC slices are in flight per lane (4*ZC independent loads for bilinear)
//    to cover HBM latency;
//  * workgroups are dealt round-robin over the 8 XCDs, each with a private L2; the tile index is
//    remapped so that one XCD owns a contiguous band of output rows and the source rows shared by
//    vertically adjacent tiles are fetched into one L2 only;
//  * outputs are written once and never re-read: non-temporal stores keep them out of the way of
//    the source lines in L2.
//
// Built with -ffp-contract=off: every multiply and add below is a separate IEEE operation in
// the same type and order as the reference, so results are bit-identical to the CPU path.
#include "plan.hpp"

#include <type_traits>

#include <algorithm>
#include "typed_convert.hpp"

namespace fimex_amd {

namespace {

__device__ __forceinline__ float undefined_f() { return __uint_as_float(0x7fc00000u); }  // MIFI_UNDEFINED_F

// coordinates beyond this, NaN or inf are "outside" (the reference casts them to int: undefined behaviour)
__device__ __forceinline__ bool usable(double x, double y)
{
    const double lim = 1073741824.0;
    return (fabs(x) < lim) && (fabs(y) < lim);  // false for NaN
}

struct PlanCounters {
    unsigned long long undefined;
    unsigned long long border;
};

// ---------------------------------------------------------------- plan classification
// src/interpolation.c:864-868
__global__ void __launch_bounds__(kBlock) classify_nearest(const double* __restrict__ px, const double* __restrict__ py,
                                                           uint32_t n, int64_t ix, int64_t iy,
                                                           uint32_t* __restrict__ pos, PlanCounters* counters)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double x = px[i], y = py[i];
    uint32_t p = kInvalidPos;
    if (usable(x, y)) {
        const int64_t rx = (int64_t)round(x);  // lround: half away from zero
        const int64_t ry = (int64_t)round(y);
        if (rx >= 0 && rx < ix && ry >= 0 && ry < iy) p = (uint32_t)(ry * ix + rx);
    }
    pos[i] = p;
    if (p == kInvalidPos) atomicAdd(&counters->undefined, 1ull);
}

// src/interpolation.c:883-954 without the z loops
__global__ void __launch_bounds__(kBlock) classify_bilinear(const double* __restrict__ px, const double* __restrict__ py,
                                                            uint32_t n, int64_t ix, int64_t iy,
                                                            uint32_t* __restrict__ pos, float* __restrict__ xf,
                                                            float* __restrict__ yf, PlanCounters* counters)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double x = px[i], y = py[i];
    uint32_t p = kInvalidPos;
    float fx = 0.f, fy = 0.f;
    bool border = false;
    if (usable(x, y)) {
        const double flx = floor(x), fly = floor(y);
        const int64_t x0 = (int64_t)flx, y0 = (int64_t)fly;
        fx = (float)(x - flx);  // :885, double difference rounded to float
        fy = (float)(y - fly);  // :888
        const bool xlin = (0 <= x0) && (x0 + 1 < ix);
        const bool ylin = (0 <= y0) && (y0 + 1 < iy);
        if (xlin && ylin) {
            p = (uint32_t)(y0 * ix + x0);
        } else if (xlin) {
            const int64_t ry = (int64_t)round(y);  // :904
            if (0 <= ry && ry < iy) { p = (uint32_t)(ry * ix + x0); fy = -1.f; border = true; }
        } else {
            const int64_t rx = (int64_t)round(x);  // :922
            if (0 <= rx && rx < ix) {
                if (ylin) {
                    p = (uint32_t)(y0 * ix + rx); fx = -1.f; border = true;
                } else {
                    const int64_t ry = (int64_t)round(y);  // :935
                    // the reference tests "ry <= iy" (:936) and then reads past the slice; undefined here
                    if (0 <= ry && ry < iy) { p = (uint32_t)(ry * ix + rx); fx = -1.f; fy = -1.f; border = true; }
                }
            }
        }
    }
    pos[i] = p;
    xf[i] = fx;
    yf[i] = fy;
    if (p == kInvalidPos) atomicAdd(&counters->undefined, 1ull);
    if (border) atomicAdd(&counters->border, 1ull);
}

// src/interpolation.c:970-976
__global__ void __launch_bounds__(kBlock) classify_bicubic(const double* __restrict__ px, const double* __restrict__ py,
                                                           uint32_t n, int64_t ix, int64_t iy,
                                                           uint32_t* __restrict__ pos, double* __restrict__ xfd,
                                                           double* __restrict__ yfd, PlanCounters* counters)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double x = px[i], y = py[i];
    uint32_t p = kInvalidPos;
    double fx = 0., fy = 0.;
    if (usable(x, y)) {
        const double flx = floor(x), fly = floor(y);
        const int64_t x0 = (int64_t)flx, y0 = (int64_t)fly;
        fx = x - flx;
        fy = y - fly;
        if ((1 <= x0) && (x0 + 2 < ix) && (1 <= y0) && (y0 + 2 < iy)) p = (uint32_t)((y0 - 1) * ix + (x0 - 1));
    }
    pos[i] = p;
    xfd[i] = fx;
    yfd[i] = fy;
    if (p == kInvalidPos) atomicAdd(&counters->undefined, 1ull);
}

// ------------------------------------------------------------------------ apply kernels
struct ApplyArgs {
    const float* in;
    float* out;
    uint32_t nOut;         // cells per output slice
    uint32_t ix;           // source row length
    size_t inLayer;        // cells per source slice
    uint32_t nz;
    uint32_t zPerBlock;    // slices handled by one workgroup (blockIdx.y selects the chunk)
    uint32_t nTiles;       // workgroup-sized tiles per slice
    uint32_t tilesPerXcd;  // ceil(nTiles / 8)
    uint32_t outX, outY;   // output grid
    uint32_t tilesX;       // tiles per output row
    uint32_t tileWLog2;    // tile = 2^tileWLog2 x (256 >> tileWLog2) output cells, lanes row-major inside it
    uint32_t xcdRemap;     // 1: contiguous band of tiles per XCD
};

// blockIdx.x -> tile so that each XCD (blockIdx.x % 8 under round-robin dispatch) works on a
// contiguous band of the output.  Placement only affects speed, never results.
__device__ __forceinline__ bool tile_cell(const ApplyArgs& a, uint32_t& cell)
{
    const uint32_t b = blockIdx.x;
    const uint32_t tile = a.xcdRemap ? (b % kXcds) * a.tilesPerXcd + b / kXcds : b;
    if (tile >= a.nTiles) return false;
    const uint32_t tx = tile % a.tilesX, ty = tile / a.tilesX;
    const uint32_t x = (tx << a.tileWLog2) + (threadIdx.x & ((1u << a.tileWLog2) - 1));
    const uint32_t y = ty * (kBlock >> a.tileWLog2) + (threadIdx.x >> a.tileWLog2);
    cell = y * a.outX + x;
    return x < a.outX && y < a.outY;
}

// Addressing: buffer instructions.  A 128-bit descriptor (4 SGPRs) is built per z chunk from
// wave-uniform values; the per-lane part of an address is one 32-bit byte offset VGPR (plus an
// immediate), the slice index goes into the scalar offset.  No 64-bit per-lane address arithmetic,
// which keeps the register budget for loads in flight.  A chunk must span < 4 GiB (checked on the
// host: kernels with ZC > 1 are only launched when ZC slices fit).
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const float* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float ld(rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// outputs are written once and never re-read by this kernel: non-temporal (aux = 2)
__device__ __forceinline__ void st_stream(rsrc_t r, uint32_t voff, uint32_t soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 2);
}
__device__ __forceinline__ void st_plain(rsrc_t r, uint32_t voff, uint32_t soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

__device__ __forceinline__ void write_undefined(const ApplyArgs& a, uint32_t cell, uint32_t z0, uint32_t z1)
{
    const uint32_t outBytes = a.nOut * 4u;
    const float* o = a.out + (size_t)z0 * a.nOut;
    for (uint32_t z = z0; z < z1; ++z, o += a.nOut) st_stream(make_rsrc(o, outBytes), cell * 4u, 0, undefined_f());
}

template <int ZC>
__global__ void __launch_bounds__(kBlock) nearest_apply(ApplyArgs a, const uint32_t* __restrict__ pos)
{
    uint32_t cell;
    if (!tile_cell(a, cell)) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const uint32_t p = pos[cell];
    if (p == kInvalidPos) { write_undefined(a, cell, z0, z1); return; }
    const uint32_t pb = p * 4u, cb = cell * 4u;
    const uint32_t inBytes = (uint32_t)a.inLayer * 4u, outBytes = a.nOut * 4u;
    const float* src = a.in + (size_t)z0 * a.inLayer;
    const float* o = a.out + (size_t)z0 * a.nOut;
    uint32_t z = z0;
    for (; z + ZC <= z1; z += ZC) {
        const rsrc_t rs = make_rsrc(src, inBytes * ZC), ro = make_rsrc(o, outBytes * ZC);
        float v[ZC];
#pragma unroll
        for (int k = 0; k < ZC; ++k) v[k] = ld(rs, pb, inBytes * k);
#pragma unroll
        for (int k = 0; k < ZC; ++k) st_stream(ro, cb, outBytes * k, v[k]);
        src += (size_t)ZC * a.inLayer;
        o += (size_t)ZC * a.nOut;
    }
    for (; z < z1; ++z, src += a.inLayer, o += a.nOut)
        st_stream(make_rsrc(o, outBytes), cb, 0, ld(make_rsrc(src, inBytes), pb, 0));
}

// interior cell, src/interpolation.c:899-900
__device__ __forceinline__ float bilinear_point(float s00, float s01, float s10, float s11, float xf, float yf)
{
    return (1.f - yf) * ((1.f - xf) * s00 + xf * s01) + yf * ((1.f - xf) * s10 + xf * s11);
}

template <int ZC, bool NT = true>
__global__ void __launch_bounds__(kBlock) bilinear_apply(ApplyArgs a, const uint32_t* __restrict__ pos,
                                                         const float* __restrict__ xfrac, const float* __restrict__ yfrac)
{
    uint32_t cell;
    if (!tile_cell(a, cell)) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const uint32_t p = pos[cell];
    const float xf = xfrac[cell], yf = yfrac[cell];
    if (p == kInvalidPos) { write_undefined(a, cell, z0, z1); return; }
    const bool nnx = (__float_as_uint(xf) >> 31) != 0;
    const bool nny = (__float_as_uint(yf) >> 31) != 0;
    const uint32_t inBytes = (uint32_t)a.inLayer * 4u, outBytes = a.nOut * 4u;
    const float* src = a.in + (size_t)z0 * a.inLayer;
    const float* o = a.out + (size_t)z0 * a.nOut;
    const uint32_t pb = p * 4u, cb = cell * 4u;
    const uint32_t pb1 = pb + a.ix * 4u;  // the row below
    if (!(nnx || nny)) {
        uint32_t z = z0;
        for (; z + ZC <= z1; z += ZC) {
            const rsrc_t rs = make_rsrc(src, inBytes * ZC), ro = make_rsrc(o, outBytes * ZC);
            float s00[ZC], s01[ZC], s10[ZC], s11[ZC];
#pragma unroll
            for (int k = 0; k < ZC; ++k) {
                s00[k] = ld(rs, pb, inBytes * k);
                s01[k] = ld(rs, pb + 4u, inBytes * k);
                s10[k] = ld(rs, pb1, inBytes * k);
                s11[k] = ld(rs, pb1 + 4u, inBytes * k);
            }
#pragma unroll
            for (int k = 0; k < ZC; ++k) {
                const float r = bilinear_point(s00[k], s01[k], s10[k], s11[k], xf, yf);
                if (NT) st_stream(ro, cb, outBytes * k, r);
                else st_plain(ro, cb, outBytes * k, r);
            }
            src += (size_t)ZC * a.inLayer;
            o += (size_t)ZC * a.nOut;
        }
        for (; z < z1; ++z, src += a.inLayer, o += a.nOut) {
            const rsrc_t rs = make_rsrc(src, inBytes);
            st_stream(make_rsrc(o, outBytes), cb, 0,
                      bilinear_point(ld(rs, pb, 0), ld(rs, pb + 4u, 0), ld(rs, pb1, 0), ld(rs, pb1 + 4u, 0), xf, yf));
        }
    } else {
        // border branches of src/interpolation.c:903-948: a handful of cells on the rim of the domain
        for (uint32_t z = z0; z < z1; ++z, src += a.inLayer, o += a.nOut) {
            const rsrc_t rs = make_rsrc(src, inBytes);
            const float s00 = ld(rs, pb, 0);
            float r;
            if (nnx && nny) r = s00;                                       // :939-942
            else if (nny) r = (1.f - xf) * s00 + xf * ld(rs, pb + 4u, 0);  // :911
            else r = (1 - yf) * s00 + (yf * ld(rs, pb1, 0));               // :931
            st_stream(make_rsrc(o, outBytes), cb, 0, r);
        }
    }
}

// ONE slice per call: one time step of a variable without levels, the reference's call pattern for surface fields
// (src/CDMInterpolator.cc:251-259).  There is no slice loop to keep loads in flight across, and a lane with
// one output cell spends its life in two dependent round trips to memory (plan entry, then stencil).  Here a lane takes CELLS
// output cells of a 64 x (4 * CELLS) tile (rows y, y + 4, ...): the plan entries of all of them are loaded first, then all
// their stencil values -- four times the bytes in flight per lane; border and undefined cells are computed by selection
// (every form evaluated, src/interpolation.c:899-948), so that no lane leaves the common path.  STENCIL 1: nearest (:869-876).
template <int STENCIL, int CELLS>
__global__ void __launch_bounds__(kBlock) apply_few(ApplyArgs a, const uint32_t* __restrict__ pos, const float* __restrict__ xfrac,
                                                    const float* __restrict__ yfrac, uint32_t tilesX, uint32_t nTiles, uint32_t tilesPerXcd)
{
    const uint32_t b = blockIdx.x;
    const uint32_t tile = (b % kXcds) * tilesPerXcd + b / kXcds;  // a contiguous band of tile rows per XCD
    if (tile >= nTiles) return;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t x = tx * 64u + (threadIdx.x & 63u);
    const uint32_t y0 = ty * (4u * CELLS) + (threadIdx.x >> 6);
    uint32_t cb[CELLS], pb[CELLS], dxb[CELLS], dyb[CELLS];
    float xf[CELLS], yf[CELLS];
    bool undef[CELLS], nnx[CELLS], nny[CELLS];
#pragma unroll
    for (int k = 0; k < CELLS; ++k) {
        const uint32_t y = y0 + 4u * k;
        const bool mine = x < a.outX && y < a.outY;
        const uint32_t cell = y * a.outX + x;
        cb[k] = mine ? cell * 4u : 0xFFFFFFFFu;  // beyond the slice: the store is dropped
        const uint32_t p = mine ? pos[cell] : kInvalidPos;
        xf[k] = (mine && STENCIL == 2) ? xfrac[cell] : 0.f;
        yf[k] = (mine && STENCIL == 2) ? yfrac[cell] : 0.f;
        undef[k] = p == kInvalidPos;
        pb[k] = undef[k] ? 0u : p * 4u;
        nnx[k] = (__float_as_uint(xf[k]) >> 31) != 0;
        nny[k] = (__float_as_uint(yf[k]) >> 31) != 0;
        dxb[k] = nnx[k] ? 0u : 4u;           // a missing neighbour repeats the cell itself (its value is not used)
        dyb[k] = nny[k] ? 0u : a.ix * 4u;
    }
    const uint32_t inBytes = (uint32_t)a.inLayer * 4u, outBytes = a.nOut * 4u;
    for (uint32_t z = 0; z < a.nz; ++z) {
        const rsrc_t rs = make_rsrc(a.in + (size_t)z * a.inLayer, inBytes), ro = make_rsrc(a.out + (size_t)z * a.nOut, outBytes);
        float s00[CELLS], s01[CELLS], s10[CELLS], s11[CELLS];
#pragma unroll
        for (int k = 0; k < CELLS; ++k) {
            s00[k] = ld(rs, pb[k], 0);
            if (STENCIL == 2) {
                s01[k] = ld(rs, pb[k] + dxb[k], 0);
                s10[k] = ld(rs, pb[k] + dyb[k], 0);
                s11[k] = ld(rs, pb[k] + dxb[k] + dyb[k], 0);
            }
        }
#pragma unroll
        for (int k = 0; k < CELLS; ++k) {
            float r = s00[k];
            if (STENCIL == 2) {
                const float top = (1.f - xf[k]) * s00[k] + xf[k] * s01[k];   // :911 when nearest in y
                const float bot = (1.f - xf[k]) * s10[k] + xf[k] * s11[k];
                const float inter = (1.f - yf[k]) * top + yf[k] * bot;       // :899-900
                const float liny = (1 - yf[k]) * s00[k] + (yf[k] * s10[k]);  // :931
                r = nnx[k] ? (nny[k] ? s00[k] : liny) : (nny[k] ? top : inter);
            }
            st_stream(ro, cb[k], 0, undef[k] ? undefined_f() : r);
        }
    }
}

// Keys kernel a = -0.5: rows of M/2 (src/interpolation.c:962-968), weights XM / MY (:977-1000)
__device__ __forceinline__ void cubic_weights(double f, double w[4])
{
    const double M[4][4] = {{0.0, 1.0, 0.0, 0.0}, {-0.5, 0.0, 0.5, 0.0}, {1.0, -2.5, 2.0, -0.5}, {-0.5, 1.5, -1.5, 0.5}};
    double X[4];
    X[0] = 1;
    X[1] = f;
    X[2] = f * f;
    X[3] = X[2] * f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double s = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += X[j] * M[j][i];
        w[i] = s;
    }
}

// one 4x4 stencil: XMF[i] = sum_j XM[j] * F[j][i] (:1015), out += XMF[i] * MY[i] into the float (:1005,1019)
__device__ __forceinline__ float bicubic_point(const float f[4][4], const double XM[4], const double MY[4])
{
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double xmf = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) xmf += XM[j] * (double)f[i][j];
        acc = (float)((double)acc + xmf * MY[i]);
    }
    return acc;
}

template <int ZC>
__global__ void __launch_bounds__(kBlock) bicubic_apply(ApplyArgs a, const uint32_t* __restrict__ pos,
                                                        const double* __restrict__ xfd, const double* __restrict__ yfd)
{
    uint32_t cell;
    if (!tile_cell(a, cell)) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const uint32_t p = pos[cell];
    if (p == kInvalidPos) { write_undefined(a, cell, z0, z1); return; }
    double XM[4], MY[4];
    cubic_weights(xfd[cell], XM);
    cubic_weights(yfd[cell], MY);
    const uint32_t inBytes = (uint32_t)a.inLayer * 4u, outBytes = a.nOut * 4u;
    const float* src = a.in + (size_t)z0 * a.inLayer;
    const float* o = a.out + (size_t)z0 * a.nOut;
    const uint32_t cb = cell * 4u;
    uint32_t rowb[4];  // byte offsets of the four stencil rows
#pragma unroll
    for (int i = 0; i < 4; ++i) rowb[i] = (p + i * a.ix) * 4u;
    uint32_t z = z0;
    for (; z + ZC <= z1; z += ZC) {
        const rsrc_t rs = make_rsrc(src, inBytes * ZC), ro = make_rsrc(o, outBytes * ZC);
        float f[ZC][4][4];
#pragma unroll
        for (int k = 0; k < ZC; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) f[k][i][j] = ld(rs, rowb[i] + 4u * j, inBytes * k);
#pragma unroll
        for (int k = 0; k < ZC; ++k) st_stream(ro, cb, outBytes * k, bicubic_point(f[k], XM, MY));
        src += (size_t)ZC * a.inLayer;
        o += (size_t)ZC * a.nOut;
    }
    for (; z < z1; ++z, src += a.inLayer, o += a.nOut) {
        const rsrc_t rs = make_rsrc(src, inBytes);
        float f[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) f[i][j] = ld(rs, rowb[i] + 4u * j, 0);
        st_stream(make_rsrc(o, outBytes), cb, 0, bicubic_point(f, XM, MY));
    }
}

ApplyArgs make_args(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, dim3& grid)
{
    ApplyArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.nOut = (uint32_t)(plan.outX * plan.outY);
    a.ix = (uint32_t)plan.inX;
    a.inLayer = plan.inX * plan.inY;
    a.nz = (uint32_t)nz;
    a.outX = (uint32_t)plan.outX;
    a.outY = (uint32_t)plan.outY;
    // tile shape: 2^k x (256 >> k) output cells per workgroup
    uint32_t k = (uint32_t)tuning("TILEW_LOG2", 6);
    if (k > 8) k = 8;
    a.tileWLog2 = k;
    a.tilesX = (uint32_t)ceil_div(plan.outX, (size_t)1 << k);
    a.nTiles = a.tilesX * (uint32_t)ceil_div(plan.outY, (size_t)(kBlock >> k));
    a.xcdRemap = tuning("XCD", 1) ? 1 : 0;
    a.tilesPerXcd = (uint32_t)ceil_div(a.nTiles, kXcds);
    // slices per workgroup: long enough to amortise the plan read, short enough that the grid
    // still holds several waves of workgroups per CU
    uint32_t zpb = (uint32_t)tuning("ZPB", 0);
    if (zpb == 0) {
        const size_t wantBlocks = 256 * 8 * 4;
        size_t chunks = ceil_div(wantBlocks, (size_t)a.nTiles);
        if (chunks > nz) chunks = nz;
        if (chunks < 1) chunks = 1;
        zpb = (uint32_t)ceil_div(nz, chunks);
        const uint32_t zpbMax = (uint32_t)tuning("ZPB_MAX", 40);
        if (zpb > zpbMax) zpb = zpbMax;
    }
    if (zpb > nz) zpb = (uint32_t)nz;
    a.zPerBlock = zpb;
    const size_t chunks = ceil_div(nz, (size_t)zpb);
    FA_REQUIRE(chunks <= 65535, "too many z chunks for one launch");
    grid = dim3(a.tilesPerXcd * kXcds, (uint32_t)chunks, 1);
    return a;
}

// ---------------------------------------------------------------------------------------------------------------
// The same three stencils on a variable's STORED type: elements of T are read as they lie in the file
// (packed shorts, bytes), become float with the fill value as NaN in the register (data2InterpolationArray), and the
// result goes back to T with NaN as the fill value (interpolationArray2Data) -- per element the operations of the
// float kernels bracketed by those of convert.hip, so the bytes moved are sizeof(T) instead of 4 + the two passes.
struct TypedArgs {
    ApplyArgs g;          // geometry; g.in / g.out unused
    const void* in;
    void* out;
    float bad;            // the fill value narrowed to float (mifi_bad2nanf's parameter)
    int hasBad;
    double fillOut;       // the fill value as interpolationArray2Data receives it
};

template <typename T>
__device__ __forceinline__ T ld_raw(rsrc_t r, uint32_t voff, uint32_t soff)
{
    if (sizeof(T) == 1) return (T)__builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0);
    if (sizeof(T) == 2) return (T)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
    return (T)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
template <typename T>
__device__ __forceinline__ void st_raw(rsrc_t r, uint32_t voff, uint32_t soff, T v)
{
    if (sizeof(T) == 1) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, r, voff, soff, 2);
    else if (sizeof(T) == 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, voff, soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b32((unsigned int)v, r, voff, soff, 2);
}

// METHOD 0 nearest, 1 bilinear, 2 bicubic
template <int METHOD, typename T>
__global__ void __launch_bounds__(kBlock) typed_apply(TypedArgs t, const uint32_t* __restrict__ pos, const float* __restrict__ xfrac,
                                                      const float* __restrict__ yfrac, const double* __restrict__ xfd,
                                                      const double* __restrict__ yfd)
{
    const ApplyArgs& a = t.g;
    uint32_t cell;
    if (!tile_cell(a, cell)) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    constexpr uint32_t E = sizeof(T);
    const uint32_t inBytes = (uint32_t)a.inLayer * E, outBytes = a.nOut * E;
    const char* src = static_cast<const char*>(t.in) + (size_t)z0 * inBytes;
    char* o = static_cast<char*>(t.out) + (size_t)z0 * outBytes;
    const uint32_t cb = cell * E;
    const T fill = static_cast<T>(t.fillOut);  // ScaleValue's newFill_ (Utils.h:456)
    const bool hasBad = t.hasBad != 0;
    auto rsrc_in = [&](const char* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, inBytes, 0x00020000); };
    auto rsrc_out = [&](char* p) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, outBytes, 0x00020000); };
    auto get = [&](rsrc_t rs, uint32_t off) { return as_float_nan(ld_raw<T>(rs, off, 0), t.bad, hasBad); };
    const uint32_t p = pos[cell];
    if (p == kInvalidPos) {
        for (uint32_t z = z0; z < z1; ++z, o += outBytes) st_raw<T>(rsrc_out(o), cb, 0, fill);  // undefined -> NaN -> fill value
        return;
    }
    const uint32_t pb = p * E;
    constexpr int ZC = 8;  // slices whose loads are in flight together (one descriptor spans them: ZC * slice bytes < 4 GiB, checked on the host)
    auto rsrc_in_n = [&](const char* q) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(q), 0, inBytes * ZC, 0x00020000); };
    auto rsrc_out_n = [&](char* q) { return __builtin_amdgcn_make_buffer_rsrc(q, 0, outBytes * ZC, 0x00020000); };
    auto getk = [&](rsrc_t rs, uint32_t off, int k) { return as_float_nan(ld_raw<T>(rs, off, inBytes * k), t.bad, hasBad); };
    if (METHOD == 0) {
        uint32_t z = z0;
        for (; z + ZC <= z1; z += ZC, src += (size_t)ZC * inBytes, o += (size_t)ZC * outBytes) {
            const rsrc_t rs = rsrc_in_n(src), ro = rsrc_out_n(o);
            float v[ZC];
#pragma unroll
            for (int k = 0; k < ZC; ++k) v[k] = getk(rs, pb, k);
#pragma unroll
            for (int k = 0; k < ZC; ++k) st_raw<T>(ro, cb, outBytes * k, from_float_fill<T>(v[k], fill));
        }
        for (; z < z1; ++z, src += inBytes, o += outBytes)
            st_raw<T>(rsrc_out(o), cb, 0, from_float_fill<T>(get(rsrc_in(src), pb), fill));
    } else if (METHOD == 1) {
        const float xf = xfrac[cell], yf = yfrac[cell];
        const bool nnx = (__float_as_uint(xf) >> 31) != 0, nny = (__float_as_uint(yf) >> 31) != 0;
        const uint32_t pb1 = pb + a.ix * E;
        uint32_t z = z0;
        if (!(nnx || nny)) {
            for (; z + ZC <= z1; z += ZC, src += (size_t)ZC * inBytes, o += (size_t)ZC * outBytes) {
                const rsrc_t rs = rsrc_in_n(src), ro = rsrc_out_n(o);
                float s00[ZC], s01[ZC], s10[ZC], s11[ZC];
#pragma unroll
                for (int k = 0; k < ZC; ++k) {
                    s00[k] = getk(rs, pb, k);
                    s01[k] = getk(rs, pb + E, k);
                    s10[k] = getk(rs, pb1, k);
                    s11[k] = getk(rs, pb1 + E, k);
                }
#pragma unroll
                for (int k = 0; k < ZC; ++k)
                    st_raw<T>(ro, cb, outBytes * k, from_float_fill<T>(bilinear_point(s00[k], s01[k], s10[k], s11[k], xf, yf), fill));
            }
        }
        for (; z < z1; ++z, src += inBytes, o += outBytes) {
            const rsrc_t rs = rsrc_in(src);
            const float s00 = get(rs, pb);
            float r;
            if (!(nnx || nny)) r = bilinear_point(s00, get(rs, pb + E), get(rs, pb1), get(rs, pb1 + E), xf, yf);
            else if (nnx && nny) r = s00;                                   // :939-942
            else if (nny) r = (1.f - xf) * s00 + xf * get(rs, pb + E);      // :911
            else r = (1 - yf) * s00 + (yf * get(rs, pb1));                  // :931
            st_raw<T>(rsrc_out(o), cb, 0, from_float_fill<T>(r, fill));
        }
    } else {
        double XM[4], MY[4];
        cubic_weights(xfd[cell], XM);
        cubic_weights(yfd[cell], MY);
        for (uint32_t z = z0; z < z1; ++z, src += inBytes, o += outBytes) {
            const rsrc_t rs = rsrc_in(src);
            float f[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) f[i][j] = get(rs, pb + (i * a.ix + j) * E);
            st_raw<T>(rsrc_out(o), cb, 0, from_float_fill<T>(bicubic_point(f, XM, MY), fill));
        }
    }
}

template <typename T>
void launch_typed_t(const fimex_amd_regrid_plan& plan, const TypedArgs& t, dim3 grid, hipStream_t stream)
{
    switch (plan.kind) {
    case PlanKind::Nearest: typed_apply<0, T><<<grid, kBlock, 0, stream>>>(t, plan.pos.get(), nullptr, nullptr, nullptr, nullptr); break;
    case PlanKind::Bilinear: typed_apply<1, T><<<grid, kBlock, 0, stream>>>(t, plan.pos.get(), plan.xf.get(), plan.yf.get(), nullptr, nullptr); break;
    case PlanKind::Bicubic: typed_apply<2, T><<<grid, kBlock, 0, stream>>>(t, plan.pos.get(), nullptr, nullptr, plan.xfd.get(), plan.yfd.get()); break;
    default: throw Error("typed apply: not a backward plan");
    }
    FA_HIP(hipGetLastError());
}

}  // namespace

void build_backward_plan(fimex_amd_regrid_plan& plan, const double* d_px, const double* d_py, hipStream_t stream)
{
    const size_t n = plan.outX * plan.outY;
    FA_REQUIRE(n > 0 && n <= kMaxSliceCells, "output grid must have between 1 and 2^30-1 cells per slice");
    FA_REQUIRE(plan.inX > 0 && plan.inY > 0 && plan.inX * plan.inY <= kMaxSliceCells,
               "input grid must have between 1 and 2^30-1 cells per slice");
    DeviceArray<PlanCounters> counters(1);
    FA_HIP(hipMemsetAsync(counters.get(), 0, sizeof(PlanCounters), stream));
    const dim3 grid((uint32_t)ceil_div(n, kBlock));
    plan.pos.allocate(n);
    switch (plan.kind) {
    case PlanKind::Nearest:
        classify_nearest<<<grid, kBlock, 0, stream>>>(d_px, d_py, (uint32_t)n, (int64_t)plan.inX, (int64_t)plan.inY,
                                                      plan.pos.get(), counters.get());
        plan.info.planBytes = plan.pos.bytes();
        break;
    case PlanKind::Bilinear:
        plan.xf.allocate(n);
        plan.yf.allocate(n);
        classify_bilinear<<<grid, kBlock, 0, stream>>>(d_px, d_py, (uint32_t)n, (int64_t)plan.inX, (int64_t)plan.inY,
                                                       plan.pos.get(), plan.xf.get(), plan.yf.get(), counters.get());
        plan.info.planBytes = plan.pos.bytes() + plan.xf.bytes() + plan.yf.bytes();
        break;
    case PlanKind::Bicubic:
        plan.xfd.allocate(n);
        plan.yfd.allocate(n);
        classify_bicubic<<<grid, kBlock, 0, stream>>>(d_px, d_py, (uint32_t)n, (int64_t)plan.inX, (int64_t)plan.inY,
                                                      plan.pos.get(), plan.xfd.get(), plan.yfd.get(), counters.get());
        plan.info.planBytes = plan.pos.bytes() + plan.xfd.bytes() + plan.yfd.bytes();
        break;
    default:
        throw Error("build_backward_plan: not a backward plan");
    }
    FA_HIP(hipGetLastError());
    PlanCounters h{};
    FA_HIP(hipMemcpyAsync(&h, counters.get(), sizeof(h), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    plan.info.undefinedCells = (size_t)h.undefined;
    plan.info.borderCells = (size_t)h.border;
    // bilinear: also the LDS-staged form (staged.hip); plans whose tiles do not fit keep the gather kernel only
    if (tuning("STAGED", 1) != 0 && (plan.kind != PlanKind::Nearest || tuning("STAGED_NEAREST", 1) != 0) &&
        build_staged_plan(plan, d_px, d_py, stream)) {
        const auto& s = plan.staged;  // the staged kernel reads LDS offsets instead of pos, plus the tile tables
        plan.info.planBytes = plan.info.planBytes - plan.pos.bytes() + s.ldsA.bytes() + s.ldsB.bytes() + s.tileHdr.bytes() +
                              (size_t)s.nTiles * 2 * 4 * 48;
        plan.info.stagedCells = s.stagedCells;
        plan.info.tileW = s.tileW;
        plan.info.tileH = s.tileH;
    }
    // the second staged form (staged2.hip) serves float slices; the first one stays for 1- and 2-byte stored types
    // (bicubic keeps the first form where it exists -- 32 x 16 tiles at four workgroups per CU suit its arithmetic -- and takes
    // the second one for the source widths the first cannot stage, inX % 4 != 0)
    const int second = tuning("STAGED2", 1);
    const bool wantSecond = second >= 2 || (second == 1 && (plan.kind != PlanKind::Bicubic || !plan.staged.valid || plan.bicubicFast));
    if (tuning("STAGED", 1) != 0 && wantSecond && build_staged2_plan(plan, d_px, d_py, stream)) {
        const auto& s = plan.staged2;
        const size_t perCell = plan.kind == PlanKind::Nearest ? 0 : (plan.kind == PlanKind::Bilinear ? plan.xf.bytes() + plan.yf.bytes()
                                                                                                       : plan.xfd.bytes() + plan.yfd.bytes());
        auto bytes_of = [&](const Staged2Plan& q) {
            return perCell + q.ldsA.bytes() + q.ldsB.bytes() + q.totalChunks * 4 + (size_t)q.nTiles * sizeof(StagedTile) + q.order.bytes();
        };
        plan.planBytesShape[0] = bytes_of(s);
        plan.planBytesShape[1] = plan.staged2Alt.valid ? bytes_of(plan.staged2Alt) : 0;
        plan.info.planBytes = plan.planBytesShape[0];
        plan.info.stagedCells = s.stagedCells;
        plan.info.tileW = s.tileWMax;
        plan.info.tileH = s.tileH;
    }
}

void launch_backward_apply(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream)
{
    if (nz == 0) return;
    FA_REQUIRE(nz <= 0xFFFFFFFFu, "too many slices");
    // the staged kernel pays a per-tile set-up (row table, chunk list) that only amortises over a few slices
    // (a bicubic plan in the reference's arithmetic that holds both forms runs the first one)
    const bool second = plan.staged2.valid && (plan.kind != PlanKind::Bicubic || plan.bicubicFast || !plan.staged.valid || tuning("STAGED2", 1) >= 2);
    if (second && tuning("STAGED", 1) != 0 && tuning("STAGED2", 1) != 0 && nz >= staged_min_nz()) {
        launch_staged2_apply(plan, d_in, nz, d_out, stream);
        return;
    }
    if (plan.staged.valid && tuning("STAGED", 1) != 0 && nz >= staged_min_nz()) {
        launch_staged_apply(plan, d_in, nz, d_out, stream);
        return;
    }
    launch_backward_gather(plan, d_in, nz, d_out, stream);
}

// The per-lane gather kernels on any backward plan (short batches, plans without a staged form, and the cross-check of
// fimex_amd_regrid_apply_gather_device).
void launch_backward_gather(const fimex_amd_regrid_plan& plan, const float* d_in, size_t nz, float* d_out, hipStream_t stream)
{
    if (nz == 0) return;
    FA_REQUIRE(nz <= 0xFFFFFFFFu, "too many slices");
    dim3 grid;
    const ApplyArgs a = make_args(plan, d_in, nz, d_out, grid);
    if (nz == 1 && plan.kind != PlanKind::Bicubic && tuning("FEW", 1) != 0) {
        // one slice: several output cells per lane (apply_few).  Measured cold, a different slice pair every call
        // (profiles/r03_sweep_few_cells*.log): 2 / 4 / 8 cells per lane 24.4 / 25.5 / 28.0 us against 29.4 us for one cell per lane;
        // with two or three slices the kernels below, which keep the slices of a cell in flight together, are ahead again
        // (40.4 against 42 us, 52 against 59 us).
        const int cells = tuning("FEW_CELLS", 2);
        auto go = [&](auto cellsTag) {
            constexpr int kCells = decltype(cellsTag)::value;
            const uint32_t tilesX = (uint32_t)ceil_div(plan.outX, (size_t)64), tilesY = (uint32_t)ceil_div(plan.outY, (size_t)(4 * kCells));
            const uint32_t nTiles = tilesX * tilesY, perXcd = (uint32_t)ceil_div(nTiles, kXcds);
            const dim3 g(perXcd * kXcds);
            if (plan.kind == PlanKind::Nearest) apply_few<1, kCells><<<g, kBlock, 0, stream>>>(a, plan.pos.get(), nullptr, nullptr, tilesX, nTiles, perXcd);
            else apply_few<2, kCells><<<g, kBlock, 0, stream>>>(a, plan.pos.get(), plan.xf.get(), plan.yf.get(), tilesX, nTiles, perXcd);
        };
        if (cells == 2) go(std::integral_constant<int, 2>());
        else if (cells == 8) go(std::integral_constant<int, 8>());
        else go(std::integral_constant<int, 4>());
        FA_HIP(hipGetLastError());
        return;
    }
    // a chunk of ZC slices is addressed through one 32-bit buffer range: fall back to ZC = 1 for huge slices
    const size_t sliceBytes = 4 * (a.inLayer > a.nOut ? a.inLayer : (size_t)a.nOut);
    auto fits = [&](size_t zc) { return sliceBytes * zc <= 0xFFFFFFFFull; };
    switch (plan.kind) {
    case PlanKind::Nearest:
        if (fits(16)) nearest_apply<16><<<grid, kBlock, 0, stream>>>(a, plan.pos.get());
        else nearest_apply<1><<<grid, kBlock, 0, stream>>>(a, plan.pos.get());
        break;
    case PlanKind::Bilinear: {
        const int zc = tuning("BILINEAR_ZC", 8);
        const bool nt = tuning("NT", 1) != 0;
        const uint32_t* pp = plan.pos.get();
        const float *xf = plan.xf.get(), *yf = plan.yf.get();
        if (zc >= 16 && fits(16)) {
            if (nt) bilinear_apply<16, true><<<grid, kBlock, 0, stream>>>(a, pp, xf, yf);
            else bilinear_apply<16, false><<<grid, kBlock, 0, stream>>>(a, pp, xf, yf);
        } else if (zc >= 8 && fits(8)) {
            if (nt) bilinear_apply<8, true><<<grid, kBlock, 0, stream>>>(a, pp, xf, yf);
            else bilinear_apply<8, false><<<grid, kBlock, 0, stream>>>(a, pp, xf, yf);
        } else if (zc >= 4 && fits(4)) {
            if (nt) bilinear_apply<4, true><<<grid, kBlock, 0, stream>>>(a, pp, xf, yf);
            else bilinear_apply<4, false><<<grid, kBlock, 0, stream>>>(a, pp, xf, yf);
        } else if (zc >= 2 && fits(2)) {
            bilinear_apply<2, true><<<grid, kBlock, 0, stream>>>(a, pp, xf, yf);
        } else {
            bilinear_apply<1, true><<<grid, kBlock, 0, stream>>>(a, pp, xf, yf);
        }
        break;
    }
    case PlanKind::Bicubic:
        if (fits(2)) bicubic_apply<2><<<grid, kBlock, 0, stream>>>(a, plan.pos.get(), plan.xfd.get(), plan.yfd.get());
        else bicubic_apply<1><<<grid, kBlock, 0, stream>>>(a, plan.pos.get(), plan.xfd.get(), plan.yfd.get());
        break;
    default:
        throw Error("launch_backward_apply: not a backward plan");
    }
    FA_HIP(hipGetLastError());
}

// Fused stored-type apply for 1- and 2-byte integer types and int32 (backward plans); returns false for everything else
// (the caller then runs the three-pass sequence: to float, regrid, from float).
bool launch_typed_apply(const fimex_amd_regrid_plan& plan, const void* d_in, int cdmType, size_t nz, double badValue, void* d_out,
                        hipStream_t stream)
{
    // forward plans with long buckets: the LDS-staged forward kernel on the stored type (forward_tiled.hip)
    if (plan.kind == PlanKind::Forward) return launch_forward_tiled_typed(plan, d_in, cdmType, nz, badValue, d_out, stream);
    // 1- and 2-byte types through the LDS-staged kernels (the same dispatch rule as for floats): the second staged form
    // (staged2.hip, nearest and bilinear), else the first one
    if (plan.kind != PlanKind::Forward && tuning("TYPED_FUSED", 1) != 0 && tuning("STAGED", 1) != 0 && tuning("TYPED_STAGED", 1) != 0 &&
        tuning("TYPED_STAGED2", 1) != 0 && nz >= staged_min_nz() && launch_staged2_apply_typed(plan, d_in, cdmType, nz, badValue, d_out, stream))
        return true;
    if (plan.kind != PlanKind::Forward && tuning("TYPED_FUSED", 1) != 0 && plan.staged.valid && tuning("STAGED", 1) != 0 && tuning("TYPED_STAGED", 1) != 0 &&
        nz >= staged_min_nz() && (plan.kind != PlanKind::Nearest || tuning("STAGED_NEAREST", 1) != 0) &&
        launch_staged_apply_typed(plan, d_in, cdmType, nz, badValue, d_out, stream))
        return true;
    // bicubic: the LDS-staged float kernel between two conversion passes beats a 16-load gather on the stored type
    if (plan.kind == PlanKind::Forward || (plan.kind == PlanKind::Bicubic && tuning("TYPED_FUSED", 1) < 2) || tuning("TYPED_FUSED", 1) == 0) return false;
    if (!(cdmType == FIMEX_AMD_CDM_CHAR || cdmType == FIMEX_AMD_CDM_UCHAR || cdmType == FIMEX_AMD_CDM_SHORT ||
          cdmType == FIMEX_AMD_CDM_USHORT || cdmType == FIMEX_AMD_CDM_INT))
        return false;
    if (nz == 0) return true;
    FA_REQUIRE(nz <= 0xFFFFFFFFu, "too many slices");
    if (8 * 4 * std::max(plan.inX * plan.inY, plan.outX * plan.outY) > 0xFFFFFFFFull) return false;  // 8 slices behind one descriptor
    TypedArgs t{};
    dim3 grid;
    t.g = make_args(plan, nullptr, nz, nullptr, grid);
    t.in = d_in;
    t.out = d_out;
    t.bad = (float)badValue;
    t.hasBad = !(t.bad != t.bad);
    t.fillOut = badValue;
    switch (cdmType) {
    case FIMEX_AMD_CDM_CHAR: launch_typed_t<signed char>(plan, t, grid, stream); break;
    case FIMEX_AMD_CDM_UCHAR: launch_typed_t<unsigned char>(plan, t, grid, stream); break;
    case FIMEX_AMD_CDM_SHORT: launch_typed_t<short>(plan, t, grid, stream); break;
    case FIMEX_AMD_CDM_USHORT: launch_typed_t<unsigned short>(plan, t, grid, stream); break;
    default: launch_typed_t<int>(plan, t, grid, stream); break;
    }
    return true;
}

}  // namespace fimex_amd
