// Device helpers shared by the LDS-staged regrid kernels (staged.hip: one 256-thread workgroup per uniform tile;
// staged2.hip: larger workgroups, tiles of varying width, slice ring of varying depth).
#pragma once

#include "plan.hpp"

namespace fimex_amd {
namespace {

__device__ __forceinline__ float undefined_f() { return __uint_as_float(0x7fc00000u); }

__device__ __forceinline__ bool usable(double x, double y)
{
    const double lim = 1073741824.0;
    return (fabs(x) < lim) && (fabs(y) < lim);
}

constexpr int kMaxRows = 160;  // source rows one tile may span

// which source cells one output cell reads: columns xa..xb of rows ya..yb (inclusive)
struct CellNeed {
    bool valid;
    int64_t xa, xb, ya, yb;
};

// STENCIL 1: nearest (src/interpolation.c:862-879); 2: bilinear incl. its border branches (:883-954); 4: bicubic (:970-976)
template <int STENCIL>
__device__ CellNeed classify(double x, double y, int64_t ix, int64_t iy)
{
    CellNeed c{};
    c.valid = false;
    if (!usable(x, y)) return c;
    if (STENCIL == 1) {  // nearest: lround half away from zero (src/interpolation.c:864-868)
        const int64_t rx = (int64_t)round(x), ry = (int64_t)round(y);
        if (rx >= 0 && rx < ix && ry >= 0 && ry < iy) { c.valid = true; c.xa = c.xb = rx; c.ya = c.yb = ry; }
        return c;
    }
    const int64_t x0 = (int64_t)floor(x), y0 = (int64_t)floor(y);
    if (STENCIL == 4) {
        if ((1 <= x0) && (x0 + 2 < ix) && (1 <= y0) && (y0 + 2 < iy)) {
            c.valid = true; c.xa = x0 - 1; c.xb = x0 + 2; c.ya = y0 - 1; c.yb = y0 + 2;
        }
        return c;
    }
    const bool xlin = (0 <= x0) && (x0 + 1 < ix);
    const bool ylin = (0 <= y0) && (y0 + 1 < iy);
    if (xlin && ylin) {
        c.valid = true; c.xa = x0; c.xb = x0 + 1; c.ya = y0; c.yb = y0 + 1;
    } else if (xlin) {
        const int64_t ry = (int64_t)round(y);
        if (0 <= ry && ry < iy) { c.valid = true; c.xa = x0; c.xb = x0 + 1; c.ya = c.yb = ry; }
    } else {
        const int64_t rx = (int64_t)round(x);
        if (0 <= rx && rx < ix) {
            if (ylin) {
                c.valid = true; c.xa = c.xb = rx; c.ya = y0; c.yb = y0 + 1;
            } else {
                const int64_t ry = (int64_t)round(y);
                if (0 <= ry && ry < iy) { c.valid = true; c.xa = c.xb = rx; c.ya = c.yb = ry; }
            }
        }
    }
    return c;
}

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// One LDS-DMA wave instruction: 64 lanes x 16 bytes from per-lane buffer offsets to ldsBase + lane * 16.
// (The builtin exists only in the device pass; the host pass of hipcc parses kernel bodies too.)
__device__ __forceinline__ void dma16(rsrc_t rs, float* ldsBase, uint32_t voff, uint32_t aux = 0)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using lds_ptr = __attribute__((address_space(3))) void*;
    switch (aux) {  // cache policy, wave-uniform (tuning knob LOAD_AUX; the default policy measured best)
    case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)ldsBase, 16, voff, 0, 0, 1); break;
    case 16: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)ldsBase, 16, voff, 0, 0, 16); break;
    case 17: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)ldsBase, 16, voff, 0, 0, 17); break;
    case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)ldsBase, 16, voff, 0, 0, 2); break;
    default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)ldsBase, 16, voff, 0, 0, 0); break;
    }
#else
    (void)rs; (void)ldsBase; (void)voff; (void)aux;
#endif
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

// s_waitcnt on vmcnt only (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "no wait")
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
    asm volatile("" ::: "memory");
}


}  // namespace
}  // namespace fimex_amd
