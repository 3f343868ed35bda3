// Calibration of the access STRUCTURE of the LDS-staged regrid launch on gfx950 (round 2): what bounds a kernel that
// streams tile row segments HBM -> LDS by LDS-DMA and writes a quarter of that volume back, on the geometry of the
// headline launch (200 slices of 4000x3000 f32 -> 2000x2000), with a synthetic aligned plan (2 source columns and 1.5
// source rows per output cell, no shear) so that the pattern is the only variable.  Not product code: no plan, no parity.
//
//   tile<NT,PER,UN,NBUF,PRIV>: the staged kernel's loop.  NT threads share a tile of TW x TH outputs (TW*TH = NT*PER);
//       per slice UN DMA instructions per lane into an LDS ring of NBUF slots, 2x2 stencil reads from LDS, PER stores.
//       PRIV: every wave owns a tile of its own (NT = 64 in the geometry, 4 waves per workgroup, no barrier at all).
//   flags: 1 no loads, 2 no stores, 4 linear source addresses (a tile's chunks contiguous), 8 plain (not nt) stores,
//          16 nt loads, 32 XCD-contiguous tile order, 64 output rows dealt so that a workgroup walks tiles along x.
//   sum3: out = a + b + c with 16-byte global loads and stores: the 3:1 read/write stream at its simplest (ceiling).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int IX = 4000, IY = 3000, OX = 2000, OY = 2000;

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ void dma16(rsrc_t rs, float* ldsBase, uint32_t voff)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using lds_ptr = __attribute__((address_space(3))) void*;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)ldsBase, 16, voff, 0, 0, AUX);
#endif
}
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
    asm volatile("" ::: "memory");
}

struct Args {
    const float* in;
    float* out;
    uint32_t nz, zpb;
    uint32_t tw, th;      // outputs per tile
    uint32_t tilesX, nTiles;
    uint32_t flags;
    // flags & 2048: the footprint of a tile whose target is TILTED against the source rows (the benchmark's rotated pole: ~0.08
    // source rows per output column): a parallelogram of more, shorter row segments.  rel[c] = (row << 20 | first cell) of chunk c
    // relative to the tile's corner, the tile to the right continues shearRows further down.
    const uint32_t* rel;
    uint32_t relCount, shearRows;
    // flags & 4096: every workgroup loads a per-output plan entry (4 + 8 bytes) like the product kernel does per (tile, z chunk)
    const uint32_t* planL;  // zeros
    const float* planW;
};

template <int NT, int PER, int UN, int NBUF, bool PRIV>
__global__ void __launch_bounds__(PRIV ? 256 : NT) tile(Args a)
{
    constexpr int WG = PRIV ? 256 : NT;
    constexpr uint32_t kSlot = UN * NT * 4;  // floats per slot (of one tile owner)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t wave = threadIdx.x / 64;
    const uint32_t tid = PRIV ? threadIdx.x % 64 : threadIdx.x;
    float* mine = PRIV ? smem + wave * NBUF * kSlot : smem;

    uint32_t b = blockIdx.x;
    const uint32_t owners = PRIV ? 4 : 1;
    uint32_t tile = PRIV ? b * 4 + wave : b;
    if (a.flags & 32) {  // XCD-contiguous: workgroup b runs on XCD b % 8; give each XCD a contiguous range of tiles
        const uint32_t per = (a.nTiles / owners + 7) / 8;
        const uint32_t t = (b % 8) * per + b / 8;
        tile = PRIV ? t * 4 + wave : t;
    }
    if (tile >= a.nTiles) return;
    const uint32_t tx = tile % a.tilesX, ty = tile / a.tilesX;
    const uint32_t nrows = a.th * 3 / 2 + 1, cpr = a.tw / 2 + 1;
    const bool sheared = (a.flags & 2048) != 0;
    const uint32_t total = sheared ? a.relCount : nrows * cpr;
    const uint32_t r0 = ty * (a.th * 3 / 2) + (sheared ? tx * a.shearRows : 0u), c0 = tx * a.tw * 2;

    uint32_t gOff[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const uint32_t c = tid + j * NT;
        gOff[j] = 0xFFFFFFFFu;
        if (c < total) {
            if (sheared) { const uint32_t q = a.rel[c]; gOff[j] = (((r0 + (q >> 20)) % IY) * IX + c0 + (q & 0xFFFFFu)) * 4u; }
            else if (a.flags & 4) gOff[j] = (tile * total + c) * 16u;
            else gOff[j] = ((r0 + c / cpr) * IX + c0 + (c % cpr) * 4u) * 4u;
        }
    }
    const uint32_t rowsPerPass = NT / a.tw;
    const bool quad = (a.flags & 128) != 0;  // lane owns 4 adjacent cells of one row (PER == 4): one 16-byte store
    uint32_t cellOff[PER], ldsOff[PER];
    float w0[PER], w1[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        uint32_t lx = tid % a.tw, ly = tid / a.tw + k * rowsPerPass;
        if (quad) { lx = 4 * (tid % (a.tw / 4)) + k; ly = tid / (a.tw / 4); }
        const uint32_t x = tx * a.tw + lx, y = ty * a.th + ly;
        cellOff[k] = (x < OX && y < OY) ? (y * OX + x) * 4u : 0xFFFFFFFFu;
        if (a.flags & 256) cellOff[k] = (tile * NT * PER + ly * a.tw + lx) * 4u;  // tile-contiguous output (upper bound for write locality)
        ldsOff[k] = ((ly * 3 / 2) * cpr * 4 + 2 * lx) * 4u;  // bytes
        w0[k] = 0.25f + 0.001f * (float)lx;
        w1[k] = 0.25f + 0.002f * (float)ly;
        if ((a.flags & 4096) && x < OX && y < OY) {
            const uint32_t cell = y * OX + x;
            ldsOff[k] += a.planL[cell];
            w0[k] = a.planW[cell];
            w1[k] = a.planW[OX * OY + cell];
        }
    }
    const uint32_t z0 = blockIdx.y * a.zpb, z1 = min(a.nz, z0 + a.zpb);
    const uint32_t inBytes = IX * IY * 4u, outBytes = OX * OY * 4u;
    const uint32_t waveBase = PRIV ? 0 : wave * 64;
    auto dma = [&](float* dst, uint32_t z) {
        const rsrc_t rs = make_rsrc(a.in + (size_t)z * IX * IY, (a.flags & 1) ? 0u : ((a.flags & 4) ? inBytes + (8u << 20) : inBytes));
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            if (a.flags & 16) dma16<2>(rs, dst + (waveBase + j * NT) * 4, gOff[j]);
            else dma16<0>(rs, dst + (waveBase + j * NT) * 4, gOff[j]);
        }
    };
    auto sync = [&]() {
        if (!PRIV) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (z0 + i < z1) dma(mine + i * kSlot, z0 + i);
    if (NBUF > 1) { wait_vmcnt<(NBUF - 2) * UN>(); sync(); }
    uint32_t slot = 0;
    for (uint32_t z = z0; z < z1; ++z) {
        const bool more = z + (NBUF - 1) < z1;
        if (NBUF == 1) { dma(mine, z); wait_vmcnt<0>(); sync(); }
        else if (more) dma(mine + ((slot + NBUF - 1) % NBUF) * kSlot, z + (NBUF - 1));
        const char* cur = reinterpret_cast<const char*>(mine + slot * kSlot);
        const rsrc_t ro = make_rsrc(a.out + (size_t)z * OX * OY, (a.flags & 2) ? 0u : outBytes);
        float s00[PER], s01[PER], s10[PER], s11[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float* pa = reinterpret_cast<const float*>(cur + ldsOff[k]);
            const float* pb = reinterpret_cast<const float*>(cur + ldsOff[k] + cpr * 16);
            s00[k] = pa[0]; s01[k] = pa[1]; s10[k] = pb[0]; s11[k] = pb[1];
        }
        float rr[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float top = (1.f - w0[k]) * s00[k] + w0[k] * s01[k];
            const float bot = (1.f - w0[k]) * s10[k] + w0[k] * s11[k];
            rr[k] = (1.f - w1[k]) * top + w1[k] * bot;
        }
        if (a.flags & 1536) {  // extra arithmetic per output (512: 32 dependent FMAs, 1024: 96): is the loop's ALU time hidden?
            const int n = ((a.flags & 512) ? 32 : 0) + ((a.flags & 1024) ? 96 : 0);
            for (int q = 0; q < n; ++q) {
#pragma unroll
                for (int k = 0; k < PER; ++k) rr[k] = __builtin_fmaf(rr[k], 1.0000001f, w1[k]);
            }
        }
        if (quad && PER == 4) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const u4 v = {__float_as_uint(rr[0]), __float_as_uint(rr[1]), __float_as_uint(rr[2]), __float_as_uint(rr[3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, ro, cellOff[0], 0, 2);
            if (NBUF == 1) { }
            else if (more) wait_vmcnt<(NBUF - 2) * UN + (NBUF - 1) * 1>();
            else wait_vmcnt<1>();
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                if (a.flags & 8) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rr[k]), ro, cellOff[k], 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rr[k]), ro, cellOff[k], 0, 2);
            }
            if (NBUF == 1) { /* next iteration's barrier */ }
            else if (more) wait_vmcnt<(NBUF - 2) * UN + (NBUF - 1) * PER>();
            else wait_vmcnt<PER>();  // conservative at the tail: everything but this slice's stores
        }
        sync();
        slot = (slot + 1 == NBUF) ? 0 : slot + 1;
    }
}


// Results of SB consecutive slices stay in registers and are stored in one burst (SB * PER stores per lane): does the memory
// system reward longer write bursts per CU?
template <int NT, int PER, int UN, int NBUF, int SB>
__global__ void __launch_bounds__(NT) tile_sb(Args a)
{
    constexpr uint32_t kSlot = UN * NT * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const uint32_t wave = threadIdx.x / 64;
    const uint32_t tid = threadIdx.x;
    uint32_t tile = blockIdx.x;
    if (tile >= a.nTiles) return;
    const uint32_t tx = tile % a.tilesX, ty = tile / a.tilesX;
    const uint32_t nrows = a.th * 3 / 2 + 1, cpr = a.tw / 2 + 1;
    const uint32_t total = nrows * cpr;
    const uint32_t r0 = ty * (a.th * 3 / 2), c0 = tx * a.tw * 2;
    uint32_t gOff[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const uint32_t c = tid + j * NT;
        gOff[j] = (c < total) ? ((r0 + c / cpr) * IX + c0 + (c % cpr) * 4u) * 4u : 0xFFFFFFFFu;
    }
    const uint32_t rowsPerPass = NT / a.tw;
    uint32_t cellOff[PER], ldsOff[PER];
    float w0[PER], w1[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t lx = tid % a.tw, ly = tid / a.tw + k * rowsPerPass;
        const uint32_t x = tx * a.tw + lx, y = ty * a.th + ly;
        cellOff[k] = (x < OX && y < OY) ? (y * OX + x) * 4u : 0xFFFFFFFFu;
        ldsOff[k] = ((ly * 3 / 2) * cpr * 4 + 2 * lx) * 4u;
        w0[k] = 0.25f + 0.001f * (float)lx;
        w1[k] = 0.25f + 0.002f * (float)ly;
    }
    const uint32_t z0 = blockIdx.y * a.zpb, z1 = min(a.nz, z0 + a.zpb);  // zpb is a multiple of SB
    const uint32_t inBytes = IX * IY * 4u, outBytes = OX * OY * 4u;
    const uint32_t waveBase = wave * 64;
    auto dma = [&](float* dst, uint32_t z) {
        const rsrc_t rs = make_rsrc(a.in + (size_t)z * IX * IY, inBytes);
#pragma unroll
        for (int j = 0; j < UN; ++j) dma16<0>(rs, dst + (waveBase + j * NT) * 4, gOff[j]);
    };
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (z0 + i < z1) dma(smem + i * kSlot, z0 + i);
    wait_vmcnt<(NBUF - 2) * UN>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    uint32_t slot = 0;
    for (uint32_t zb = z0; zb < z1; zb += SB) {
        float rr[SB][PER];
#pragma unroll
        for (int sidx = 0; sidx < SB; ++sidx) {
            const uint32_t z = zb + sidx;
            const bool more = z + (NBUF - 1) < z1;
            if (more) dma(smem + ((slot + NBUF - 1) % NBUF) * kSlot, z + (NBUF - 1));
            const char* cur = reinterpret_cast<const char*>(smem + slot * kSlot);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const float* pa = reinterpret_cast<const float*>(cur + ldsOff[k]);
                const float* pb = reinterpret_cast<const float*>(cur + ldsOff[k] + cpr * 16);
                const float top = (1.f - w0[k]) * pa[0] + w0[k] * pa[1];
                const float bot = (1.f - w0[k]) * pb[0] + w0[k] * pb[1];
                rr[sidx][k] = (1.f - w1[k]) * top + w1[k] * bot;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (sidx == SB - 1) {  // the burst: SB * PER stores
#pragma unroll
                for (int q = 0; q < SB; ++q) {
                    const rsrc_t ro = make_rsrc(a.out + (size_t)(zb + q) * OX * OY, outBytes);
#pragma unroll
                    for (int k = 0; k < PER; ++k) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rr[q][k]), ro, cellOff[k], 0, 2);
                }
            }
            // slice z + 1 landed: younger are the DMAs of z + 2 .. z + NBUF - 1 and (after a burst) its stores; stores issued
            // before DMA(z + 1) are older and covered.  Conservative: count only what was certainly issued after DMA(z + 1).
            if (more) {
                if (sidx == SB - 1) wait_vmcnt<((NBUF - 2) * UN + SB * PER < 63 ? (NBUF - 2) * UN + SB * PER : 63)>();
                else wait_vmcnt<(NBUF - 2) * UN>();
            } else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            slot = (slot + 1 == NBUF) ? 0 : slot + 1;
        }
    }
}

template <int NT, int PER, int UN, int NBUF, int SB>
static void run_sb(const float* in, float* out, uint32_t nz, uint32_t tw, uint32_t zpb);

// loader / consumer split: waves 0..L-1 only issue DMA (ring of NBUF slots, run ahead), the others only read LDS and store.
// Hand-off through LDS counters: full[slot] counts loader waves done with a slot generation, free[slot] consumer waves.
template <int NT, int PER, int UN, int NBUF, int LOADERS>
__global__ void __launch_bounds__(NT + LOADERS * 64) tile_split(Args a)
{
    constexpr uint32_t kSlot = UN * NT * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ volatile uint32_t full[NBUF], freed[NBUF];
    const uint32_t wave = threadIdx.x / 64;
    const bool loader = wave < LOADERS;
    if (threadIdx.x < NBUF) { full[threadIdx.x] = 0; freed[threadIdx.x] = 0; }
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    if (tile >= a.nTiles) return;
    const uint32_t tx = tile % a.tilesX, ty = tile / a.tilesX;
    const uint32_t nrows = a.th * 3 / 2 + 1, cpr = a.tw / 2 + 1;
    const uint32_t total = nrows * cpr;
    const uint32_t r0 = ty * (a.th * 3 / 2), c0 = tx * a.tw * 2;
    const uint32_t z0 = blockIdx.y * a.zpb, z1 = min(a.nz, z0 + a.zpb);
    const uint32_t inBytes = IX * IY * 4u, outBytes = OX * OY * 4u;
    constexpr int CONSUMERS = NT / 64;
    if (loader) {
        // loader wave l issues pieces c = lane + 64 * (l + LOADERS * j): UNL = UN * NT / (64 * LOADERS) instructions per slice
        constexpr int UNL = UN * NT / (64 * LOADERS);
        constexpr int K = (NBUF - 2) < 48 / UNL ? (NBUF - 2) : 48 / UNL;  // slices a loader wave keeps in flight (vmcnt < 64)
        const uint32_t lane = threadIdx.x % 64;
        uint32_t gOff[UNL];
#pragma unroll
        for (int j = 0; j < UNL; ++j) {
            const uint32_t c = lane + 64 * (wave + LOADERS * j);
            gOff[j] = (c < total) ? ((r0 + c / cpr) * IX + c0 + (c % cpr) * 4u) * 4u : 0xFFFFFFFFu;
        }
        uint32_t slot = 0, gen = 0;
        for (uint32_t z = z0; z < z1; ++z) {
            // wait until every consumer wave has released this slot's previous generation
            if (gen > 0) while (freed[slot] < gen * CONSUMERS) __builtin_amdgcn_s_sleep(1);
            const rsrc_t rs = make_rsrc(a.in + (size_t)z * IX * IY, (a.flags & 1) ? 0u : inBytes);
            float* dst = smem + slot * kSlot;
#pragma unroll
            for (int j = 0; j < UNL; ++j) dma16<0>(rs, dst + 64 * (wave + LOADERS * j) * 4, gOff[j]);
            // publish the slot that is NBUF - 2 behind (keeps NBUF - 2 slices in flight per loader wave); simplest: wait all but
            // the youngest (NBUF - 2) * UNL
            if (z - z0 >= (uint32_t)K) {
                wait_vmcnt<K * UNL>();
                const uint32_t ps = (slot + NBUF - K) % NBUF;
                if (lane == 0) atomicAdd((uint32_t*)&full[ps], 1u);
            }
            slot = slot + 1; if (slot == NBUF) { slot = 0; ++gen; }
        }
        // drain: publish the remaining slices
        wait_vmcnt<0>();
        const uint32_t done = z1 - z0;
        for (uint32_t i = (done >= (uint32_t)K ? done - K : 0); i < done; ++i)
            if (lane == 0) atomicAdd((uint32_t*)&full[i % NBUF], 1u);
        return;
    }
    const uint32_t tid = threadIdx.x - LOADERS * 64;
    const uint32_t rowsPerPass = NT / a.tw;
    const uint32_t lx = tid % a.tw, ly0 = tid / a.tw;
    uint32_t cellOff[PER], ldsOff[PER];
    float w0[PER], w1[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t ly = ly0 + k * rowsPerPass;
        const uint32_t x = tx * a.tw + lx, y = ty * a.th + ly;
        cellOff[k] = (x < OX && y < OY) ? (y * OX + x) * 4u : 0xFFFFFFFFu;
        ldsOff[k] = ((ly * 3 / 2) * cpr * 4 + 2 * lx) * 4u;
        w0[k] = 0.25f + 0.001f * (float)lx;
        w1[k] = 0.25f + 0.002f * (float)ly;
    }
    uint32_t slot = 0, gen = 1;
    for (uint32_t z = z0; z < z1; ++z) {
        while (full[slot] < gen * LOADERS) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const char* cur = reinterpret_cast<const char*>(smem + slot * kSlot);
        const rsrc_t ro = make_rsrc(a.out + (size_t)z * OX * OY, (a.flags & 2) ? 0u : outBytes);
        float s00[PER], s01[PER], s10[PER], s11[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float* pa = reinterpret_cast<const float*>(cur + ldsOff[k]);
            const float* pb = reinterpret_cast<const float*>(cur + ldsOff[k] + cpr * 16);
            s00[k] = pa[0]; s01[k] = pa[1]; s10[k] = pb[0]; s11[k] = pb[1];
        }
        float r[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float top = (1.f - w0[k]) * s00[k] + w0[k] * s01[k];
            const float bot = (1.f - w0[k]) * s10[k] + w0[k] * s11[k];
            r[k] = (1.f - w1[k]) * top + w1[k] * bot;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (threadIdx.x % 64 == 0) atomicAdd((uint32_t*)&freed[slot], 1u);
#pragma unroll
        for (int k = 0; k < PER; ++k) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r[k]), ro, cellOff[k], 0, 2);
        slot = slot + 1; if (slot == NBUF) { slot = 0; ++gen; }
    }
}

__global__ void __launch_bounds__(256) sum3(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                            float4* __restrict__ o, size_t n, int nt)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float4 x = a[i], y = b[i], z = c[i];
        float4 r = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
        if (nt) {
            float* q = reinterpret_cast<float*>(&o[i]);
            typedef float v4 __attribute__((ext_vector_type(4)));
            v4 rv = {r.x, r.y, r.z, r.w};
            __builtin_nontemporal_store(rv, reinterpret_cast<v4*>(q));
        } else o[i] = r;
    }
}

__global__ void __launch_bounds__(256) fill_kernel(float* p, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = (float)(i & 1023) * 0.001f;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
};

template <typename F>
static void run(const char* name, double readBytes, double writeBytes, F&& launch, int reps = 4)
{
    Timer t;
    std::vector<float> ms;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(t.a));
        launch();
        CK(hipEventRecord(t.b));
        CK(hipEventSynchronize(t.b));
        CK(hipGetLastError());
        float m = 0;
        CK(hipEventElapsedTime(&m, t.a, t.b));
        if (r > 0) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    printf("{\"case\": \"%s\", \"ms_min\": %.4f, \"ms_med\": %.4f, \"read_GB\": %.3f, \"write_GB\": %.3f, \"TBps_med\": %.3f}\n", name, ms[0], med,
           readBytes / 1e9, writeBytes / 1e9, (readBytes + writeBytes) / med / 1e9);
    fflush(stdout);
}

static uint32_t* gPlanL = nullptr;
static float* gPlanW = nullptr;

template <int NT, int PER, int UN, int NBUF, bool PRIV>
static void run_tile(const char* tag, const float* in, float* out, uint32_t nz, uint32_t tw, uint32_t zpb, uint32_t flags)
{
    Args a{};
    a.planL = gPlanL; a.planW = gPlanW;
    a.in = in; a.out = out; a.nz = nz; a.zpb = zpb; a.tw = tw; a.th = NT * PER / tw; a.flags = flags;
    a.tilesX = (OX + tw - 1) / tw;
    a.nTiles = a.tilesX * ((OY + a.th - 1) / a.th);
    const uint32_t nrows = a.th * 3 / 2 + 1, cpr = tw / 2 + 1;
    if (nrows * cpr > (uint32_t)UN * NT) { printf("skip %s: %u chunks > %d\n", tag, nrows * cpr, UN * NT); return; }
    const size_t lds = (size_t)(PRIV ? 4 : 1) * NBUF * UN * NT * 16;
    if (lds > 160 * 1024) { printf("skip %s: lds %zu\n", tag, lds); return; }
    auto kern = tile<NT, PER, UN, NBUF, PRIV>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t owners = PRIV ? 4 : 1;
    uint32_t gx = (a.nTiles + owners - 1) / owners;
    if (flags & 32) gx = ((gx + 7) / 8) * 8;
    const dim3 grid(gx, (nz + zpb - 1) / zpb);
    char name[256];
    snprintf(name, sizeof name, "%s NT%d PER%d UN%d NBUF%d%s tw%u th%u zpb%u flags%u lds%zu", tag, NT, PER, UN, NBUF, PRIV ? " PRIV" : "", tw, a.th, zpb,
             flags, lds);
    const double rd = (flags & 1) ? 0 : (double)a.nTiles * nrows * cpr * 16.0 * nz;
    const double wr = (flags & 2) ? 0 : (double)OX * OY * 4.0 * nz;
    run(name, rd, wr, [&] { kern<<<grid, PRIV ? 256 : NT, lds>>>(a); });
}

// the same launch on a sheared footprint: tau source rows per output column
template <int NT, int PER, int UN, int NBUF>
static void run_shear(const float* in, float* out, uint32_t nz, uint32_t tw, uint32_t zpb, uint32_t flags, double tau)
{
    Args a{};
    a.in = in; a.out = out; a.nz = nz; a.zpb = zpb; a.tw = tw; a.th = NT * PER / tw; a.flags = flags | 2048u;
    a.planL = gPlanL; a.planW = gPlanW;
    a.tilesX = (OX + tw - 1) / tw;
    a.nTiles = a.tilesX * ((OY + a.th - 1) / a.th);
    const int nrows = (int)a.th * 3 / 2 + 1;
    const int S = (int)(tw * tau + 0.5);
    std::vector<uint32_t> rel;
    int rowsUsed = 0;
    for (int i = 0; i < nrows + S; ++i) {
        int xlo = 0, xhi = (int)tw - 1;
        if (tau > 0) {
            xlo = std::max(0, (int)std::floor((i - nrows) / tau) + 1);
            xhi = std::min((int)tw - 1, (int)std::floor(i / tau));
        } else if (i >= nrows) break;
        if (xhi < xlo) continue;
        const int first = (2 * xlo) & ~3, last = 2 * xhi + 1;
        for (int c = first; c <= last; c += 4) rel.push_back(((uint32_t)i << 20) | (uint32_t)c);
        ++rowsUsed;
    }
    if (rel.size() > (size_t)UN * NT) { printf("skip shear %.3f: %zu chunks > %d\n", tau, rel.size(), UN * NT); return; }
    uint32_t* dRel;
    CK(hipMalloc(&dRel, rel.size() * 4));
    CK(hipMemcpy(dRel, rel.data(), rel.size() * 4, hipMemcpyHostToDevice));
    a.rel = dRel; a.relCount = (uint32_t)rel.size(); a.shearRows = (uint32_t)S;
    const size_t lds = (size_t)NBUF * UN * NT * 16;
    auto kern = tile<NT, PER, UN, NBUF, false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint32_t gx = a.nTiles;
    if (flags & 32) gx = ((gx + 7) / 8) * 8;
    const dim3 grid(gx, (nz + zpb - 1) / zpb);
    char name[256];
    snprintf(name, sizeof name, "shear tau%.3f rows%d seg%.0fB NT%d PER%d UN%d NBUF%d tw%u th%u zpb%u flags%u", tau, rowsUsed, rel.size() * 16.0 / rowsUsed, NT, PER, UN,
             NBUF, tw, a.th, zpb, flags);
    run(name, (double)a.nTiles * rel.size() * 16.0 * nz, (double)OX * OY * 4.0 * nz, [&] { kern<<<grid, NT, lds>>>(a); });
    CK(hipFree(dRel));
}

template <int NT, int PER, int UN, int NBUF, int LOADERS>
static void run_split(const char* tag, const float* in, float* out, uint32_t nz, uint32_t tw, uint32_t zpb, uint32_t flags)
{
    Args a{};
    a.in = in; a.out = out; a.nz = nz; a.zpb = zpb; a.tw = tw; a.th = NT * PER / tw; a.flags = flags;
    a.tilesX = (OX + tw - 1) / tw;
    a.nTiles = a.tilesX * ((OY + a.th - 1) / a.th);
    const uint32_t nrows = a.th * 3 / 2 + 1, cpr = tw / 2 + 1;
    if (nrows * cpr > (uint32_t)UN * NT) { printf("skip %s: %u chunks > %d\n", tag, nrows * cpr, UN * NT); return; }
    const size_t lds = (size_t)NBUF * UN * NT * 16;
    if (lds > 158 * 1024) { printf("skip %s: lds %zu\n", tag, lds); return; }
    auto kern = tile_split<NT, PER, UN, NBUF, LOADERS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(a.nTiles, (nz + zpb - 1) / zpb);
    char name[256];
    snprintf(name, sizeof name, "%s SPLIT NT%d PER%d UN%d NBUF%d L%d tw%u th%u zpb%u flags%u lds%zu", tag, NT, PER, UN, NBUF, LOADERS, tw, a.th, zpb, flags, lds);
    const double rd = (flags & 1) ? 0 : (double)a.nTiles * nrows * cpr * 16.0 * nz;
    const double wr = (flags & 2) ? 0 : (double)OX * OY * 4.0 * nz;
    run(name, rd, wr, [&] { kern<<<grid, NT + LOADERS * 64, lds>>>(a); });
}



template <int NT, int PER, int UN, int NBUF, int SB>
static void run_sb(const float* in, float* out, uint32_t nz, uint32_t tw, uint32_t zpb)
{
    Args a{};
    a.in = in; a.out = out; a.nz = nz; a.zpb = zpb; a.tw = tw; a.th = NT * PER / tw; a.flags = 0;
    a.tilesX = (OX + tw - 1) / tw;
    a.nTiles = a.tilesX * ((OY + a.th - 1) / a.th);
    const uint32_t nrows = a.th * 3 / 2 + 1, cpr = tw / 2 + 1;
    if (nrows * cpr > (uint32_t)UN * NT) { printf("skip sb: %u chunks > %d\n", nrows * cpr, UN * NT); return; }
    const size_t lds = (size_t)NBUF * UN * NT * 16;
    auto kern = tile_sb<NT, PER, UN, NBUF, SB>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(a.nTiles, (nz + zpb - 1) / zpb);
    char name[256];
    snprintf(name, sizeof name, "sb NT%d PER%d UN%d NBUF%d SB%d tw%u th%u zpb%u lds%zu", NT, PER, UN, NBUF, SB, tw, a.th, zpb, lds);
    run(name, (double)a.nTiles * nrows * cpr * 16.0 * nz, (double)OX * OY * 4.0 * nz, [&] { kern<<<grid, NT, lds>>>(a); });
}

template <int NT, int PER, int TW>
struct Shape {
    static constexpr int TH = NT * PER / TW;
    static constexpr int CHUNKS = (TH * 3 / 2 + 1) * (TW / 2 + 1);
    static constexpr int UN = (CHUNKS + NT - 1) / NT;
};

template <int NT, int PER, int TW, int NBUF>
static void grid_case(const float* in, float* out, uint32_t nz, uint32_t zpb, uint32_t flags)
{
    using S = Shape<NT, PER, TW>;
    if constexpr ((size_t)NBUF * S::UN * NT * 16 <= 160 * 1024 && ((NBUF - 2) * S::UN + (NBUF - 1) * PER) < 64)
        run_tile<NT, PER, S::UN, NBUF, false>("grid", in, out, nz, TW, zpb, flags);
}

template <int NT, int PER, int TW>
static void grid_shape(const float* in, float* out, uint32_t nz)
{
    for (uint32_t flags : {0u, 32u}) {
        grid_case<NT, PER, TW, 2>(in, out, nz, 50, flags);
        grid_case<NT, PER, TW, 3>(in, out, nz, 50, flags);
        grid_case<NT, PER, TW, 4>(in, out, nz, 50, flags);
    }
}

int main(int argc, char** argv)
{
    const uint32_t nz = 200;
    const char* only = argc > 1 ? argv[1] : "";
    const size_t nIn = (size_t)nz * IX * IY + (size_t)64 * 1024 * 1024;  // slack: linear mode and the last rows read a little past
    const size_t nOut = (size_t)nz * OX * OY;
    float *in, *out;
    CK(hipMalloc(&in, nIn * 4));
    CK(hipMalloc(&out, nOut * 4));
    fill_kernel<<<4096, 256>>>(in, nIn);
    CK(hipMemset(out, 0, nOut * 4));
    CK(hipDeviceSynchronize());
    CK(hipMalloc(&gPlanL, (size_t)OX * OY * 4));
    CK(hipMalloc(&gPlanW, (size_t)OX * OY * 8));
    CK(hipMemset(gPlanL, 0, (size_t)OX * OY * 4));
    fill_kernel<<<1024, 256>>>(gPlanW, (size_t)OX * OY * 2);
    CK(hipDeviceSynchronize());
    auto want = [&](const char* g) { return only[0] == 0 || strstr(only, g) != nullptr; };

    if (want("ceil")) {
        const size_t n4 = nOut / 4;
        const float4* a = reinterpret_cast<const float4*>(in);
        for (int nt = 0; nt < 2; ++nt)
            for (int blocks : {2048, 8192}) {
                char name[64];
                snprintf(name, sizeof name, "sum3 nt%d blocks%d", nt, blocks);
                run(name, 3.0 * nOut * 4, 1.0 * nOut * 4, [&] { sum3<<<blocks, 256>>>(a, a + n4, a + 2 * n4, reinterpret_cast<float4*>(out), n4, nt); });
            }
    }
    if (want("base")) {
        run_tile<256, 4, 4, 2, false>("base", in, out, nz, 128, 50, 0);
        run_tile<256, 4, 4, 2, false>("base-noload", in, out, nz, 128, 50, 1);
        run_tile<256, 4, 4, 2, false>("base-nostore", in, out, nz, 128, 50, 2);
        run_tile<256, 4, 4, 2, false>("base-neither", in, out, nz, 128, 50, 3);
        run_tile<256, 4, 4, 2, false>("base-linear", in, out, nz, 128, 50, 4);
        run_tile<256, 4, 4, 2, false>("base-linear-nostore", in, out, nz, 128, 50, 6);
        run_tile<256, 4, 4, 2, false>("base-plainstore", in, out, nz, 128, 50, 8);
        run_tile<256, 4, 4, 2, false>("base-ntload", in, out, nz, 128, 50, 16);
        run_tile<256, 4, 4, 2, false>("base-xcd", in, out, nz, 128, 50, 32);
        run_tile<256, 4, 4, 2, false>("base-zpb200", in, out, nz, 128, 200, 0);
        run_tile<256, 4, 4, 2, false>("base-zpb25", in, out, nz, 128, 25, 0);
    }
    if (want("depth")) {
        run_tile<256, 4, 4, 3, false>("depth", in, out, nz, 128, 50, 0);
        run_tile<256, 4, 4, 4, false>("depth", in, out, nz, 128, 50, 0);
        run_tile<256, 4, 4, 6, false>("depth", in, out, nz, 128, 50, 0);
        run_tile<256, 4, 4, 8, false>("depth", in, out, nz, 128, 50, 0);
        run_tile<256, 4, 4, 4, false>("depth-nostore", in, out, nz, 128, 50, 2);
        run_tile<256, 4, 4, 8, false>("depth-nostore", in, out, nz, 128, 50, 2);
        run_tile<256, 4, 4, 4, false>("depth-linear", in, out, nz, 128, 50, 4);
        run_tile<256, 4, 4, 4, false>("depth-ntload", in, out, nz, 128, 50, 16);
    }
    if (want("wide")) {
        run_tile<512, 4, 4, 2, false>("wide", in, out, nz, 256, 50, 0);
        run_tile<512, 4, 4, 3, false>("wide", in, out, nz, 256, 50, 0);
        run_tile<512, 4, 4, 4, false>("wide", in, out, nz, 256, 50, 0);
        run_tile<1024, 4, 4, 2, false>("wide", in, out, nz, 512, 50, 0);
        run_tile<1024, 4, 4, 2, false>("wide", in, out, nz, 1024, 50, 0);
        run_tile<512, 4, 4, 2, false>("wide", in, out, nz, 512, 50, 0);
        run_tile<256, 4, 4, 2, false>("wide", in, out, nz, 256, 50, 0);
        run_tile<256, 8, 7, 2, false>("tall", in, out, nz, 128, 50, 0);
        run_tile<256, 8, 7, 3, false>("tall", in, out, nz, 128, 50, 0);
        run_tile<256, 8, 7, 2, false>("tall", in, out, nz, 256, 50, 0);
    }
    if (want("priv")) {
        run_tile<64, 4, 4, 2, true>("priv", in, out, nz, 64, 50, 0);
        run_tile<64, 4, 4, 3, true>("priv", in, out, nz, 64, 50, 0);
        run_tile<64, 4, 4, 4, true>("priv", in, out, nz, 64, 50, 0);
        run_tile<64, 8, 7, 2, true>("priv", in, out, nz, 64, 50, 0);
        run_tile<64, 8, 7, 3, true>("priv", in, out, nz, 64, 50, 0);
        run_tile<64, 8, 7, 4, true>("priv", in, out, nz, 64, 50, 0);
        run_tile<64, 8, 7, 3, true>("priv-nostore", in, out, nz, 64, 50, 2);
        run_tile<64, 16, 13, 2, true>("priv", in, out, nz, 64, 50, 0);
        run_tile<64, 16, 13, 3, true>("priv", in, out, nz, 64, 50, 0);
    }
    if (want("split")) {
        run_split<256, 4, 4, 4, 1>("split", in, out, nz, 128, 50, 0);
        run_split<256, 4, 4, 6, 1>("split", in, out, nz, 128, 50, 0);
        run_split<256, 4, 4, 4, 2>("split", in, out, nz, 128, 50, 0);
        run_split<256, 4, 4, 6, 2>("split", in, out, nz, 128, 50, 0);
        run_split<256, 4, 4, 8, 2>("split", in, out, nz, 128, 50, 0);
        run_split<256, 4, 4, 6, 2>("split-nostore", in, out, nz, 128, 50, 2);
        run_split<512, 4, 4, 4, 2>("split", in, out, nz, 256, 50, 0);
        run_split<512, 4, 4, 4, 4>("split", in, out, nz, 256, 50, 0);
    }

    if (want("grid")) {
        grid_shape<256, 4, 128>(in, out, nz);
        grid_shape<256, 4, 256>(in, out, nz);
        grid_shape<256, 8, 256>(in, out, nz);
        grid_shape<512, 4, 256>(in, out, nz);
        grid_shape<512, 2, 256>(in, out, nz);
        grid_shape<512, 4, 512>(in, out, nz);
        grid_shape<512, 2, 512>(in, out, nz);
        grid_shape<1024, 4, 512>(in, out, nz);
        grid_shape<1024, 2, 512>(in, out, nz);
        grid_shape<1024, 2, 256>(in, out, nz);
        grid_shape<1024, 4, 1024>(in, out, nz);
        grid_shape<1024, 2, 1024>(in, out, nz);
    }
    if (want("big")) {
        using S = Shape<1024, 4, 512>;
        for (uint32_t flags : {1u, 2u, 3u, 8u, 16u, 32u, 34u, 33u})
            run_tile<1024, 4, S::UN, 2, false>("big", in, out, nz, 512, 50, flags);
        for (uint32_t zpb : {10u, 25u, 40u, 100u, 200u})
            run_tile<1024, 4, S::UN, 2, false>("big-zpb", in, out, nz, 512, zpb, 0);
        run_split<512, 4, 4, 4, 2>("split", in, out, nz, 256, 50, 0);
        run_split<512, 4, 4, 4, 1>("split", in, out, nz, 256, 50, 0);
        run_split<512, 4, 4, 3, 1>("split", in, out, nz, 256, 50, 0);
        run_split<512, 4, 4, 3, 2>("split", in, out, nz, 256, 50, 0);
        run_split<256, 4, 4, 3, 1>("split", in, out, nz, 128, 50, 0);
        run_split<256, 4, 4, 4, 1>("split", in, out, nz, 128, 50, 0);
        run_split<256, 4, 4, 5, 1>("split", in, out, nz, 128, 50, 0);
    }

    if (want("store")) {
        for (uint32_t flags : {0u, 128u, 256u, 32u, 32u + 128u, 32u + 256u, 2u}) {
            run_tile<256, 4, 4, 2, false>("store", in, out, nz, 128, 50, flags);
            run_tile<256, 4, 4, 3, false>("store", in, out, nz, 128, 50, flags);
            run_tile<512, 4, 4, 3, false>("store", in, out, nz, 256, 50, flags);
            run_tile<1024, 4, 4, 2, false>("store", in, out, nz, 512, 50, flags);
        }
    }

    if (want("sb")) {
        run_sb<256, 4, 4, 2, 1>(in, out, nz, 128, 48);
        run_sb<256, 4, 4, 2, 2>(in, out, nz, 128, 48);
        run_sb<256, 4, 4, 2, 4>(in, out, nz, 128, 48);
        run_sb<256, 4, 4, 2, 8>(in, out, nz, 128, 48);
        run_sb<256, 4, 4, 3, 4>(in, out, nz, 128, 48);
        run_sb<512, 4, 4, 2, 1>(in, out, nz, 256, 48);
        run_sb<512, 4, 4, 2, 4>(in, out, nz, 256, 48);
        run_sb<512, 4, 4, 2, 8>(in, out, nz, 256, 48);
        run_sb<1024, 4, 4, 2, 1>(in, out, nz, 512, 48);
        run_sb<1024, 4, 4, 2, 2>(in, out, nz, 512, 48);
        run_sb<1024, 4, 4, 2, 4>(in, out, nz, 512, 48);
        run_sb<1024, 4, 4, 2, 8>(in, out, nz, 512, 48);
        run_sb<1024, 4, 4, 2, 12>(in, out, nz, 512, 48);
    }

    if (want("shear")) {
        for (uint32_t flags : {0u, 32u})
            for (double tau : {0.0, 0.02, 0.04, 0.08, 0.12, 0.2}) {
                run_shear<1024, 4, 4, 2>(in, out, nz, 512, 50, flags, tau);
                run_shear<1024, 4, 4, 2>(in, out, nz, 256, 50, flags, tau);
                run_shear<512, 4, 4, 3>(in, out, nz, 256, 50, flags, tau);
            }
    }
    if (want("plan")) {
        for (uint32_t zpb : {50u, 25u, 100u, 200u, 10u})
            for (uint32_t flags : {0u, 4096u})
                run_tile<1024, 4, 4, 2, false>("plan", in, out, nz, 512, zpb, flags);
        for (uint32_t flags : {0u, 4096u}) run_shear<1024, 4, 4, 2>(in, out, nz, 512, 50, flags, 0.08);
    }
    if (want("alu")) {
        for (uint32_t flags : {0u, 512u, 1024u, 1536u, 3u, 512u + 3u, 1024u + 3u, 1536u + 3u}) {
            run_tile<1024, 4, 4, 2, false>("alu", in, out, nz, 512, 50, flags);
            run_tile<256, 4, 4, 2, false>("alu", in, out, nz, 128, 50, flags);
        }
    }
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
