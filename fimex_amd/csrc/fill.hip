// In-place 2-D fills (SOR Laplace fill and creep fill) for gfx950.
//
// Replaces mifi_fill2d_f (src/interpolation.c:1246-1376) and mifi_creepfill2d_f /
// mifi_creepfillval2d_f (:1378-1537) as driven slice by slice by processArray_
// (src/CDMInterpolator.cc:136-159).
//
// Both are Gauss-Seidel sweeps in place: cell (x, y) sees the already updated (x-1, y) and
// (x, y-1) and the not yet updated (x+1, y) and (x, y+1).  Any other order changes the result, so
// the sweep is executed as an anti-diagonal wavefront: all cells with x + y = d are independent
// once diagonal d-1 is done.  One workgroup owns one slice (slices are independent, which is where
// the chip-wide parallelism comes from) and steps through the diagonals with a workgroup barrier
// between them.  The sums that feed the first guess (mean, mean absolute deviation) are
// accumulated in the reference's scan order in double, by one wave reading LDS-staged tiles, so
// the first guess is bit-identical too.
#include "plan.hpp"

#include <vector>

namespace fimex_amd {

namespace {

constexpr int kFillBlock = 1024;
constexpr int kTile = 4096;  // floats staged in LDS per serial-sum step

struct SliceStats {
    unsigned long long nUndef;
    double average;
    double meanAbsDev;
    int status;  // 1 ok, -1 error
};

// sum of the defined values in scan order, double accumulator (interpolation.c:1256-1264, 1502-1513);
// mode 1: sum of |v - average| instead (:1288-1299).  Every lane of wave 0 runs the same chain.
template <int BLOCK = kFillBlock>
__device__ double serial_sum(const float* __restrict__ f, size_t total, int mode, double average, float* lds,
                             unsigned long long* nUndefOut)
{
    // The additions form one dependent chain (that is the point: the reference's order).  Everything around it is
    // taken off the chain: tiles are staged by the whole workgroup, wave 0 reads 8 values per LDS instruction pair
    // and prepares the 8 addends (NaN -> +0.0, which leaves a sum that started at +0.0 unchanged) before adding them.
    double sum = 0;
    unsigned long long nUndef = 0;
    for (size_t base = 0; base < total; base += kTile) {
        const size_t len = (total - base < (size_t)kTile) ? total - base : (size_t)kTile;
        __syncthreads();
        for (size_t i = threadIdx.x; i < (size_t)kTile; i += BLOCK) lds[i] = (i < len) ? f[base + i] : __uint_as_float(0x7fc00000u);
        __syncthreads();
        if (threadIdx.x < kWave) {
            const float4* t4 = reinterpret_cast<const float4*>(lds);
            const size_t groups = (len + 7) / 8;  // the tile is padded with NaN: padding adds +0.0 and is not counted
            for (size_t g = 0; g < groups; ++g) {
                const float4 a = t4[2 * g], b = t4[2 * g + 1];
                const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                double t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool undef = isnan(v[k]);
                    nUndef += undef;
                    t[k] = undef ? 0.0 : (mode == 0 ? (double)v[k] : fabs((double)v[k] - average));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += t[k];
            }
            nUndef -= (groups * 8 - len);  // the NaN padding of the last group
        }
    }
    if (nUndefOut) *nUndefOut = nUndef;
    return sum;  // valid in wave 0
}

// ---------------------------------------------------------------------------------- fill2d
struct Fill2dArgs {
    float* field;
    float* w;           // workspace, one float per cell
    SliceStats* stats;  // per slice
    uint32_t nx, ny;
    float relaxCrit, corrEff;
    unsigned long long maxLoop;
};

__global__ void __launch_bounds__(kFillBlock) fill2d_kernel(Fill2dArgs a)
{
    __shared__ __align__(16) float lds[kTile];
    __shared__ double shAverage, shCrit;
    __shared__ unsigned long long shUndef;
    const uint32_t nx = a.nx, ny = a.ny;
    const size_t total = (size_t)nx * ny;
    float* f = a.field + (size_t)blockIdx.x * total;
    float* w = a.w + (size_t)blockIdx.x * total;
    SliceStats* st = a.stats + blockIdx.x;

    unsigned long long nUndef = 0;
    const double sum = serial_sum(f, total, 0, 0., lds, &nUndef);
    if (threadIdx.x == 0) {
        shUndef = nUndef;
        const unsigned long long nDef = total - nUndef;
        shAverage = (nDef != 0) ? sum / (double)nDef : 0.;  // :1281
        st->nUndef = nUndef;
        st->status = 1;
    }
    __syncthreads();
    nUndef = shUndef;
    const unsigned long long nDef = total - nUndef;
    if (nDef == 0 || nUndef == 0) return;  // nothing to do, :1266-1268
    if (nx < 2 || ny < 2) { if (threadIdx.x == 0) st->status = -1; return; }  // the reference reads out of bounds here
    const double average = shAverage;

    const double dev = serial_sum(f, total, 1, average, lds, nullptr);
    if (threadIdx.x == 0) shCrit = (double)a.relaxCrit * (dev / (double)nDef);  // :1300-1302
    __syncthreads();
    const double crit = shCrit;

    const uint32_t nxm1 = nx - 1, nym1 = ny - 1;
    for (size_t i = threadIdx.x; i < total; i += kFillBlock) {  // :1288-1299 and :1311-1315
        const uint32_t x = (uint32_t)(i % nx), y = (uint32_t)(i / nx);
        float wi = 0.f;
        if (isnan(f[i])) {
            f[i] = (float)average;
            wi = 1.f;
        }
        if (x >= 1 && x < nxm1 && y >= 1 && y < nym1) wi *= a.corrEff;
        w[i] = wi;
    }
    __syncthreads();

    const float crtest = (float)(crit * a.corrEff);  // :1341
    // interior cells 1 <= x <= nx-2, 1 <= y <= ny-2; diagonal d = x + y runs 2 .. nx+ny-4
    const bool hasInterior = nx > 2 && ny > 2;
    for (unsigned long long n = 0; n < a.maxLoop; ++n) {
        const bool check = (n < (a.maxLoop - 5)) && (n % 10 == 0);  // :1339-1340, unsigned like the reference
        int bad = 0;
        if (hasInterior) {
            const uint32_t dLast = (nx - 2) + (ny - 2);
            for (uint32_t d = 2; d <= dLast; ++d) {
                const uint32_t xlo = (d > (ny - 2)) ? d - (ny - 2) : 1;
                const uint32_t xhi = (d - 1 < nx - 2) ? d - 1 : nx - 2;
                for (uint32_t x = xlo + threadIdx.x; x <= xhi; x += kFillBlock) {
                    const size_t p = (size_t)(d - x) * nx + x;
                    const float fc = f[p];
                    const float e = (float)((double)(f[p + 1] + f[p - 1] + f[p + nx] + f[p - nx]) * 0.25 - (double)fc);  // :1332
                    const float wp = w[p];
                    f[p] = fc + e * wp;  // :1333
                    if (check && (fabsf(e * wp) > crtest)) bad = 1;  // :1349
                }
                __syncthreads();
            }
        }
        if (check) {
            if (!__syncthreads_or(bad)) return;  // converged, :1355-1359 (before the border pass)
        }
        for (uint32_t y = 1 + threadIdx.x; y < nym1; y += kFillBlock) {  // :1363-1366
            const size_t r = (size_t)y * nx;
            f[r] += (f[r + 1] - f[r]) * w[r];
            f[r + nxm1] += (f[r + nx - 2] - f[r + nxm1]) * w[r + nxm1];
        }
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < nx; x += kFillBlock) {  // :1367-1370
            const size_t b = (size_t)nym1 * nx + x;
            f[x] += (f[nx + x] - f[x]) * w[x];
            f[b] += (f[b - nx] - f[b]) * w[b];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ fill2d, systolic version
// The anti-diagonal wavefront above pays one workgroup barrier per diagonal (nx + ny of them per sweep).  Here the
// same Gauss-Seidel order is kept with (almost) no barriers: a wave owns a band of 64 consecutive rows, lane l owns
// row y0 + l and walks it left to right, one column per step, l steps behind the lane above it:
//     step s: lane l updates column x = 1 + s - l.
// Then everything a cell needs is one step old in a neighbouring lane or in the lane itself:
//     left  f_new(x-1, y)   own result of the previous step
//     up    f_new(x, y-1)   the previous step's result of lane l-1            (wave shift by one lane)
//     right f_old(x+1, y)   own row, next column                              (LDS ring, see below)
//     down  f_old(x, y+1)   what lane l+1 reads as its "right" in this step   (wave shift by one lane)
// Rows are streamed through a per-wave LDS ring in SKEWED columns x' = x + l, so that all lanes are at the same
// x' = 1 + s: 16-column chunks of all 64 rows are loaded two chunks ahead with coalesced row-segment loads, results
// overwrite the ring in place and finished chunks are flushed with coalesced stores.  The NaN mask that selects the
// weight is kept as bits, one word per 32 skewed columns and row, so every lane switches words in the same step.
// Bands are pipelined over the 16 waves of the workgroup: the first lane of band b needs the last row of band b-1,
// published through an LDS progress counter (release/acquire at workgroup scope; all waves of a workgroup share
// the CU's L1).  One workgroup barrier per sweep remains (border pass, convergence test).
constexpr int kV2Waves = 16;
constexpr int kV2Threads = kV2Waves * kWave;
constexpr int kCh = 16;            // skewed columns per chunk
constexpr int kRingW = 2 * kCh;    // ring width (two chunks)
constexpr int kPitch = kRingW + 1; // conflict-free: bank = (lane + x') mod 32
constexpr int kMaxBands = 4096;

struct Fill2dV2Args {
    float* field;
    uint32_t* maskS;          // [nz][ny][mws] skewed NaN-mask words of the interior rows
    unsigned char* mbRows;    // [nz][2][nx] NaN mask of row 0 and row ny-1
    unsigned char* mbCols;    // [nz][2][ny] NaN mask of column 0 and column nx-1
    SliceStats* stats;
    uint32_t nx, ny, mws;
    float relaxCrit, corrEff;
    unsigned long long maxLoop;
};

// value of lane l-1 (lane 0 keeps its own): one DPP move, "wave_shr:1" (0x138), no LDS round trip
__device__ __forceinline__ float lane_from_above(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
// value held by lane `idx` (wave-uniform index) broadcast through an SGPR
__device__ __forceinline__ float lane_value(float v, int idx)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), idx));
}

constexpr int kHandW = 192;  // columns of a band's last row kept in LDS for the band below

// LDS hand-off between consecutive bands: the wave of band b publishes its last row's new values in
// hand[b % 16][(b / 16) & 1][x % 192] and a counter "band, columns finished"; the wave of band b + 1 reads them 64
// columns at a time and publishes how far it has read, which bounds how far the producer may run ahead.
struct Handoff {
    float* data;             // [16][2][kHandW]
    unsigned int* produced;  // [16][2]  (band + 1) << 19 | columns of the last row that are final
    unsigned int* consumed;  // [16][2]  (band + 1) << 19 | columns the band below has taken over
};
__device__ __forceinline__ unsigned int hand_tag(uint32_t band, uint32_t cols) { return ((band + 1) << 19) | cols; }

// one band of one sweep, executed by one wave.  Global memory is touched only in the "event" between two 16-step
// chunks: loads issued there are consumed one event later, stores are never waited for (the sweep ends with a
// workgroup barrier); the 16 steps in between run on registers and LDS.
__device__ void fill2d_band(float* __restrict__ f, const uint32_t* __restrict__ maskS, float* ring, Handoff hand, uint32_t b,
                            uint32_t nx, uint32_t ny, uint32_t mws, float wInt, float wZero, bool check, float crtest, int& bad)
{
    using rsrc_t = __amdgpu_buffer_rsrc_t;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t y0 = 1 + kWave * b;
    const uint32_t nrow = min((uint32_t)kWave, (ny - 1) - y0);  // rows y0 .. y0 + nrow - 1 <= ny - 2
    const uint32_t L = nrow - 1;                                 // last lane with a row
    const bool rowValid = lane < nrow;
    const uint32_t y = y0 + min(lane, L);
    const uint32_t C = nx - 2;                                   // interior columns 1 .. C
    const uint32_t xpEnd = C + L;                                // last skewed column with work
    float* ringRow = ring + lane * kPitch;
    const float* ringBelow = ring + min(lane + 1, (uint32_t)kWave - 1) * kPitch;
    const float left0 = f[(size_t)y * nx];                       // border column 0, not touched by the sweep
    const uint32_t* mrow = maskS + (size_t)y * mws;
    // the band's rows plus the row above and the row below as one buffer: masked lanes use an out-of-range offset
    // (loads return 0, stores are dropped), so every memory instruction is issued unconditionally
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(f + (size_t)(y0 - 1) * nx, 0, (nrow + 2) * nx * 4u, 0x00020000);
    const uint32_t kOob = 0xFFFFFFFFu;

    // chunk c = skewed columns [c*kCh, c*kCh + kCh) of all 64 rows; lane -> (row 4*it + lane/16, column lane%16)
    const uint32_t crow = lane >> 4, ccol = lane & 15;
    float stage[16];
    auto chunk_off = [&](uint32_t c, uint32_t it, bool store) -> uint32_t {
        const uint32_t row = 4 * it + crow;
        const int64_t x = (int64_t)c * kCh + ccol - row;  // unskewed column
        const bool ok = row < nrow && (store ? (x >= 1 && x <= (int64_t)C) : (x >= 0 && x <= (int64_t)nx - 1));
        return ok ? (uint32_t)(((row + 1) * nx + x) * 4u) : kOob;
    };
    auto load_chunk = [&](uint32_t c) {
#pragma unroll
        for (uint32_t it = 0; it < 16; ++it)
            stage[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, chunk_off(c, it, false), 0, 0));
    };
    auto commit_chunk = [&](uint32_t c) {
#pragma unroll
        for (uint32_t it = 0; it < 16; ++it) ring[(4 * it + crow) * kPitch + ((c * kCh + ccol) & (kRingW - 1))] = stage[it];
    };
    auto flush_chunk = [&](uint32_t c) {
        float v[16];
#pragma unroll
        for (uint32_t it = 0; it < 16; ++it) v[it] = ring[(4 * it + crow) * kPitch + ((c * kCh + ccol) & (kRingW - 1))];
#pragma unroll
        for (uint32_t it = 0; it < 16; ++it)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[it]), rs, chunk_off(c, it, true), 0, 0);
    };
    auto load_block = [&](uint32_t rowInBuf, uint32_t k) {  // 64 columns of the row above (0) / below (nrow + 1)
        const uint32_t col = 64 * k + lane;
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, col <= nx - 1 ? (rowInBuf * nx + col) * 4u : kOob, 0, 0));
    };

    // hand-off slots: mine (towards band b + 1) and the one of band b - 1
    const uint32_t slotOut = (b % kV2Waves) * 2 + ((b / kV2Waves) & 1);
    const uint32_t slotIn = ((b - 1) % kV2Waves) * 2 + (((b - 1) / kV2Waves) & 1);  // unused for b == 0
    float* handOut = hand.data + slotOut * kHandW;
    const float* handIn = hand.data + slotIn * kHandW;
    if (lane == 0) {
        __hip_atomic_store(&hand.produced[slotOut], hand_tag(b, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&hand.consumed[slotOut], hand_tag(b, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // block k (columns 64k .. 64k+63) of the row above from the hand-off of band b - 1
    auto take_above = [&](uint32_t k) -> float {
        const unsigned int need = hand_tag(b - 1, min(64 * k + 64, C + 1));
        // a larger band tag means the producer has finished band b - 1 long ago (its data stay in the other parity slot)
        while (__hip_atomic_load(&hand.produced[slotIn], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
        const float v = handIn[(64 * k + lane) % kHandW];
        if (lane == 0)
            __hip_atomic_store(&hand.consumed[slotIn], hand_tag(b - 1, 64 * k + 64), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return v;
    };

    // ---- prologue: chunks 0 and 1 in LDS, chunk 2 in flight; first blocks and mask words.
    // Registers that receive a load at an event (stage[], upLd, downLd, mwLd) are read only at a LATER event.
    load_chunk(0);
    commit_chunk(0);
    load_chunk(1);
    commit_chunk(1);
    load_chunk(2);
    float upCur = (b == 0) ? load_block(0, 0) : take_above(0), upLd = 0.f;
    float downA = load_block(nrow + 1, 0), downB = downA, downLd = 0.f;  // current / next (landed) / in flight
    uint32_t downIssued = 0;
    bool downLdValid = false;
    uint32_t mw = mrow[0], mwN = mrow[1], mwLd = mrow[2];
    float prevRes = 0.f;
    float prevRight = ringRow[1];  // lane 0 is at column 1 in the first step: its centre is skewed column 1

    const uint32_t nChunks = xpEnd / kCh + 1;
    for (uint32_t c = 0; c < nChunks; ++c) {
        const uint32_t xpc = c * kCh;
        if (c > 0) {
            // ---- event at the start of chunk c
            flush_chunk(c - 1);      // results of the chunk just finished -> global (never waited for)
            commit_chunk(c + 1);     // loaded one event ago, into the ring slot the flush has just read
            load_chunk(c + 2);       // consumed at the next event
            if ((c & 1) == 0) {      // x' is a multiple of 32: every lane switches mask words now
                mw = mwN;
                mwN = mwLd;
                mwLd = mrow[min(c / 2 + 2, mws - 1)];
            }
            // publish how far the last row has got, and do not run more than the hand-off window ahead of the band below
            if (xpc > L) {
                if (lane == 0)
                    __hip_atomic_store(&hand.produced[slotOut], hand_tag(b, xpc - L), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (y0 + nrow < ny - 1) {  // there is a band below
                    const unsigned int limit = xpc + kCh - L;  // columns < limit are written during this chunk
                    while (true) {
                        const unsigned int cns = __hip_atomic_load(&hand.consumed[slotOut], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (limit <= (cns & 0x7FFFFu) + kHandW) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            // row above: lane 0 is at column x'
            if ((xpc & 63) == 0) {
                if (b == 0) upCur = upLd;  // border row 0: from global, requested two chunks ago
                else upCur = take_above(xpc >> 6);
            }
            if (b == 0 && ((xpc + 2 * kCh) & 63) == 0) upLd = load_block(0, (xpc + 2 * kCh) >> 6);
            // row below: the last lane is at column x' - L
            if (downLdValid) { downB = downLd; downLdValid = false; }
            if (xpc + 3 * kCh > L) {
                const uint32_t k = (xpc + 3 * kCh - L) >> 6;
                if (k > downIssued) { downLd = load_block(nrow + 1, k); downIssued = k; downLdValid = true; }
            }
        }
        const uint32_t xp0 = max(xpc, 1u), xp1 = min(xpc + kCh - 1, xpEnd);
        for (uint32_t xp = xp0; xp <= xp1; ++xp) {
            if (xp > L && ((xp - L) & 63) == 0) downA = downB;  // the last lane enters block (x' - L) / 64
            const int64_t x = (int64_t)xp - lane;
            const bool active = rowValid && x >= 1 && x <= (int64_t)C;
            const uint32_t rp = (xp + 1) & (kRingW - 1);
            const float right = ringRow[rp];
            float down = ringBelow[rp];  // f_old(x, y+1): row lane+1 holds column x at its skewed column x' + 1
            const float center = prevRight;
            float up = lane_from_above(prevRes);
            const float upFirst = lane_value(upCur, (int)(xp & 63));
            if (lane == 0) up = upFirst;
            const float downLast = lane_value(downA, (int)((xp >= L) ? ((xp - L) & 63) : 0));
            if (lane == L) down = downLast;
            const float left = (x == 1) ? left0 : prevRes;
            const float wv = ((mw >> (xp & 31)) & 1u) ? wInt : wZero;
            const float e = (float)((double)(((right + left) + down) + up) * 0.25 - (double)center);  // interpolation.c:1332
            const float res = center + e * wv;                                                        // :1333
            if (active) {
                ringRow[xp & (kRingW - 1)] = res;
                prevRes = res;
                if (lane == L) handOut[(uint32_t)x % kHandW] = res;  // the band below reads its "up" values here
                if (check && (fabsf(e * wv) > crtest)) bad = 1;     // :1349
            }
            prevRight = right;
        }
    }
    flush_chunk(nChunks - 1);
    if (lane == 0) __hip_atomic_store(&hand.produced[slotOut], hand_tag(b, C + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ void __launch_bounds__(kV2Threads) fill2d_kernel_v2(Fill2dV2Args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];  // rings [16][64][33] floats, hand-off [16][2][192] + counters
    __shared__ double shAverage, shCrit;
    __shared__ unsigned long long shUndef;
    float* rings = smem;
    Handoff hand;
    hand.data = smem + kV2Waves * kWave * kPitch;
    hand.produced = reinterpret_cast<unsigned int*>(hand.data + kV2Waves * 2 * kHandW);
    hand.consumed = hand.produced + kV2Waves * 2;
    const uint32_t nx = a.nx, ny = a.ny, mws = a.mws;
    const size_t total = (size_t)nx * ny;
    float* f = a.field + (size_t)blockIdx.x * total;
    uint32_t* maskS = a.maskS + (size_t)blockIdx.x * ny * mws;
    unsigned char* mbTop = a.mbRows + (size_t)blockIdx.x * 2 * nx;
    unsigned char* mbBot = mbTop + nx;
    unsigned char* mbLeft = a.mbCols + (size_t)blockIdx.x * 2 * ny;
    unsigned char* mbRight = mbLeft + ny;
    SliceStats* st = a.stats + blockIdx.x;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;

    unsigned long long nUndef = 0;
    const double sum = serial_sum<kV2Threads>(f, total, 0, 0., smem, &nUndef);
    if (threadIdx.x == 0) {
        shUndef = nUndef;
        const unsigned long long nDef = total - nUndef;
        shAverage = (nDef != 0) ? sum / (double)nDef : 0.;
        st->nUndef = nUndef;
        st->status = 1;
    }
    __syncthreads();
    nUndef = shUndef;
    const unsigned long long nDef = total - nUndef;
    if (nDef == 0 || nUndef == 0) return;
    const double average = shAverage;
    const double dev = serial_sum<kV2Threads>(f, total, 1, average, smem, nullptr);
    if (threadIdx.x == 0) shCrit = (double)a.relaxCrit * (dev / (double)nDef);
    __syncthreads();
    const double crit = shCrit;
    const float avgf = (float)average;

    // first guess + masks (:1288-1299).  Interior rows: one wave per row, skewed ballot words.
    for (uint32_t y = wave; y < ny; y += kV2Waves) {
        float* row = f + (size_t)y * nx;
        if (y == 0 || y == ny - 1) {
            unsigned char* mb = (y == 0) ? mbTop : mbBot;
            for (uint32_t x = lane; x < nx; x += kWave) {
                const bool u = isnan(row[x]);
                mb[x] = u;
                if (u) row[x] = avgf;
            }
        } else {
            const uint32_t l = (y - 1) & (kWave - 1);
            uint32_t* mrow = maskS + (size_t)y * mws;
            for (uint32_t base = 0; base < mws * 32; base += kWave) {
                const int64_t x = (int64_t)base + lane - l;
                const bool in = x >= 0 && x < (int64_t)nx;
                const bool u = in && isnan(row[in ? x : 0]);
                const unsigned long long m = __ballot(u);
                if (lane == 0) {
                    mrow[base / 32] = (uint32_t)m;
                    if (base / 32 + 1 < mws) mrow[base / 32 + 1] = (uint32_t)(m >> 32);
                }
                if (u) row[x] = avgf;
                if (in && x == 0) mbLeft[y] = u;
                if (in && x == (int64_t)nx - 1) mbRight[y] = u;
            }
        }
    }
    __syncthreads();

    const float wInt = 1.f * a.corrEff, wZero = 0.f * a.corrEff;  // :1311-1315
    const float crtest = (float)(crit * a.corrEff);
    const uint32_t nxm1 = nx - 1, nym1 = ny - 1;
    const uint32_t nBands = (ny - 2 + kWave - 1) / kWave;
    float* ring = rings + wave * kWave * kPitch;
    for (unsigned long long n = 0; n < a.maxLoop; ++n) {
        const bool check = (n < (a.maxLoop - 5)) && (n % 10 == 0);
        int bad = 0;
        if (threadIdx.x < kV2Waves * 2) { hand.produced[threadIdx.x] = 0; hand.consumed[threadIdx.x] = 0; }
        __syncthreads();
        for (uint32_t b = wave; b < nBands; b += kV2Waves)
            fill2d_band(f, maskS, ring, hand, b, nx, ny, mws, wInt, wZero, check, crtest, bad);
        if (check) {
            if (!__syncthreads_or(bad)) return;  // converged (:1355-1359), before the border pass
        } else {
            __syncthreads();
        }
        for (uint32_t y = 1 + threadIdx.x; y < nym1; y += kV2Threads) {  // :1363-1366
            const size_t r = (size_t)y * nx;
            const float wl = mbLeft[y] ? 1.f : 0.f, wr = mbRight[y] ? 1.f : 0.f;
            f[r] += (f[r + 1] - f[r]) * wl;
            f[r + nxm1] += (f[r + nx - 2] - f[r + nxm1]) * wr;
        }
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < nx; x += kV2Threads) {  // :1367-1370
            const size_t bo = (size_t)nym1 * nx + x;
            const float wt = mbTop[x] ? 1.f : 0.f, wb = mbBot[x] ? 1.f : 0.f;
            f[x] += (f[nx + x] - f[x]) * wt;
            f[bo] += (f[bo - nx] - f[bo]) * wb;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------- creep fill
struct CreepArgs {
    float* field;
    signed char* w;     // workspace, one byte per cell (:1389)
    unsigned short* r;  // workspace, one ushort per cell (:1394)
    SliceStats* stats;
    uint32_t nx, ny;
    int useDefault;
    float defaultVal;
    unsigned short repeat;
    signed char setWeight;
};

__global__ void __launch_bounds__(kFillBlock) creepfill_kernel(CreepArgs a)
{
    __shared__ __align__(16) float lds[kTile];
    __shared__ unsigned long long shUndef;
    __shared__ float shDefault;
    __shared__ unsigned int shChanged;
    const uint32_t nx = a.nx, ny = a.ny;
    const size_t total = (size_t)nx * ny;
    float* f = a.field + (size_t)blockIdx.x * total;
    signed char* w = a.w + (size_t)blockIdx.x * total;
    unsigned short* r = a.r + (size_t)blockIdx.x * total;
    SliceStats* st = a.stats + blockIdx.x;

    unsigned long long nUndef = 0;
    const double sum = serial_sum(f, total, 0, 0., lds, &nUndef);
    if (threadIdx.x == 0) {
        shUndef = nUndef;
        const unsigned long long nDef = total - nUndef;
        shDefault = a.useDefault ? a.defaultVal : ((nDef != 0) ? (float)(sum / (double)nDef) : 0.f);  // :1516
        st->nUndef = nUndef;
        st->status = 1;
    }
    __syncthreads();
    nUndef = shUndef;
    const unsigned long long nDef = total - nUndef;
    if (nDef == 0 || nUndef == 0) return;  // :1384-1386, :1515
    if (nx < 2 || ny < 2) { if (threadIdx.x == 0) st->status = -1; return; }
    const float defaultVal = shDefault;
    const unsigned short repeat = a.repeat;

    for (size_t i = threadIdx.x; i < total; i += kFillBlock) {  // :1408-1421
        if (isnan(f[i])) { w[i] = 0; r[i] = 0; f[i] = defaultVal; }
        else { w[i] = a.setWeight; r[i] = repeat; }
    }
    __syncthreads();

    const uint32_t nxm1 = nx - 1, nym1 = ny - 1;
    const bool hasInterior = nx > 2 && ny > 2;
    unsigned long long l = 0;
    unsigned int changedInLoop = 1;
    while (changedInLoop > 0 && l < nDef) {  // :1430
        l++;
        if (threadIdx.x == 0) shChanged = 0;
        __syncthreads();
        unsigned int mine = 0;
        if (hasInterior) {
            const uint32_t dLast = (nx - 2) + (ny - 2);
            for (uint32_t d = 2; d <= dLast; ++d) {
                const uint32_t xlo = (d > (ny - 2)) ? d - (ny - 2) : 1;
                const uint32_t xhi = (d - 1 < nx - 2) ? d - 1 : nx - 2;
                for (uint32_t x = xlo + threadIdx.x; x <= xhi; x += kFillBlock) {
                    const size_t p = (size_t)(d - x) * nx + x;
                    if (r[p] < repeat) {  // :1443
                        const int wr = w[p + 1], wl = w[p - 1], wd = w[p + nx], wu = w[p - nx];
                        const size_t wsum = (size_t)(wr + wl + wd + wu);  // :1445
                        if (wsum != 0) {
                            float v = f[p];
                            v += wr * f[p + 1] + wl * f[p - 1] + wd * f[p + nx] + wu * f[p - nx];  // :1451
                            v /= (float)(1 + wsum);                                                // :1452
                            f[p] = v;
                            w[p] = 1;
                            r[p] = r[p] + 1;
                            mine++;
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (mine) atomicAdd(&shChanged, mine);
        __syncthreads();
        changedInLoop = shChanged;
        __syncthreads();
    }
    for (unsigned int k = 0; k < repeat; ++k) {  // :1464-1489
        for (uint32_t y = 1 + threadIdx.x; y < nym1; y += kFillBlock) {
            const size_t row = (size_t)y * nx;
            if (r[row] < repeat) {
                f[row] += f[row + 1] * w[row + 1];
                f[row] /= (float)(1 + w[row + 1]);
                w[row] = 1;
            }
            if (r[row + nxm1] < repeat) {
                f[row + nxm1] += f[row + nx - 2] * w[row + nx - 2];
                f[row + nxm1] /= (float)(1 + w[row + nx - 2]);
                w[row + nxm1] = 1;
            }
        }
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < nx; x += kFillBlock) {
            const size_t b = (size_t)nym1 * nx + x;
            if (r[x] < repeat) {
                f[x] += f[nx + x] * w[nx + x];
                f[x] /= (float)(1 + w[nx + x]);
                w[x] = 1;
            }
            if (r[b] < repeat) {
                f[b] += f[b - nx] * w[b - nx];
                f[b] /= (float)(1 + w[b - nx]);
                w[b] = 1;
            }
        }
        __syncthreads();
    }
}

void collect_stats(const DeviceArray<SliceStats>& d_stats, size_t nz, size_t* h_nChanged, hipStream_t stream, const char* what)
{
    std::vector<SliceStats> st(nz);
    FA_HIP(hipMemcpyAsync(st.data(), d_stats.get(), nz * sizeof(SliceStats), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    bool failed = false;
    for (size_t z = 0; z < nz; ++z) {
        if (h_nChanged) h_nChanged[z] = (size_t)st[z].nUndef;
        if (st[z].status != 1) failed = true;
    }
    if (failed) throw Error(std::string(what) + ": slices need nx >= 2 and ny >= 2");
}

}  // namespace

void run_fill2d(size_t nx, size_t ny, size_t nz, float* d_field, float relaxCrit, float corrEff, size_t maxLoop,
                size_t* h_nChanged, hipStream_t stream)
{
    if (nx * ny == 0 || nz == 0) return;  // :1248
    FA_REQUIRE(nx <= 0x7FFFFFFFu && ny <= 0x7FFFFFFFu && nz <= 0x7FFFFFFFu, "fill2d: slice too large");
    DeviceArray<SliceStats> stats(nz);
    FA_HIP(hipMemsetAsync(stats.get(), 0, nz * sizeof(SliceStats), stream));
    const size_t nBands = ny > 2 ? (ny - 2 + kWave - 1) / kWave : 0;
    // the systolic kernel packs "band, column" into 32-bit hand-off counters and addresses 66 rows through one buffer
    if (tuning("FILL_V2", 1) != 0 && nx >= 4 && ny >= 4 && nBands < (size_t)kMaxBands && nx < (1u << 19) &&
        (size_t)(kWave + 2) * nx * 4 < 0xFFFFFFFFull) {
        const uint32_t mws = (uint32_t)((nx + kWave + 31) / 32 + 2);  // skewed columns 0 .. nx + 62, plus prefetch slack
        DeviceArray<uint32_t> maskS(nz * ny * mws);
        DeviceArray<unsigned char> mbRows(nz * 2 * nx), mbCols(nz * 2 * ny);
        Fill2dV2Args a{};
        a.field = d_field;
        a.maskS = maskS.get();
        a.mbRows = mbRows.get();
        a.mbCols = mbCols.get();
        a.stats = stats.get();
        a.nx = (uint32_t)nx;
        a.ny = (uint32_t)ny;
        a.mws = mws;
        a.relaxCrit = relaxCrit;
        a.corrEff = corrEff;
        a.maxLoop = maxLoop;
        constexpr size_t ldsBytes = (size_t)kV2Waves * kWave * kPitch * sizeof(float) + (size_t)kV2Waves * 2 * kHandW * sizeof(float) +
                                    (size_t)kV2Waves * 4 * sizeof(unsigned int);
        static bool attrSet = false;
        if (!attrSet) {
            FA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill2d_kernel_v2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)ldsBytes));
            attrSet = true;
        }
        fill2d_kernel_v2<<<dim3((uint32_t)nz), kV2Threads, ldsBytes, stream>>>(a);
        FA_HIP(hipGetLastError());
        collect_stats(stats, nz, h_nChanged, stream, "fill2d");
        return;
    }
    DeviceArray<float> w(nx * ny * nz);
    Fill2dArgs a{};
    a.field = d_field;
    a.w = w.get();
    a.stats = stats.get();
    a.nx = (uint32_t)nx;
    a.ny = (uint32_t)ny;
    a.relaxCrit = relaxCrit;
    a.corrEff = corrEff;
    a.maxLoop = maxLoop;
    fill2d_kernel<<<dim3((uint32_t)nz), kFillBlock, 0, stream>>>(a);
    FA_HIP(hipGetLastError());
    collect_stats(stats, nz, h_nChanged, stream, "fill2d");
}

void run_creepfill(size_t nx, size_t ny, size_t nz, float* d_field, bool useDefault, float defaultVal,
                   unsigned short repeat, char setWeight, size_t* h_nChanged, hipStream_t stream)
{
    if (nx * ny == 0 || nz == 0) return;  // :1380
    FA_REQUIRE(nx <= 0x7FFFFFFFu && ny <= 0x7FFFFFFFu && nz <= 0x7FFFFFFFu, "creepfill: slice too large");
    DeviceArray<signed char> w(nx * ny * nz);
    DeviceArray<unsigned short> r(nx * ny * nz);
    DeviceArray<SliceStats> stats(nz);
    FA_HIP(hipMemsetAsync(stats.get(), 0, nz * sizeof(SliceStats), stream));
    CreepArgs a{};
    a.field = d_field;
    a.w = w.get();
    a.r = r.get();
    a.stats = stats.get();
    a.nx = (uint32_t)nx;
    a.ny = (uint32_t)ny;
    a.useDefault = useDefault ? 1 : 0;
    a.defaultVal = defaultVal;
    a.repeat = repeat;
    a.setWeight = (signed char)setWeight;
    creepfill_kernel<<<dim3((uint32_t)nz), kFillBlock, 0, stream>>>(a);
    FA_HIP(hipGetLastError());
    collect_stats(stats, nz, h_nChanged, stream, useDefault ? "creepfillval2d" : "creepfill2d");
}

}  // namespace fimex_amd
