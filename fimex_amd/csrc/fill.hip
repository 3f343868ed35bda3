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
#include "creep_rects.hpp"

#include <cstdio>

#include <vector>

namespace fimex_amd {

namespace {

constexpr int kFillBlock = 1024;

struct SliceStats {
    unsigned long long nUndef;
    double average;     // first guess (mean of the defined cells, or the caller's default value)
    double meanAbsDev;  // fill2d: relaxCrit * mean absolute deviation = the convergence criterion
    int status;         // 1 ok, -1 error
    int skip;           // nothing to fill or nothing defined: the slice is left alone
    unsigned long long sweepBound;  // creep fills: the loop ends after this many sweeps at the latest (:1430: the number of defined cells;
                                    // for a rectangle of a decomposed fill: of the whole slice)
};

// sum of the defined values in scan order, double accumulator (interpolation.c:1256-1264, 1502-1513);
// mode 1: sum of |v - average| instead (:1288-1299); mode 2: only count the undefined cells.
//
// The additions form one dependent chain -- that is the point: the reference's order, hence its rounding.  All
// else is taken off the chain: waves 1.. turn tile t+1 into ready double addends in LDS (undefined -> +0.0, which
// leaves a sum that started at +0.0 unchanged; padding likewise) and count the undefined cells, while wave 0 walks
// tile t with nothing but 16-byte LDS reads and v_add_f64.  buf: 2 * kSumTile doubles of LDS.
constexpr int kSumTile = 2048;

__device__ __forceinline__ double sum_addend(float v, int mode, double average, unsigned int& nUndef)
{
    const bool undef = isnan(v);
    nUndef += undef;
    return undef ? 0.0 : (mode == 0 ? (double)v : fabs((double)v - average));
}

template <int BLOCK = kFillBlock>
__device__ double serial_sum(const float* __restrict__ f, size_t total, int mode, double average, double* buf,
                             unsigned long long* nUndefOut)
{
    __shared__ unsigned long long shCount;
    constexpr int kProducers = BLOCK - kWave;
    const size_t nTiles = (total + kSumTile - 1) / kSumTile;
    unsigned int myUndef = 0;
    double sum = 0;
    if (threadIdx.x == 0) shCount = 0;
    // tile 0 by everybody
    for (size_t i = threadIdx.x; i < (size_t)kSumTile; i += BLOCK)
        buf[i] = (i < total) ? sum_addend(f[i], mode, average, myUndef) : 0.0;
    __syncthreads();
    for (size_t t = 0; t < nTiles; ++t) {
        if (threadIdx.x < kWave) {
            if (mode != 2) {
                const double2* b2 = reinterpret_cast<const double2*>(buf + (t & 1) * kSumTile);
                double2 q0 = b2[0], q1 = b2[1], q2 = b2[2], q3 = b2[3];
#pragma unroll 2
                for (int g = 1; g <= kSumTile / 8; ++g) {  // the next 8 addends are read while these 8 are added
                    const int h = (g < kSumTile / 8) ? g : 0;
                    const double2 n0 = b2[4 * h], n1 = b2[4 * h + 1], n2 = b2[4 * h + 2], n3 = b2[4 * h + 3];
                    sum += q0.x; sum += q0.y; sum += q1.x; sum += q1.y;
                    sum += q2.x; sum += q2.y; sum += q3.x; sum += q3.y;
                    q0 = n0; q1 = n1; q2 = n2; q3 = n3;
                }
            }
        } else if (t + 1 < nTiles) {
            const size_t base = (t + 1) * kSumTile;
            double* dst = buf + ((t + 1) & 1) * kSumTile;
            for (size_t i = threadIdx.x - kWave; i < (size_t)kSumTile; i += kProducers)
                dst[i] = (base + i < total) ? sum_addend(f[base + i], mode, average, myUndef) : 0.0;
        }
        __syncthreads();
    }
    if (nUndefOut) {
        if (myUndef) atomicAdd(&shCount, (unsigned long long)myUndef);
        __syncthreads();
        *nUndefOut = shCount;
        __syncthreads();
    }
    return sum;  // valid in wave 0
}

// ---- the same sums without walking the chain: "binade-parallel" evaluation, bit for bit the sequential result.
//
// While the running sum S stays inside one binade [2^e, 2^(e+1)), it is a multiple of u = 2^(e-52) and every
// S <- fl(S + a) rounds the exact value to a multiple of u, so fl(S + a) = S + rn_u(a) whenever a is not exactly halfway
// between two multiples of u (rn_u: round to the nearest multiple).  The rounded addends k = rn_u(a) / u are integers and
// integer sums are associative: a chunk of 1024 elements contributes I = sum k, in any order, PROVIDED S provably stays
// inside the binade for all 1024 partial sums.  With A = sum |k| and m = |S| / u (an integer in [2^52, 2^53)) that is
// guaranteed by  m - A >= 2^52 + 1  and  m + A <= 2^53 - 1  (the +-1 keeps the exact, unrounded partial sums inside as
// well), and A < 2^50 keeps all integer arithmetic exact in doubles.  A chunk that fails any test -- a tie, a binade
// crossing, S = 0, non-finite values -- is re-evaluated at the binade S has by then, or walked element by element.
// Per super-block of 16 chunks: every wave evaluates its chunk at the binade S had after the previous super-block,
// then wave 0 strings the 16 results together (lanes = chunks, prefix over I) and repairs what failed.
constexpr int kSumE = 16;                 // elements per lane
constexpr int kChunk = kWave * kSumE;     // elements per wave and super-block
constexpr int kNoBinade = 0x7fffffff;

__device__ __forceinline__ double pow2d(int e) { return __longlong_as_double((long long)(e + 1023) << 52); }  // |e| < 1000
__device__ __forceinline__ int exponent_of(double s) { return (int)((__double_as_longlong(s) >> 52) & 0x7FF) - 1023; }
__device__ __forceinline__ bool binade_usable(double s, int e) { return s != 0.0 && e > -900 && e < 900; }  // excludes inf, NaN, subnormals
__device__ __forceinline__ double lane_value_d(double v, int idx)
{
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)b, idx), hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), idx);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// v of the lane CTRL names (DPP: 0x110 + n = n lanes up within the row of 16, 0x142 / 0x143 = last lane of the previous
// row / of the first half), 0.0 where there is none or the row is masked out: cross-lane adds without an LDS round trip
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_d(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// inclusive prefix sum within each row of 16 lanes
__device__ __forceinline__ double row_scan_d(double v)
{
    v += dpp_d<0x111>(v);
    v += dpp_d<0x112>(v);
    v += dpp_d<0x114>(v);
    v += dpp_d<0x118>(v);
    return v;
}
// sum over the wave, in every lane (exact integers: the order does not matter)
__device__ __forceinline__ double wave_sum_d(double v)
{
    v = row_scan_d(v);           // lane 15 of each row: the row's sum
    v += dpp_d<0x142, 0xa>(v);   // rows 1 and 3 += row before
    v += dpp_d<0x143, 0xc>(v);   // rows 2 and 3 += first half
    return lane_value_d(v, kWave - 1);
}

struct ChunkSum {
    double I, A;
    bool ok;
};

__device__ __forceinline__ void chunk_load(const float* __restrict__ f, size_t base, size_t total, float (&v)[kSumE])
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int j = 0; j < kSumE; ++j) {  // element j * 64 + lane of the chunk: coalesced; the integer sums do not care about order
        const size_t i = base + (size_t)j * kWave + lane;
        v[j] = (i < total) ? f[i] : 0.f;
    }
}

__device__ __forceinline__ void chunk_addends(const float (&v)[kSumE], size_t base, size_t total, int mode, double average,
                                              double (&a)[kSumE], unsigned int* nUndef)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int j = 0; j < kSumE; ++j) {
        const bool in = base + (size_t)j * kWave + lane < total;
        const bool undef = isnan(v[j]);
        if (nUndef) *nUndef += undef;
        a[j] = (undef || !in) ? 0.0 : (mode == 0 ? (double)v[j] : fabs((double)v[j] - average));
    }
}

// one wave, 1024 consecutive elements from `base` (already in v): integer image of the addends at binade e
__device__ ChunkSum chunk_eval(const float (&v)[kSumE], size_t base, size_t total, int mode, double average, int e, unsigned int* nUndef)
{
    double a[kSumE];
    chunk_addends(v, base, total, mode, average, a, nUndef);
    const double scale = pow2d(52 - e);
    double sI = 0, sA = 0;
    bool tie = false;
#pragma unroll
    for (int j = 0; j < kSumE; ++j) {
        const double t = a[j] * scale;  // exact: a power of two
        const double k = rint(t);
        tie |= (fabs(t - k) == 0.5);
        sI += k;
        sA += fabs(k);
    }
    ChunkSum r;
    r.I = wave_sum_d(sI);
    r.A = wave_sum_d(sA);
    r.ok = !__any(tie) && r.A < 0x1p50;  // false for inf and NaN as well
    return r;
}

// one wave, the same 1024 elements one after the other on the running sum
__device__ double chunk_chain(const float* __restrict__ f, size_t base, size_t total, int mode, double average, double S)
{
    float v[kSumE];
    double a[kSumE];
    chunk_load(f, base, total, v);
    chunk_addends(v, base, total, mode, average, a, nullptr);
#pragma unroll
    for (int j = 0; j < kSumE; ++j) {
        for (int l = 0; l < kWave; ++l) S += lane_value_d(a[j], l);
    }
    return S;
}

template <int BLOCK = kFillBlock>
__device__ double binade_sum(const float* __restrict__ f, size_t total, int mode, double average, unsigned long long* nUndefOut)
{
    constexpr int kWaves = BLOCK / kWave;
    static_assert(kWaves <= 16, "lanes 0..15 of wave 0 stand for the chunks of a super-block");
    constexpr size_t kSuper = (size_t)kWaves * kChunk;
    __shared__ double shI[kWaves], shA[kWaves];
    __shared__ int shOk[kWaves];
    __shared__ int shE;
    __shared__ unsigned long long shCount;
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    unsigned int myUndef = 0;
    double S = 0;  // wave 0
    if (threadIdx.x == 0) { shE = kNoBinade; shCount = 0; }
    __syncthreads();
    if (mode == 2) {
        for (size_t i = threadIdx.x; i < total; i += BLOCK) myUndef += isnan(f[i]);
    } else {
        float vNext[kSumE];
        chunk_load(f, (size_t)wave * kChunk, total, vNext);
        for (size_t sb = 0; sb < total; sb += kSuper) {
            const int e = shE;
            float vCur[kSumE];
#pragma unroll
            for (int j = 0; j < kSumE; ++j) vCur[j] = vNext[j];
            chunk_load(f, sb + kSuper + (size_t)wave * kChunk, total, vNext);  // the next super-block, while this one is worked on
            ChunkSum cs = chunk_eval(vCur, sb + (size_t)wave * kChunk, total, mode, average, e == kNoBinade ? 0 : e, &myUndef);
            if (lane == 0) { shI[wave] = cs.I; shA[wave] = cs.A; shOk[wave] = (cs.ok && e != kNoBinade) ? 1 : 0; }
            __syncthreads();
            if (wave == 0) {
                const int nCh = (int)(((total - sb < kSuper ? total - sb : kSuper) + kChunk - 1) / kChunk);
                const double I = lane < (uint32_t)kWaves ? shI[lane] : 0.0, A = lane < (uint32_t)kWaves ? shA[lane] : 0.0;
                const bool ok = lane < (uint32_t)kWaves && shOk[lane] != 0;
                int w0 = 0;
                while (w0 < nCh) {
                    int eS = exponent_of(S);
                    if (binade_usable(S, eS) && eS == e) {
                        const double n = fabs(S) * pow2d(52 - eS), sg = S < 0 ? -1.0 : 1.0;
                        const double x = ((int)lane >= w0 && (int)lane < nCh) ? sg * I : 0.0;
                        const double incl = row_scan_d(x);
                        const double m = n + (incl - x);  // |S| / u before chunk `lane`, if all chunks from w0 on can be taken
                        const bool good = ok && (m - A >= 0x1p52 + 1.0) && (m + A <= 0x1p53 - 1.0);
                        const unsigned long long bad = __ballot((int)lane >= w0 && (int)lane < nCh && !good);
                        const int wf = bad ? (int)__ffsll((long long)bad) - 1 : nCh;
                        if (wf > w0) S = sg * ((n + lane_value_d(incl, wf - 1)) * pow2d(eS - 52));
                        w0 = wf;
                        if (w0 == nCh) break;
                    }
                    // chunk w0 on its own: at the binade S is in now, else element by element
                    const size_t cb = sb + (size_t)w0 * kChunk;
                    eS = exponent_of(S);
                    bool done = false;
                    if (binade_usable(S, eS)) {
                        float vOne[kSumE];
                        chunk_load(f, cb, total, vOne);
                        const ChunkSum one = chunk_eval(vOne, cb, total, mode, average, eS, nullptr);
                        const double n = fabs(S) * pow2d(52 - eS), sg = S < 0 ? -1.0 : 1.0;
                        if (one.ok && (n - one.A >= 0x1p52 + 1.0) && (n + one.A <= 0x1p53 - 1.0)) {
                            S = sg * ((n + sg * one.I) * pow2d(eS - 52));
                            done = true;
                        }
                    }
                    if (!done) S = chunk_chain(f, cb, total, mode, average, S);
                    ++w0;
                }
                if (lane == 0) {
                    const int eS = exponent_of(S);
                    shE = binade_usable(S, eS) ? eS : kNoBinade;
                }
            }
            __syncthreads();
        }
    }
    if (nUndefOut) {
        if (myUndef) atomicAdd(&shCount, (unsigned long long)myUndef);
        __syncthreads();
        *nUndefOut = shCount;
        __syncthreads();
    }
    return S;  // valid in wave 0
}

// algo 0: the chain (serial_sum), 1: binade-parallel
template <int BLOCK = kFillBlock>
__device__ double scan_order_sum(const float* __restrict__ f, size_t total, int mode, double average, double* buf,
                                 unsigned long long* nUndefOut, int algo)
{
    if (algo == 0) return serial_sum<BLOCK>(f, total, mode, average, buf, nUndefOut);
    return binade_sum<BLOCK>(f, total, mode, average, nUndefOut);
}

// ---- the same sum over the whole chip (algo 2).  One workgroup walking a 36 MB slice super-block by super-block takes
// 2.3 ms per pass, all of it synchronisation and the stitch of wave 0.  What a chunk needs is only the BINADE the running
// sum has when it arrives there, and an approximate prefix sum predicts that: (1) every chunk's plain double sum, (2) their
// exclusive prefix -> predicted binade per chunk, (3) every chunk's integer image at its predicted binade, (4) one wave per
// slice strings the chunks together exactly as wave 0 does above -- a chunk whose prediction is wrong (next to a binade
// crossing) or whose test fails is re-evaluated or walked element by element there.  The result is the reference's sum
// whatever the prediction was; a bad prediction only costs time.
struct SumWork {
    double* approx;        // [slices][nChunks] plain sum of the chunk's addends (any order)
    double* I;             // integer image of the chunk at binade e
    double* A;
    int* e;                // predicted binade of the running sum before the chunk (kNoBinade: none)
    int* ok;
    unsigned int* undef;   // undefined cells of the chunk
    size_t nChunks;
};

struct SumJob {
    const float* values;   // [slices][total]
    size_t total;
    int mode;              // 0 sum, 1 sum of |v - average|, 2 count only
    const SliceStats* stats;  // mode 1: average per slice; slices with skip set are left out (nullptr: averageAll, none skipped)
    double averageAll;
};

__device__ __forceinline__ bool sum_slice_active(const SumJob& j, uint32_t slice, double& average)
{
    average = j.averageAll;
    if (j.stats && j.mode == 1) {
        if (j.stats[slice].skip) return false;
        average = j.stats[slice].average;
    }
    return true;
}

// (grids of the per-chunk kernels are flat: blocks of a slice, then the next slice -- gridDim.y stops at 65535 slices)
__global__ void __launch_bounds__(kBlock) sum_approx_kernel(SumJob j, SumWork w, uint32_t blocksPerSlice)
{
    const uint32_t lane = threadIdx.x & (kWave - 1), slice = blockIdx.x / blocksPerSlice;
    const size_t c = (size_t)(blockIdx.x % blocksPerSlice) * (kBlock / kWave) + threadIdx.x / kWave;
    double average;
    if (c >= w.nChunks || !sum_slice_active(j, slice, average)) return;
    const float* f = j.values + (size_t)slice * j.total;
    float v[kSumE];
    double a[kSumE];
    unsigned int nUndef = 0;
    chunk_load(f, c * kChunk, j.total, v);
    chunk_addends(v, c * kChunk, j.total, j.mode == 2 ? 0 : j.mode, average, a, &nUndef);
    double s = 0;
#pragma unroll
    for (int k = 0; k < kSumE; ++k) s += a[k];
    s = wave_sum_d(s);
    const unsigned int u = (unsigned int)wave_sum_d((double)nUndef);
    if (lane == 0) {
        w.approx[(size_t)slice * w.nChunks + c] = s;
        w.undef[(size_t)slice * w.nChunks + c] = u;
    }
}

// exclusive prefix of the approximate chunk sums -> predicted binade; one workgroup per slice
__global__ void __launch_bounds__(kFillBlock) sum_predict_kernel(SumJob j, SumWork w)
{
    __shared__ double shTot[kFillBlock];
    const uint32_t slice = blockIdx.x;
    double average;
    if (!sum_slice_active(j, slice, average)) return;
    const double* ap = w.approx + (size_t)slice * w.nChunks;
    int* e = w.e + (size_t)slice * w.nChunks;
    const size_t per = (w.nChunks + kFillBlock - 1) / kFillBlock;
    const size_t c0 = (size_t)threadIdx.x * per, c1 = c0 + per < w.nChunks ? c0 + per : w.nChunks;
    double mine = 0;
    for (size_t c = c0; c < c1; ++c) mine += ap[c];
    shTot[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < kFillBlock; off <<= 1) {  // inclusive scan of the thread totals
        const double add = threadIdx.x >= (uint32_t)off ? shTot[threadIdx.x - off] : 0.0;
        __syncthreads();
        shTot[threadIdx.x] += add;
        __syncthreads();
    }
    double P = shTot[threadIdx.x] - mine;
    for (size_t c = c0; c < c1; ++c) {
        const int eP = exponent_of(P);
        e[c] = binade_usable(P, eP) ? eP : kNoBinade;
        P += ap[c];
    }
}

__global__ void __launch_bounds__(kBlock) sum_eval_kernel(SumJob j, SumWork w, uint32_t blocksPerSlice)
{
    const uint32_t lane = threadIdx.x & (kWave - 1), slice = blockIdx.x / blocksPerSlice;
    const size_t c = (size_t)(blockIdx.x % blocksPerSlice) * (kBlock / kWave) + threadIdx.x / kWave;
    double average;
    if (c >= w.nChunks || !sum_slice_active(j, slice, average)) return;
    const size_t idx = (size_t)slice * w.nChunks + c;
    const int e = w.e[idx];
    float v[kSumE];
    chunk_load(j.values + (size_t)slice * j.total, c * kChunk, j.total, v);
    const ChunkSum cs = chunk_eval(v, c * kChunk, j.total, j.mode, average, e == kNoBinade ? 0 : e, nullptr);
    if (lane == 0) {
        w.I[idx] = cs.I;
        w.A[idx] = cs.A;
        w.ok[idx] = (cs.ok && e != kNoBinade) ? 1 : 0;
    }
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double wave_scan_d(double v)
{
    v = row_scan_d(v);
    v += dpp_d<0x142, 0xa>(v);
    v += dpp_d<0x143, 0xc>(v);
    return v;
}

// one wave per slice: the chunks in order, 64 at a time (lanes = chunks)
__device__ double stitch_chunks(const SumJob& j, const SumWork& w, uint32_t slice, double average, unsigned long long* nUndefOut)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const float* f = j.values + (size_t)slice * j.total;
    const size_t off = (size_t)slice * w.nChunks;
    double S = 0;
    unsigned long long undef = 0;
    for (size_t g0 = 0; g0 < w.nChunks; g0 += kWave) {
        const int nCh = (int)(w.nChunks - g0 < (size_t)kWave ? w.nChunks - g0 : (size_t)kWave);
        const bool mine = (int)lane < nCh;
        undef += mine ? w.undef[off + g0 + lane] : 0u;
        if (j.mode == 2) continue;
        const double I = mine ? w.I[off + g0 + lane] : 0.0, A = mine ? w.A[off + g0 + lane] : 0.0;
        const bool ok = mine && w.ok[off + g0 + lane] != 0;
        const int eC = mine ? w.e[off + g0 + lane] : kNoBinade;
        int w0 = 0;
        while (w0 < nCh) {
            int eS = exponent_of(S);
            if (binade_usable(S, eS)) {
                const double n = fabs(S) * pow2d(52 - eS), sg = S < 0 ? -1.0 : 1.0;
                const bool cand = (int)lane >= w0 && mine && ok && eC == eS;
                const double x = cand ? sg * I : 0.0;
                const double incl = wave_scan_d(x);
                const double m = n + (incl - x);  // |S| / u before chunk `lane`, if all chunks from w0 on can be taken
                const bool good = cand && (m - A >= 0x1p52 + 1.0) && (m + A <= 0x1p53 - 1.0);
                const unsigned long long bad = __ballot((int)lane >= w0 && mine && !good);
                const int wf = bad ? (int)__ffsll((long long)bad) - 1 : nCh;
                if (wf > w0) S = sg * ((n + lane_value_d(incl, wf - 1)) * pow2d(eS - 52));
                w0 = wf;
                if (w0 == nCh) break;
            }
            // chunk w0 on its own: at the binade S is in now, else element by element
            const size_t cb = (g0 + (size_t)w0) * kChunk;
            eS = exponent_of(S);
            bool done = false;
            if (binade_usable(S, eS)) {
                float vOne[kSumE];
                chunk_load(f, cb, j.total, vOne);
                const ChunkSum one = chunk_eval(vOne, cb, j.total, j.mode, average, eS, nullptr);
                const double n = fabs(S) * pow2d(52 - eS), sg = S < 0 ? -1.0 : 1.0;
                if (one.ok && (n - one.A >= 0x1p52 + 1.0) && (n + one.A <= 0x1p53 - 1.0)) {
                    S = sg * ((n + sg * one.I) * pow2d(eS - 52));
                    done = true;
                }
            }
            if (!done) S = chunk_chain(f, cb, j.total, j.mode, average, S);
            ++w0;
        }
    }
    if (nUndefOut) *nUndefOut = (unsigned long long)wave_sum_d((double)undef);  // < 2^53: exact
    return S;
}

// the two uses: a plain sum into host-visible cells (scan_sum), and the statistics of the fills
struct StitchOut {
    double* sum;                 // [slices] or nullptr
    unsigned long long* nUndef;  // [slices] or nullptr
    SliceStats* stats;           // fills: nullptr otherwise
    size_t total;
    int useDefault;
    float defaultVal;
    float relaxCrit;
};

__global__ void __launch_bounds__(kWave) sum_stitch_kernel(SumJob j, SumWork w, StitchOut o)
{
    const uint32_t slice = blockIdx.x;
    double average;
    if (!sum_slice_active(j, slice, average)) return;
    unsigned long long nUndef = 0;
    const double S = stitch_chunks(j, w, slice, average, &nUndef);
    if (threadIdx.x != 0) return;
    if (o.sum) o.sum[slice] = S;
    if (o.nUndef) o.nUndef[slice] = nUndef;
    if (!o.stats) return;
    SliceStats* st = o.stats + slice;
    if (j.mode != 1) {  // first pass: count, first guess (:1281, :1516)
        const unsigned long long nDef = o.total - nUndef;
        st->nUndef = nUndef;
        st->average = o.useDefault ? (double)o.defaultVal : ((nDef != 0) ? S / (double)nDef : 0.);
        st->status = 1;
        st->skip = (nDef == 0 || nUndef == 0);
        st->sweepBound = nDef;
    } else {            // second pass: the convergence criterion (:1302)
        const unsigned long long nDef = o.total - st->nUndef;
        st->meanAbsDev = (double)o.relaxCrit * (S / (double)nDef);
    }
}

struct SumBuffers {
    DeviceArray<double> approx, I, A;
    DeviceArray<int> e, ok;
    DeviceArray<unsigned int> undef;
    SumWork work{};
    SumBuffers(size_t total, size_t slices)
    {
        const size_t nChunks = ceil_div(total, (size_t)kChunk), n = nChunks * slices;
        approx.allocate(n); I.allocate(n); A.allocate(n); e.allocate(n); ok.allocate(n); undef.allocate(n);
        work = SumWork{approx.get(), I.get(), A.get(), e.get(), ok.get(), undef.get(), nChunks};
    }
};

void launch_chip_sum(const SumJob& j, const SumBuffers& b, size_t slices, const StitchOut& o, hipStream_t stream)
{
    const size_t blocksPerSlice = ceil_div(b.work.nChunks, (size_t)(kBlock / kWave));
    FA_REQUIRE(blocksPerSlice * slices <= 0x7FFFFFFFull, "too many slices for one call");
    const dim3 perChunk((uint32_t)(blocksPerSlice * slices));
    sum_approx_kernel<<<perChunk, kBlock, 0, stream>>>(j, b.work, (uint32_t)blocksPerSlice);
    if (j.mode != 2) {
        sum_predict_kernel<<<dim3((uint32_t)slices), kFillBlock, 0, stream>>>(j, b.work);
        sum_eval_kernel<<<perChunk, kBlock, 0, stream>>>(j, b.work, (uint32_t)blocksPerSlice);
    }
    sum_stitch_kernel<<<dim3((uint32_t)slices), kWave, 0, stream>>>(j, b.work, o);
    FA_HIP(hipGetLastError());
}

struct ScanSumArgs {
    const float* values;
    size_t n;
    int mode, algo;
    double average;
    double* sum;
    unsigned long long* nUndef;
};

__global__ void __launch_bounds__(kFillBlock) scan_sum_kernel(ScanSumArgs a)
{
    __shared__ __align__(16) double lds[2 * kSumTile];
    unsigned long long nUndef = 0;
    const double s = scan_order_sum(a.values, a.n, a.mode, a.average, lds, &nUndef, a.algo);
    if (threadIdx.x == 0) { *a.sum = s; *a.nUndef = nUndef; }
}

// ---------------------------------------------------------------------------------- fill2d
struct Fill2dArgs {
    float* field;
    float* w;           // workspace, one float per cell
    SliceStats* stats;  // per slice
    uint32_t nx, ny;
    float relaxCrit, corrEff;
    unsigned long long maxLoop;
    int sumAlgo;
};

__global__ void __launch_bounds__(kFillBlock) fill2d_kernel(Fill2dArgs a)
{
    __shared__ __align__(16) double lds[2 * kSumTile];
    __shared__ double shAverage, shCrit;
    __shared__ unsigned long long shUndef;
    const uint32_t nx = a.nx, ny = a.ny;
    const size_t total = (size_t)nx * ny;
    float* f = a.field + (size_t)blockIdx.x * total;
    float* w = a.w + (size_t)blockIdx.x * total;
    SliceStats* st = a.stats + blockIdx.x;

    unsigned long long nUndef = 0;
    const double sum = scan_order_sum(f, total, 0, 0., lds, &nUndef, a.sumAlgo);
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

    const double dev = scan_order_sum(f, total, 1, average, lds, nullptr, a.sumAlgo);
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
// Two geometries are built: 16 waves on 16-column chunks (more bands in flight, the faster first-guess sums: small
// batches and calls that leave after a few sweeps) and 8 waves on 32-column chunks (whole 128-byte row pieces per
// memory event: large batches, where the chunk traffic of all slices meets in L2).  FILL_GEOMETRY(CH, WAVES) puts the
// derived constants into the scope of a function template.
#define FILL_GEOMETRY(CH, WAVES)                                                                                        \
    constexpr int kCh = (CH);                  /* skewed columns per chunk: one global-memory event per chunk */      \
    constexpr int kV2Waves = (WAVES);                                                                                   \
    constexpr int kV2Threads = kV2Waves * kWave;                                                                        \
    constexpr int kRowsPerIt = kWave / kCh;    /* rows one wave-wide load / store of a chunk covers */                \
    constexpr int kChunksPerWord = 32 / kCh;   /* chunks per 32-column mask word */                                   \
    constexpr int kRingW = 2 * kCh;            /* ring width (two chunks) */                                          \
    constexpr int kPitch = kRingW + 1;         /* conflict-free: bank = (lane + x') mod 32 */                         \
    (void)kV2Threads; (void)kRowsPerIt; (void)kChunksPerWord; (void)kRingW; (void)kPitch
constexpr int kMaxBands = 4096;

// ---- what precedes the sweeps of both systolic kernels, as kernels of their own: the sums stay one workgroup per slice
// (the reference's order of additions), the first guess and the mask words are spread over the chip -- one workgroup
// streaming a 36 MB slice is latency bound (4 ms of a 15 ms call before the split).
struct FillStatsArgs {
    const float* field;
    SliceStats* stats;
    size_t total;
    int wantDeviation;   // fill2d: second pass for the convergence criterion (:1284-1302)
    int useDefault;      // creepfillval2d: the caller's value is the first guess, only the undefined cells are counted
    float defaultVal;
    float relaxCrit;
    int sumAlgo;
    const double* defaults;  // per slice, instead of defaultVal (the rectangles of a decomposed creep fill: the whole slice's average)
    const unsigned long long* bounds;  // per slice, with defaults: SliceStats::sweepBound
    const double* devs;                // per slice, with defaults: SliceStats::meanAbsDev (fill2d by rectangles: the whole field's criterion)
};

__global__ void __launch_bounds__(kFillBlock) fill_stats_kernel(FillStatsArgs a)
{
    __shared__ __align__(16) double lds[2 * kSumTile];
    __shared__ double shAverage;
    __shared__ unsigned long long shUndef;
    const float* f = a.field + (size_t)blockIdx.x * a.total;
    SliceStats* st = a.stats + blockIdx.x;
    unsigned long long nUndef = 0;
    const double sum = scan_order_sum(f, a.total, a.useDefault ? 2 : 0, 0., lds, &nUndef, a.sumAlgo);
    if (threadIdx.x == 0) {
        const unsigned long long nDef = a.total - nUndef;
        shUndef = nUndef;
        shAverage = a.defaults ? a.defaults[blockIdx.x] : (a.useDefault ? (double)a.defaultVal : ((nDef != 0) ? sum / (double)nDef : 0.));  // :1281, :1516
        st->nUndef = nUndef;
        st->average = shAverage;
        st->status = 1;
        st->skip = (nDef == 0 || nUndef == 0);  // :1266-1269, :1384-1386
        st->sweepBound = a.bounds ? a.bounds[blockIdx.x] : nDef;
    }
    __syncthreads();
    nUndef = shUndef;
    const unsigned long long nDef = a.total - nUndef;
    if (a.devs) { if (threadIdx.x == 0) st->meanAbsDev = a.devs[blockIdx.x]; return; }
    if (!a.wantDeviation || nDef == 0 || nUndef == 0) return;
    const double dev = scan_order_sum(f, a.total, 1, shAverage, lds, nullptr, a.sumAlgo);
    if (threadIdx.x == 0) st->meanAbsDev = (double)a.relaxCrit * (dev / (double)nDef);  // :1302
}

struct FirstGuessArgs {
    float* field;
    const SliceStats* stats;
    uint32_t* mask;           // [nz][ny][mws]: fill2d NaN bits of the interior rows, creepfill "defined" bits of all rows
    unsigned char* mbRows;    // fill2d: [nz][2][nx] NaN mask of row 0 and row ny - 1
    unsigned char* mbCols;    // fill2d: [nz][2][ny] NaN mask of column 0 and column nx - 1
    uint32_t nx, ny, mws;
    uint32_t blocksPerSlice;
};

// One wave per row: undefined cells take the first guess (:1288-1299, :1408-1421) and the mask words are written in the
// row's skew (interior row y: bit x + ((y - 1) & 63)), eight row pieces in flight per wave.
template <bool CREEP>
__global__ void __launch_bounds__(kBlock) first_guess_kernel(FirstGuessArgs a)
{
    const uint32_t slice = blockIdx.x / a.blocksPerSlice;  // flat grid: gridDim.y stops at 65535 slices
    const SliceStats st = a.stats[slice];
    if (st.skip) return;
    const uint32_t nx = a.nx, ny = a.ny, mws = a.mws;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t y = (blockIdx.x % a.blocksPerSlice) * (kBlock / kWave) + threadIdx.x / kWave;
    if (y >= ny) return;
    const float guess = (float)st.average;
    float* row = a.field + ((size_t)slice * ny + y) * nx;
    const bool edgeRow = y == 0 || y == ny - 1;
    if (!CREEP && edgeRow) {
        unsigned char* mb = a.mbRows + ((size_t)slice * 2 + (y == 0 ? 0 : 1)) * nx;
        for (uint32_t x = lane; x < nx; x += kWave) {
            const bool u = isnan(row[x]);
            mb[x] = u;
            if (u) row[x] = guess;
        }
        return;
    }
    const uint32_t l = edgeRow ? 0u : ((y - 1) & (kWave - 1));
    uint32_t* mrow = a.mask + ((size_t)slice * ny + y) * mws;
    unsigned char* mbLeft = CREEP ? nullptr : a.mbCols + (size_t)slice * 2 * ny;
    constexpr int kAhead = 8;
    for (uint32_t base0 = 0; base0 < mws * 32; base0 += kAhead * kWave) {
        float v[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int64_t x = (int64_t)base0 + k * kWave + lane - l;
            v[k] = (x >= 0 && x < (int64_t)nx && base0 + k * kWave < mws * 32) ? row[x] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const uint32_t base = base0 + k * kWave;
            if (base >= mws * 32) break;
            const int64_t x = (int64_t)base + lane - l;
            const bool in = x >= 0 && x < (int64_t)nx;
            const bool u = in && isnan(v[k]);
            const unsigned long long m = __ballot(CREEP ? (in && !u) : u);
            if (lane == 0) {
                mrow[base / 32] = (uint32_t)m;
                if (base / 32 + 1 < mws) mrow[base / 32 + 1] = (uint32_t)(m >> 32);
            }
            if (u) row[x] = guess;
            if (!CREEP) {
                if (in && x == 0) mbLeft[y] = u;
                if (in && x == (int64_t)nx - 1) mbLeft[ny + y] = u;
            }
        }
    }
}

void launch_fill_prologue(bool creep, float* d_field, SliceStats* d_stats, size_t nx, size_t ny, size_t nz, uint32_t* mask, uint32_t mws,
                          unsigned char* mbRows, unsigned char* mbCols, bool wantDeviation, bool useDefault, float defaultVal, float relaxCrit,
                          hipStream_t stream, const double* d_defaults = nullptr, const unsigned long long* d_bounds = nullptr,
                          const double* d_devs = nullptr)
{
    FillStatsArgs s{};
    s.defaults = d_defaults;
    s.bounds = d_bounds;
    s.devs = d_devs;
    s.field = d_field;
    s.stats = d_stats;
    s.total = nx * ny;
    s.wantDeviation = wantDeviation;
    s.useDefault = useDefault;
    s.defaultVal = defaultVal;
    s.relaxCrit = relaxCrit;
    // few slices: the sums over the whole chip (two reads of the data per sum, but 0.5 instead of 2.3 ms per 9 M-cell
    // pass); many slices: one workgroup per slice fills the chip already and reads the data once
    s.sumAlgo = tuning("SUM_ALGO", 3);
    if (s.sumAlgo == 3) s.sumAlgo = nz < (size_t)tuning("SUM_CHIP_NZ", 100) ? 2 : 1;
    if (d_defaults) s.sumAlgo = 1;  // only the undefined cells are counted
    if (s.sumAlgo >= 2) {
        const SumBuffers buffers(nx * ny, nz);
        StitchOut o{};
        o.stats = d_stats;
        o.total = nx * ny;
        o.useDefault = useDefault;
        o.defaultVal = defaultVal;
        o.relaxCrit = relaxCrit;
        SumJob first{d_field, nx * ny, useDefault ? 2 : 0, d_stats, 0.};
        launch_chip_sum(first, buffers, nz, o, stream);
        if (wantDeviation) {
            SumJob second{d_field, nx * ny, 1, d_stats, 0.};
            launch_chip_sum(second, buffers, nz, o, stream);
        }
        FA_HIP(hipStreamSynchronize(stream));  // the work arrays are released on return
    } else {
        fill_stats_kernel<<<dim3((uint32_t)nz), kFillBlock, 0, stream>>>(s);
        FA_HIP(hipGetLastError());
    }
    FirstGuessArgs g{};
    g.field = d_field;
    g.stats = d_stats;
    g.mask = mask;
    g.mbRows = mbRows;
    g.mbCols = mbCols;
    g.nx = (uint32_t)nx;
    g.ny = (uint32_t)ny;
    g.mws = mws;
    const size_t rowBlocks = ceil_div(ny, (size_t)(kBlock / kWave));
    FA_REQUIRE(rowBlocks * nz <= 0x7FFFFFFFull, "too many slices for one call");
    g.blocksPerSlice = (uint32_t)rowBlocks;
    const dim3 grid((uint32_t)(rowBlocks * nz));
    if (creep) first_guess_kernel<true><<<grid, kBlock, 0, stream>>>(g);
    else first_guess_kernel<false><<<grid, kBlock, 0, stream>>>(g);
    FA_HIP(hipGetLastError());
}

struct Fill2dV2Args {
    float* field;
    uint32_t* maskS;          // [nz][ny][mws] skewed NaN-mask words of the interior rows
    unsigned char* mbRows;    // [nz][2][nx] NaN mask of row 0 and row ny-1
    unsigned char* mbCols;    // [nz][2][ny] NaN mask of column 0 and column nx-1
    SliceStats* stats;
    uint32_t nx, ny, mws;
    float relaxCrit, corrEff;
    unsigned long long maxLoop;
    int sumAlgo;
    unsigned int* error;      // one word per launch: set by a wait that gave up (see MultiWg)
    // several workgroups per slice (fill2d_kernel_v3): per slice [0] barrier counter, [1..2] "not converged" by parity of the
    // check, [4 .. 4 + bands) progress words of the bands whose hand-off crosses workgroups
    unsigned int* sync;
    uint32_t syncStride, groups, nz;
    uint32_t experiment;
    unsigned long long* prof;
    // > 0 (fill2d by rectangles): slices i, i + couple, i + 2 couple, ... are rectangles of ONE field and end their sweeps together,
    // by the criterion over all of them (:1338-1359); word [3] of slice i % couple's sync words is their barrier counter
    uint32_t couple;
};

// Flags of the LDS hand-off.  The LDS executes one wave's operations in issue order and is coherent within the CU, so a
// flag written after the data (and read before it) needs no fence -- and must not get one: a release / acquire at
// workgroup scope makes the compiler wait for ALL outstanding vector-memory operations (s_waitcnt vmcnt(0)), i.e. for the
// chunk prefetch that was issued a moment ago, once per event.  Compiler barriers keep the program order.
__device__ __forceinline__ void lds_publish(unsigned int* flag, unsigned int value)
{
    asm volatile("" ::: "memory");
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned int lds_observe(const unsigned int* flag)
{
    asm volatile("" ::: "memory");
    const unsigned int v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    return v;
}

// value of lane l-1 (lane 0 keeps its own): one DPP move, "wave_shr:1" (0x138), no LDS round trip
__device__ __forceinline__ float lane_from_above(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
// the same with lane 0 (which has no lane above) receiving `first`: the DPP move leaves lanes without a source at the
// old value of the destination, so the separate select for lane 0 is not needed
__device__ __forceinline__ float lane_from_above_or(float v, float first)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
// value held by lane `idx` (wave-uniform index) broadcast through an SGPR
__device__ __forceinline__ float lane_value(float v, int idx)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), idx));
}

// ---- several workgroups per slice (small batches): the bands of one slice are dealt to G workgroups, W = waves per workgroup
// at a time (bands 0 .. W-1 to workgroup 0, W .. 2W-1 to workgroup 1, ...), so that a batch of 16 slices uses 96 CUs instead of
// 16.  Inside a workgroup nothing changes; the hand-off of every W-th band boundary, which already went through global memory,
// now crosses workgroups: the producer's stores of that band are write-through (sc0 sc1), it publishes its progress in a
// global word after s_waitcnt vmcnt(0) (relaxed agent-scope store = sc1), the consumer polls that word and reads the row
// above with sc0 sc1 loads (MI355X_MICROARCH.md, inter-workgroup visibility: every store and every load of the handed-off
// bytes bypasses the non-coherent caches).  The sweeps of the workgroups of a slice are separated by a barrier on a global
// counter with agent-scope release / acquire, which makes everything else (the row below a band, the border columns) visible.
// Every wait is bounded: a spin that exceeds its cap sets the launch's error word, every other wait then falls through, the
// kernel ends and the host call fails with a message -- a wrong counter cannot hang the GPU.
struct MultiWg {
    uint32_t g, G;            // this workgroup and the number of workgroups of its slice (1: the single-workgroup kernels)
    uint32_t experiment = 0;  // tuning build: 1 = the producer does not wait for its stores (timing experiment, results invalid)
    unsigned int* flags;      // [bands] progress of the bands whose hand-off crosses workgroups: columns final + 1
    unsigned int* error;      // one word per launch
    unsigned long long* prof = nullptr;  // tuning build, experiment 4: cycles summed over bands: [0] events, [1] steps, [2] bands
};
// A wait gives up after kSpinCapTicks of WALL time (s_memrealtime, the 100 MHz constant clock): long enough that workgroups
// kept off their CUs by other work on the device -- a concurrent one-workgroup-per-slice fill of a long batch, another process --
// still arrive (they are queued behind that work, not lost), short enough that a wrong counter ends the call instead of hanging
// the GPU.  The clock is read only every 4096th (256th) poll, for the first time after that many polls: a wait that ends
// quickly never reads it.
constexpr unsigned long long kSpinCapTicks = 30ull * 100000000ull;  // 30 s
__device__ __forceinline__ bool spin_expired(unsigned long long& t0)
{
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (t0 == 0) { t0 = now | 1ull; return false; }
    return now - t0 > kSpinCapTicks;
}

__device__ __forceinline__ bool launch_failed(const unsigned int* error)
{
    return __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__device__ __forceinline__ void fail_launch(unsigned int* error, unsigned int code)
{
    __hip_atomic_store(error, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// waits until the LDS word reaches `need`; false: gave up (cap or another wave's failure)
__device__ __forceinline__ bool wait_lds_at_least(const unsigned int* flag, unsigned int need, unsigned int* error)
{
    unsigned long long t0 = 0;
    for (unsigned int it = 0;; ++it) {
        asm volatile("" ::: "memory");
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) { asm volatile("" ::: "memory"); return true; }
        __builtin_amdgcn_s_sleep(1);
        if ((it & 0xFFF) == 0xFFF && (spin_expired(t0) || launch_failed(error))) { fail_launch(error, 1); return false; }
    }
}
__device__ __forceinline__ bool wait_global_at_least(const unsigned int* flag, unsigned int need, unsigned int* error)
{
    unsigned long long t0 = 0;
    for (unsigned int it = 0;; ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { asm volatile("" ::: "memory"); return true; }
        __builtin_amdgcn_s_sleep(2);
        if ((it & 0xFF) == 0xFF && (spin_expired(t0) || launch_failed(error))) { fail_launch(error, 2); return false; }
    }
}
// barrier of the G workgroups of one slice on a monotone global counter (instance k waits for k * G arrivals)
__device__ __forceinline__ void slice_barrier(unsigned int* counter, unsigned int target, unsigned int* error)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wait_global_at_least(counter, target, error);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
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

// e = (f1 + f2 + f3 + f4) * 0.25 - f of interpolation.c:1332: the float sum times the double constant, minus the float
// as double, rounded to float.  One fused multiply-add gives the same bits with a third of the dependent operations: the
// product is exact either way, and the difference of two floats either fits a double exactly (exponents at most 29
// apart: then both paths round the same exact value once) or is dominated by the larger one so completely that both
// round to it; infinities and NaN take the same way through both.  (-ffp-contract=off forbids the compiler to fuse on
// its own; this fusion is deliberate.)  tests/test_gpu_parity.py::test_sor_error_is_the_reference_expression walks the
// exponent gaps.
__device__ __forceinline__ float sor_error(float sum, float center) { return __builtin_fmaf(sum, 0.25f, -center); }

// one band of one sweep, executed by one wave.  Global memory is touched only in the "event" between two 16-step
// chunks: loads issued there are consumed one event later, stores are never waited for (the sweep ends with a
// workgroup barrier); the 16 steps in between run on registers and LDS.
// CHECK: a sweep that also tests convergence (every tenth, :1339-1360); the other nine carry no trace of the test
template <int CH, int WAVES, bool CHECK, bool MULTI = false>
__device__ void fill2d_band(float* __restrict__ f, const uint32_t* __restrict__ maskS, float* ring, Handoff hand, uint32_t b,
                            uint32_t nx, uint32_t ny, uint32_t mws, float wInt, float wZero, float crtest, int& bad, MultiWg mg)
{
    FILL_GEOMETRY(CH, WAVES);
    constexpr bool check = CHECK;
    using rsrc_t = __amdgpu_buffer_rsrc_t;
    // The band is the same for the whole wave, but it derives from threadIdx: said explicitly, the buffer descriptor of the
    // band's rows and every "is this the band that hands over through global memory" test stay in scalar registers.  Left to
    // the compiler, each of the 32 buffer loads and stores of an event sat in a loop over the lanes' (identical) descriptors
    // and each store behind a divergent branch: 7 000 cycles per event against 1 900 for the sixteen steps between two events.
    b = __builtin_amdgcn_readfirstlane(b);
    {   // (where this function is not inlined its arguments arrive in vector registers: the same for the slice's pointers)
        auto uniform = [](auto* ptr) {
            const uint64_t v = reinterpret_cast<uint64_t>(ptr);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
            return reinterpret_cast<decltype(ptr)>(((uint64_t)hi << 32) | lo);
        };
        f = uniform(f);
        maskS = uniform(maskS);
        nx = __builtin_amdgcn_readfirstlane(nx);
        ny = __builtin_amdgcn_readfirstlane(ny);
        mws = __builtin_amdgcn_readfirstlane(mws);
    }
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t y0 = 1 + kWave * b;
    const uint32_t nrow = min((uint32_t)kWave, (ny - 1) - y0);  // rows y0 .. y0 + nrow - 1 <= ny - 2
    const uint32_t L = nrow - 1;                                 // last lane with a row
    const bool rowValid = lane < nrow;
    const uint32_t y = y0 + min(lane, L);
    const uint32_t C = nx - 2;                                   // interior columns 1 .. C
    const uint32_t xpEnd = C + L;                                // last skewed column with work
    float* ringRow = ring + lane * kPitch;
    // the ring's row below lane L (row nrow <= 64) is virtual: the first row of the band below, written chunk by chunk from
    // the 64-column blocks downA / downB, so that the last lane reads its "down" like every other lane
    const float* ringBelow = ring + min(lane + 1, L + 1) * kPitch;
    float* ringVirtual = ring + (L + 1) * kPitch;
    const float left0 = f[(size_t)y * nx];                       // border column 0, not touched by the sweep
    const uint32_t* mrow = maskS + (size_t)y * mws;
    // the band's rows plus the row above and the row below as one buffer: masked lanes use an out-of-range offset
    // (loads return 0, stores are dropped), so every memory instruction is issued unconditionally
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(f + (size_t)(y0 - 1) * nx, 0, (nrow + 2) * nx * 4u, 0x00020000);
    const uint32_t kOob = 0xFFFFFFFFu;
    const bool prof = kTuningBuild && mg.experiment == 4 && mg.prof != nullptr;
    const bool earlyPrefetch = !(kTuningBuild && mg.experiment == 5);  // tuning build, FILL_EXPERIMENT=5: the prefetch after the waits (round 2's order)
    unsigned long long tWait = 0, tFlush = 0;
    // Every W-th boundary (band W-1 -> W, 2W-1 -> 2W, ...) goes through global memory: in one workgroup the wave of band b + 1
    // is still busy with band b + 1 - W there (a bounded LDS window would close a cycle of waiting waves on wide grids), with
    // several workgroups per slice (MULTI) band b + 1 belongs to the next workgroup.  The producer's flush already writes the
    // row; it only has to publish how far its stores have completed.
    const bool hasBelow = y0 + nrow < ny - 1;
    const bool outGlobal = hasBelow && (b % kV2Waves) == kV2Waves - 1;
    const bool inGlobal = b > 0 && (b % kV2Waves) == 0;
    const bool writeThrough = MULTI && outGlobal;  // the band below reads these rows on another CU, maybe another XCD

    // chunk c = skewed columns [c*kCh, c*kCh + kCh) of all 64 rows; lane -> (row kRowsPerIt*it + lane/kCh, column lane%kCh)
    const uint32_t crow = lane / kCh, ccol = lane % kCh;
    float stage[kCh];
    auto chunk_off = [&](uint32_t c, uint32_t it, bool store) -> uint32_t {
        const uint32_t row = kRowsPerIt * it + crow;
        const int64_t x = (int64_t)c * kCh + ccol - row;  // unskewed column
        const bool ok = row < nrow && (store ? (x >= 1 && x <= (int64_t)C) : (x >= 0 && x <= (int64_t)nx - 1));
        return ok ? (uint32_t)(((row + 1) * nx + x) * 4u) : kOob;
    };
    // A chunk is "interior" when every one of its 64 x 16 cells is a cell the sweep updates (all rows of the band exist,
    // 1 <= x <= C for all of them): no per-lane conditions are needed then, and since a lone wave issues roughly one
    // instruction per 8-9 clocks, instructions are what the band's time consists of.  Interior chunks address memory as
    // one per-lane offset plus a scalar offset per row group (buffer soffset) and the LDS ring with immediate offsets.
    auto interior = [&](uint32_t c) -> bool { return nrow == (uint32_t)kWave && c * kCh >= (uint32_t)kWave && c * kCh + kCh - 1 <= C; };
    const uint32_t voffLane = ((crow + 1) * nx + ccol - crow) * 4u;           // row crow, chunk 0, column ccol - crow
    const uint32_t rowStep = (uint32_t)kRowsPerIt * (nx - 1) * 4u;            // next row group: kRowsPerIt rows down, as many columns back
    float* ringLane = ring + crow * kPitch + ccol;
    auto load_chunk = [&](uint32_t c) {
        if (kTuningBuild && mg.experiment == 2) return;  // timing experiment: no chunk loads
        if (interior(c)) {
            const uint32_t s0 = c * kCh * 4u;
#pragma unroll
            for (uint32_t it = 0; it < (uint32_t)kCh; ++it)
                stage[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voffLane, s0 + it * rowStep, 0));
            return;
        }
#pragma unroll
        for (uint32_t it = 0; it < (uint32_t)kCh; ++it)
            stage[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, chunk_off(c, it, false), 0, 0));
    };
    auto commit_chunk = [&](uint32_t c) {
        float* dst = ringLane + ((c * kCh) & kCh);
#pragma unroll
        for (uint32_t it = 0; it < (uint32_t)kCh; ++it) dst[kRowsPerIt * it * kPitch] = stage[it];
    };
    // The results of a finished chunk leave in two steps: out of the ring into registers (before the ring slot is refilled),
    // and to global memory AFTER the event's loads have been issued.  Vector-memory operations complete in issue order: a load
    // that sits behind sixteen stores is not counted as done before they are, so the wait for the next chunk's data was a wait
    // for this chunk's stores -- 3-4 us per event, which set the pace of every band (measured with the tuning build's cycle
    // counters: 8 400 cycles per event against 1 900 for the sixteen steps between two events).
    float v[kCh];
    auto flush_read = [&](uint32_t c) {
        const float* src = ringLane + ((c * kCh) & kCh);
#pragma unroll
        for (uint32_t it = 0; it < (uint32_t)kCh; ++it) v[it] = src[kRowsPerIt * it * kPitch];
    };
    auto flush_store = [&](uint32_t c) {
        if (kTuningBuild && mg.experiment == 3) return;  // timing experiment: no chunk stores
        if (interior(c)) {
            const uint32_t s0 = c * kCh * 4u;
#pragma unroll
            for (uint32_t it = 0; it < (uint32_t)kCh; ++it) {
                if (writeThrough) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[it]), rs, voffLane, s0 + it * rowStep, 17);
                else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[it]), rs, voffLane, s0 + it * rowStep, 0);
            }
            return;
        }
#pragma unroll
        for (uint32_t it = 0; it < (uint32_t)kCh; ++it) {
            if (writeThrough) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[it]), rs, chunk_off(c, it, true), 0, 17);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[it]), rs, chunk_off(c, it, true), 0, 0);
        }
    };
    auto load_block = [&](uint32_t rowInBuf, uint32_t k) {  // 64 columns of the row above (0) / below (nrow + 1)
        const uint32_t col = 64 * k + lane;
        const uint32_t off = col <= nx - 1 ? (rowInBuf * nx + col) * 4u : kOob;
        // the row above a band whose predecessor runs in another workgroup: written during this sweep, read past L1 and L2
        if (MULTI && inGlobal && rowInBuf == 0) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 17));
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
    };

    // hand-off slots: mine (towards band b + 1) and the one of band b - 1
    const uint32_t round = kV2Waves * (MULTI ? mg.G : 1u);  // bands between two bands of one wave
    const uint32_t slotOut = (b % kV2Waves) * 2 + ((b / round) & 1);
    const uint32_t slotIn = ((b - 1) % kV2Waves) * 2 + (((b - 1) / round) & 1);  // unused for b == 0
    float* handOut = hand.data + slotOut * kHandW;
    const float* handIn = hand.data + slotIn * kHandW;
    if (lane == 0) {
        __hip_atomic_store(&hand.produced[slotOut], hand_tag(b, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&hand.consumed[slotOut], hand_tag(b, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // The hand-off advances chunk by chunk, not in blocks of 64 columns: band b runs behind band b - 1 by the 63 columns of the
    // skew plus one or two chunks, and these lags add up over all bands of a sweep -- the critical path of a small batch.
    auto wait_above = [&](uint32_t cols) {  // the stores of band b - 1 have completed for columns < cols of its last row
        cols = min(cols, C + 1);
        const unsigned long long t0 = prof ? clock64() : 0;
        if (MULTI) wait_global_at_least(&mg.flags[b - 1], cols + 1, mg.error);
        else wait_lds_at_least(&hand.produced[slotIn], hand_tag(b - 1, cols), mg.error);
        if (prof) tWait += clock64() - t0;
    };
    // columns [xpc, xpc + kCh) of the row above from the LDS hand-off of band b - 1, into lanes (column % 64) of upCur
    auto take_above = [&](uint32_t xpc, float& upCur) {
        const unsigned int need = hand_tag(b - 1, min(xpc + kCh, C + 1));
        // a larger band tag means the producer has finished band b - 1 long ago (its data stay in the other parity slot)
        const unsigned long long t0 = prof ? clock64() : 0;
        wait_lds_at_least(&hand.produced[slotIn], need, mg.error);
        if (prof) tWait += clock64() - t0;
        const float v = handIn[((xpc & ~63u) + lane) % kHandW];
        if (lane - (xpc & 63u) < (uint32_t)kCh) upCur = v;
        if (lane == 0)
            lds_publish(&hand.consumed[slotIn], hand_tag(b - 1, xpc + kCh));
    };

    // ---- prologue: chunks 0 and 1 in LDS, chunk 2 in flight; first blocks and mask words.
    // Registers that receive a load at an event (stage[], upLd, downLd, mwLd) are read only at a LATER event.
    load_chunk(0);
    commit_chunk(0);
    load_chunk(1);
    commit_chunk(1);
    load_chunk(2);
    // row above: lanes (column % 64) of upCur hold the chunk in work; from global memory (row 0, or a band whose predecessor
    // hands over through global memory) the next chunk's block is requested one event ahead into upLd
    float upCur = 0.f, upLd = 0.f;
    const bool fromGlobal = b == 0 || inGlobal;
    if (fromGlobal) {
        if (inGlobal) wait_above(kCh);
        upCur = load_block(0, 0);
        if (inGlobal) wait_above(2 * kCh);
        upLd = load_block(0, kCh >> 6);
    } else take_above(0, upCur);
    float downA = load_block(nrow + 1, 0), downB = downA, downLd = 0.f;  // current / next (landed) / in flight
    uint32_t downIssued = 0;
    bool downLdValid = false;
    uint32_t mw = mrow[0], mwN = mrow[1], mwLd = mrow[2];
    float prevRes = 0.f;
    float prevRight = ringRow[1];  // lane 0 is at column 1 in the first step: its centre is skewed column 1

    const uint32_t nChunks = xpEnd / kCh + 1;
    unsigned long long tEvents = 0, tSteps = 0, tMark = prof ? clock64() : 0;
    for (uint32_t c = 0; c < nChunks; ++c) {
        const uint32_t xpc = c * kCh;
        if (prof) { const unsigned long long t = clock64(); tSteps += t - tMark; tMark = t; }
        if (c > 0) {
            // ---- event at the start of chunk c
            if (outGlobal && xpc > L) {  // stores of the previous event (chunk c - 2) have landed: columns < 16 (c - 1) - L of the last row
                if (!(kTuningBuild && mg.experiment == 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0 && xpc - kCh > L) {
                    if (MULTI) __hip_atomic_store(&mg.flags[b], xpc - kCh - L + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else lds_publish(&hand.produced[slotOut], hand_tag(b, xpc - kCh - L));
                }
            }
            // Order matters: vector-memory results come back in issue order, so waiting for a load also waits for every
            // load issued before it.  The small loads (mask word, 64-column blocks of the rows above / below) go first,
            // the 16 loads of the chunk prefetch last: whatever the compiler makes of the small ones, it never has to wait
            // for the prefetch before the next event.
            if ((c % kChunksPerWord) == 0) {      // x' is a multiple of 32: every lane switches mask words now
                mw = mwN;
                mwN = mwLd;
                mwLd = mrow[min(c / kChunksPerWord + 2, mws - 1)];
            }
            // row below: the last lane is at column x' - L
            if (downLdValid) { downB = downLd; downLdValid = false; }
            if (xpc + 2 * kCh > L && ((xpc + 2 * kCh - L) >> 6) > downIssued) {  // block j + 1 is requested two chunks before the last lane
                // reaches it and lands in downB at the next event: after block j has moved on to downA, never skipping one
                ++downIssued;
                downLd = load_block(nrow + 1, downIssued);
                downLdValid = true;
            }
            // row above: lane 0 is at column x'
            if (fromGlobal) {
                if (lane - (xpc & 63u) < (uint32_t)kCh) upCur = upLd;  // this chunk's columns, requested one event ago
                if (inGlobal) wait_above(xpc + 2 * kCh);
                upLd = load_block(0, (xpc + kCh) >> 6);
            }
            flush_read(c - 1);       // results of the chunk just finished: ring -> registers
            commit_chunk(c + 1);     // loaded one event ago, into the ring slot the flush has just read
            // The prefetch of chunk c + 2 goes out as soon as its registers are free -- before the waits on the neighbouring bands
            // below, which may take as long as the 16 steps do: the loads are what the next event waits for.
            if (earlyPrefetch) load_chunk(c + 2);
            // publish how far the last row has got, and do not run more than the hand-off window ahead of the band below
            if (xpc > L && !outGlobal) {
                if (lane == 0)
                    lds_publish(&hand.produced[slotOut], hand_tag(b, xpc - L));
                if (hasBelow) {
                    const unsigned int limit = xpc + kCh - L;  // columns < limit are written during this chunk
                    const unsigned long long t0 = prof ? clock64() : 0;
                    unsigned long long tSpin = 0;
                    for (unsigned int it = 0;; ++it) {
                        const unsigned int cns = lds_observe(&hand.consumed[slotOut]);
                        if (limit <= (cns & 0x7FFFFu) + kHandW) break;
                        __builtin_amdgcn_s_sleep(1);
                        if ((it & 0xFFF) == 0xFFF && (spin_expired(tSpin) || launch_failed(mg.error))) { fail_launch(mg.error, 3); break; }
                    }
                    if (prof) tFlush += clock64() - t0;  // (window waits, counted apart)
                }
            }
            if (!fromGlobal) take_above(xpc, upCur);
            if (!earlyPrefetch) load_chunk(c + 2);       // consumed at the next event
            flush_store(c - 1);      // registers -> global, behind the loads (never waited for)
        }
        if (prof) { const unsigned long long t = clock64(); tEvents += t - tMark; tMark = t; }
        const uint32_t xp0 = max(xpc, 1u), xp1 = min(xpc + kCh - 1, xpEnd);
        if (interior(c) && xpc > (uint32_t)kWave) {
            // ---- every lane is at a cell the sweep updates, at x >= 2: no range tests, "left" is the previous result
            const uint32_t half = xpc & kCh;
            float* rc = ringRow + half;
            const float* rb = ringBelow + half;
            const uint32_t rNext = (half ^ kCh);
            const uint32_t sh0 = xpc & 31, up0 = xpc & 63;
            const uint32_t kSwitch = (L - xpc) & 63;
            const bool switches = kSwitch < (uint32_t)kCh;  // xpc + kSwitch > L holds: xpc >= 64 > L - kSwitch
            const int dBase = (int)((xpc - L) & 63);
            {   // the chunk's kCh values of the row below the band into the virtual ring row (lane j: step j)
                const uint32_t j = lane & (kCh - 1);
                const int src = (int)((dBase + j) & 63);   // (the choice of the block is the reading lane's, not the source lane's)
                const float fromA = __shfl(downA, src), fromB = __shfl(downB, src);  // by all lanes: a shuffle under a divergent branch misses its source lanes
                const float vd = (switches && j >= kSwitch) ? fromB : fromA;
                if (lane < (uint32_t)kCh) ringVirtual[(xpc + 1 + j) & (kRingW - 1)] = vd;
            }
            // Everything a group of 16 steps reads from the ring is read first -- a position is read (as "right" / "down" of
            // the step before) strictly before the step that rewrites it, so the values are the ones the interleaved order saw
            // -- and the results are written after the group's last step: the dependent chain of a step (the lane above's
            // previous result -> sum -> error -> result) then runs on registers, DPP and readlane alone, with no LDS latency
            // in it.  The chain is what a band's time consists of, and the bands' lags add up to a sweep's critical path.
            // (Workgroups of 16 waves have 128 registers per lane and four waves per SIMD to cover the latency: interleaved.)
            constexpr bool kPreload = WAVES <= 8;
            constexpr int kGroup = kPreload ? 16 : 1;
#pragma unroll
            for (int g0 = 0; g0 < kCh; g0 += kGroup) {
                float R[kGroup], D[kGroup], out[kGroup];
#pragma unroll
                for (int q = 0; q < kGroup; ++q) {
                    const int k = g0 + q;
                    R[q] = (k < kCh - 1) ? rc[k + 1] : ringRow[rNext];
                    D[q] = (k < kCh - 1) ? rb[k + 1] : ringBelow[rNext];
                }
#pragma unroll
                for (int q = 0; q < kGroup; ++q) {
                    const int k = g0 + q;
                    const float center = prevRight;
                    const float up = lane_from_above_or(prevRes, lane_value(upCur, (int)(up0 + k)));
                    const float wv = ((mw >> (sh0 + k)) & 1u) ? wInt : wZero;
                    const float e = sor_error(((R[q] + prevRes) + D[q]) + up, center);  // interpolation.c:1332
                    const float res = center + e * wv;                                                           // :1333
                    out[q] = res;
                    prevRes = res;
                    if (check && (fabsf(e * wv) > crtest)) bad = 1;                                              // :1349
                    prevRight = R[q];
                }
#pragma unroll
                for (int q = 0; q < kGroup; ++q) rc[g0 + q] = out[q];
            }
            if (switches) downA = downB;
            if (lane < (uint32_t)kCh) handOut[(xpc + lane - L) % kHandW] = ring[L * kPitch + ((xpc + lane) & (kRingW - 1))];
            continue;
        }
        if (xpc >= 1 && xpc + kCh - 1 <= xpEnd) {
            // ---- a whole chunk: 16 steps unrolled, everything that is the same for all lanes in scalar registers, no
            // branches (inactive lanes rewrite the ring slot with its own value), hand-off copied once at the end
            const uint32_t half = xpc & kCh;                                   // ring half this chunk lives in
            float* rc = ringRow + half;
            const float* rb = ringBelow + half;
            const uint32_t rNext = (half ^ kCh);                               // first column of the other half
            const int x0 = (int)xpc - (int)lane;
            const uint32_t sh0 = xpc & 31, up0 = xpc & 63;
            const uint32_t kSwitch = (L - xpc) & 63;                           // step at which the last lane enters the next 64-column block
            const bool switches = kSwitch < (uint32_t)kCh && xpc + kSwitch > L;
            const int dBase = (int)((xpc - L) & 63);                           // column of the last lane within its block (valid when xpc >= L)
            {   // the virtual ring row, as in the interior path (columns before the last lane's first one are never used)
                const uint32_t j = lane & (kCh - 1);
                const int src = (int)((xpc + j >= L) ? ((dBase + j) & 63) : 0);
                const float fromA = __shfl(downA, src), fromB = __shfl(downB, src);  // by all lanes: a shuffle under a divergent branch misses its source lanes
                const float vd = (switches && j >= kSwitch) ? fromB : fromA;
                if (lane < (uint32_t)kCh) ringVirtual[(xpc + 1 + j) & (kRingW - 1)] = vd;
            }
#pragma unroll
            for (int k = 0; k < kCh; ++k) {
                const int x = x0 + k;
                const bool active = rowValid && x >= 1 && x <= (int)C;
                const float right = (k < kCh - 1) ? rc[k + 1] : ringRow[rNext];
                const float down = (k < kCh - 1) ? rb[k + 1] : ringBelow[rNext];
                const float center = prevRight;
                const float up = lane_from_above_or(prevRes, lane_value(upCur, (int)(up0 + k)));
                const float left = (x == 1) ? left0 : prevRes;
                const float wv = ((mw >> (sh0 + k)) & 1u) ? wInt : wZero;
                const float e = sor_error(((right + left) + down) + up, center);  // interpolation.c:1332
                const float res = active ? center + e * wv : center;                                      // :1333
                rc[k] = res;
                prevRes = res;
                if (check && active && (fabsf(e * wv) > crtest)) bad = 1;                                 // :1349
                prevRight = right;
            }
            if (switches) downA = downB;
            {   // the band below reads its "up" values from the hand-off: lanes 0..15 copy one column of the last row each
                const uint32_t xpk = xpc + lane;
                const int xk = (int)xpk - (int)L;
                if (lane < (uint32_t)kCh && xk >= 1 && xk <= (int)C) handOut[(uint32_t)xk % kHandW] = ring[L * kPitch + (xpk & (kRingW - 1))];
            }
            continue;
        }
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
            const float e = sor_error(((right + left) + down) + up, center);  // interpolation.c:1332
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
    if (prof && lane == 0) {
        atomicAdd(&mg.prof[0], tEvents);
        atomicAdd(&mg.prof[1], tSteps + (clock64() - tMark));
        atomicAdd(&mg.prof[2], 1ull);
        atomicAdd(&mg.prof[3], tWait);
        atomicAdd(&mg.prof[4], tFlush);
        if (b == 0) { atomicAdd(&mg.prof[5], tEvents); atomicAdd(&mg.prof[6], tSteps); atomicAdd(&mg.prof[7], 1ull); }
    }
    flush_read(nChunks - 1);
    flush_store(nChunks - 1);
    if (outGlobal) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        if (MULTI && outGlobal) __hip_atomic_store(&mg.flags[b], C + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_publish(&hand.produced[slotOut], hand_tag(b, C + 1));
    }
}

template <int CH, int WAVES>
__global__ void __launch_bounds__(WAVES * kWave) fill2d_kernel_v2(Fill2dV2Args a)
{
    FILL_GEOMETRY(CH, WAVES);
    extern __shared__ __attribute__((aligned(16))) float smem[];  // rings [16][64][33] floats, hand-off [16][2][192] + counters
    float* rings = smem;
    Handoff hand;
    hand.data = smem + kV2Waves * (kWave + 1) * kPitch;   // every wave's ring has a 65th row: the row below the band
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
    const uint32_t wave = threadIdx.x / kWave;

    // sums, first guess and masks were made by fill_stats_kernel / first_guess_kernel
    if (st->skip) return;  // :1266-1269
    const double crit = st->meanAbsDev;

    const float wInt = 1.f * a.corrEff, wZero = 0.f * a.corrEff;  // :1311-1315
    const float crtest = (float)(crit * a.corrEff);
    const uint32_t nxm1 = nx - 1, nym1 = ny - 1;
    const uint32_t nBands = (ny - 2 + kWave - 1) / kWave;
    float* ring = rings + wave * (kWave + 1) * kPitch;
    const MultiWg single{0u, 1u, 0u, nullptr, a.error};
    for (unsigned long long n = 0; n < a.maxLoop; ++n) {
        const bool check = (n < (a.maxLoop - 5)) && (n % 10 == 0);
        int bad = 0;
        if (threadIdx.x < kV2Waves * 2) { hand.produced[threadIdx.x] = 0; hand.consumed[threadIdx.x] = 0; }
        __syncthreads();
        for (uint32_t b = wave; b < nBands; b += kV2Waves)
            if (check) fill2d_band<CH, WAVES, true>(f, maskS, ring, hand, b, nx, ny, mws, wInt, wZero, crtest, bad, single);
            else fill2d_band<CH, WAVES, false>(f, maskS, ring, hand, b, nx, ny, mws, wInt, wZero, crtest, bad, single);
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

// The same sweeps with the bands of a slice dealt to a.groups workgroups (MultiWg above).  Workgroup i serves slice
// (i % 8) + 8 * (i / (8 * groups)) as its member (i / 8) % groups: the workgroups of a slice have the same i % 8, which is how
// workgroups are dealt to the XCDs today (a speed bonus for the hand-off, not a condition: the write-through stores and the
// loads past the caches hold on any placement).  Launched cooperatively: the workgroups of a slice wait for each other.
template <int CH, int WAVES>
__global__ void __launch_bounds__(WAVES * kWave) fill2d_kernel_v3(Fill2dV2Args a)
{
    FILL_GEOMETRY(CH, WAVES);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rings = smem;
    Handoff hand;
    hand.data = smem + kV2Waves * (kWave + 1) * kPitch;
    hand.produced = reinterpret_cast<unsigned int*>(hand.data + kV2Waves * 2 * kHandW);
    hand.consumed = hand.produced + kV2Waves * 2;
    const uint32_t G = a.groups;
    const uint32_t slice = (blockIdx.x % kXcds) + kXcds * (blockIdx.x / (kXcds * G));
    const uint32_t g = (blockIdx.x / kXcds) % G;
    if (slice >= a.nz) return;
    const uint32_t nx = a.nx, ny = a.ny, mws = a.mws;
    const size_t total = (size_t)nx * ny;
    float* f = a.field + (size_t)slice * total;
    uint32_t* maskS = a.maskS + (size_t)slice * ny * mws;
    unsigned char* mbTop = a.mbRows + (size_t)slice * 2 * nx;
    unsigned char* mbBot = mbTop + nx;
    unsigned char* mbLeft = a.mbCols + (size_t)slice * 2 * ny;
    unsigned char* mbRight = mbLeft + ny;
    SliceStats* st = a.stats + slice;
    unsigned int* sync = a.sync + (size_t)slice * a.syncStride;
    MultiWg mg{g, G, 0u, sync + 4, a.error};
    mg.experiment = a.experiment;
    mg.prof = a.prof;
    const uint32_t wave = threadIdx.x / kWave;
    if (st->skip) return;  // :1266-1269 (the same for every workgroup of the slice)
    const double crit = st->meanAbsDev;
    const float wInt = 1.f * a.corrEff, wZero = 0.f * a.corrEff;  // :1311-1315
    const float crtest = (float)(crit * a.corrEff);
    const uint32_t nxm1 = nx - 1, nym1 = ny - 1;
    const uint32_t nBands = (ny - 2 + kWave - 1) / kWave;
    float* ring = rings + wave * (kWave + 1) * kPitch;
    unsigned int barriers = 0, checks = 0, groupBarriers = 0;
    const uint32_t leader = a.couple ? slice % a.couple : slice;
    const uint32_t members = a.couple ? a.nz / a.couple : 1u;
    unsigned int* lsync = a.sync + (size_t)leader * a.syncStride;
    for (unsigned long long n = 0; n < a.maxLoop; ++n) {
        const bool check = (n < (a.maxLoop - 5)) && (n % 10 == 0);
        int bad = 0;
        if (threadIdx.x < kV2Waves * 2) { hand.produced[threadIdx.x] = 0; hand.consumed[threadIdx.x] = 0; }
        __syncthreads();
        for (uint32_t b = g * kV2Waves + wave; b < nBands; b += G * kV2Waves)
            if (check) fill2d_band<CH, WAVES, true, true>(f, maskS, ring, hand, b, nx, ny, mws, wInt, wZero, crtest, bad, mg);
            else fill2d_band<CH, WAVES, false, true>(f, maskS, ring, hand, b, nx, ny, mws, wInt, wZero, crtest, bad, mg);
        unsigned int* notConverged = lsync + 1 + (checks & 1);
        if (check) {
            if (__syncthreads_or(bad) && threadIdx.x == 0) __hip_atomic_fetch_or(notConverged, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        slice_barrier(sync, ++barriers * G, a.error);
        if (launch_failed(a.error)) return;
        if (check) {
            if (members > 1) {
                // the rectangles of one field: the word of the NEXT check is cleared before anybody can have passed this barrier
                // (it was read last at the previous check, which everybody has left behind), then all of them meet
                if (slice == leader && g == 0 && threadIdx.x == 0)
                    __hip_atomic_store(lsync + 1 + ((checks + 1) & 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                slice_barrier(lsync + 3, ++groupBarriers * members * G, a.error);
                if (launch_failed(a.error)) return;
            }
            if (__hip_atomic_load(notConverged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;  // converged (:1355-1359)
            ++checks;
            if (members == 1 && g == 0 && threadIdx.x == 0)  // the word of the check after next (read last ten sweeps ago)
                __hip_atomic_store(sync + 1 + (checks & 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // progress words of this workgroup's outgoing hand-offs: back to "nothing" for the next sweep
        for (uint32_t b = g * kV2Waves + kV2Waves - 1 + threadIdx.x * G * kV2Waves; b < nBands; b += kV2Threads * G * kV2Waves)
            __hip_atomic_store(mg.flags + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t y = 1 + g * kV2Threads + threadIdx.x; y < nym1; y += G * kV2Threads) {  // :1363-1366
            const size_t r = (size_t)y * nx;
            const float wl = mbLeft[y] ? 1.f : 0.f, wr = mbRight[y] ? 1.f : 0.f;
            f[r] += (f[r + 1] - f[r]) * wl;
            f[r + nxm1] += (f[r + nx - 2] - f[r + nxm1]) * wr;
        }
        slice_barrier(sync, ++barriers * G, a.error);
        for (uint32_t x = g * kV2Threads + threadIdx.x; x < nx; x += G * kV2Threads) {  // :1367-1370
            const size_t bo = (size_t)nym1 * nx + x;
            const float wt = mbTop[x] ? 1.f : 0.f, wb = mbBot[x] ? 1.f : 0.f;
            f[x] += (f[nx + x] - f[x]) * wt;
            f[bo] += (f[bo - nx] - f[bo]) * wb;
        }
        slice_barrier(sync, ++barriers * G, a.error);
        if (launch_failed(a.error)) return;
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
    int sumAlgo;
    const double* defaults;  // per slice, see FillStatsArgs
    const unsigned long long* bounds;
};

__global__ void __launch_bounds__(kFillBlock) creepfill_kernel(CreepArgs a)
{
    __shared__ __align__(16) double lds[2 * kSumTile];
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
    const double sum = scan_order_sum(f, total, a.useDefault ? 2 : 0, 0., lds, &nUndef, a.sumAlgo);  // a default value needs no average
    if (threadIdx.x == 0) {
        shUndef = nUndef;
        const unsigned long long nDef = total - nUndef;
        shDefault = a.defaults ? (float)a.defaults[blockIdx.x] : (a.useDefault ? a.defaultVal : ((nDef != 0) ? (float)(sum / (double)nDef) : 0.f));  // :1516
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
    const unsigned long long bound = a.bounds ? a.bounds[blockIdx.x] : nDef;
    while (changedInLoop > 0 && l < bound) {  // :1430
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

// ------------------------------------------------------------------ creep fill, systolic version
// Same row-band pipeline as fill2d_kernel_v2 (lane = row, skewed columns, LDS ring, LDS hand-off between bands).
// The reference's per-cell state (:1389-1421) is folded into bit masks in the same skewed layout as the rings:
//   D  cell was defined on entry                      w = setWeight, r = repeat, never updated
//   U  cell has been updated at least once            w = 1
// A cell that starts being updated in sweep s is updated in every sweep s .. s + repeat - 1 (its neighbours never
// lose their weight), so "r[p] < repeat" in sweep l is "p is not in U as of sweep l - repeat": U is kept for the
// last repeat + 1 sweeps instead of a counter per cell.  Weights travel as floats (0, 1, setWeight: all exact).
constexpr int kCreepWaves = 8;    // 8 waves x 256 registers: the creep step keeps more state than the 128 registers of a 16-wave workgroup hold
constexpr int kCreepThreads = kCreepWaves * kWave;
constexpr int kCreepCh = 32;                        // 128-byte row pieces per chunk: whole lines, one memory event per 32 columns
constexpr int kCreepRingW = 2 * kCreepCh;
constexpr int kCreepPitch = kCreepRingW + 1;
constexpr int kCreepRowsPerIt = kWave / kCreepCh;
constexpr int kCreepChunksPerWord = 32 / kCreepCh;
constexpr int kHandWC = 128;  // hand-off window of the creep kernel: values and weight codes share the LDS left

struct CreepV2Args {
    float* field;
    uint32_t* maskD;   // [nz][ny][mws]           interior rows skewed by (y - 1) & 63, rows 0 and ny - 1 unskewed
    uint32_t* maskU;   // [nz][gens][ny][mws]     zero on entry
    SliceStats* stats;
    uint32_t nx, ny, mws, gens;
    int useDefault;
    float defaultVal;
    uint32_t repeat;
    int setWeight;     // >= 0
    int sumAlgo;
    int skipIdle;
    unsigned int* error;  // one word per launch: set by a wait that gave up (see MultiWg)
    // several workgroups per slice (creepfill_kernel_v3): per slice [0] barrier counter, [1..3] "something changed" by sweep
    // mod 3, [4 .. 4 + bands) progress words of the hand-offs that cross workgroups
    unsigned int* sync;
    uint32_t syncStride, groups, nz;
};

struct HandoffC {
    float* data;             // [16][2][kHandWC]
    unsigned char* wcode;    // [16][2][kHandWC]  0, 1, 2 = setWeight
    unsigned int* produced;  // [16][2]
    unsigned int* consumed;  // [16][2]
};

// value of lane l+1 (lane 63 keeps its own): "wave_shl:1" (0x130)
__device__ __forceinline__ uint32_t lane_from_below(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false);
}

template <bool MULTI>
__device__ void creep_band(float* __restrict__ f, const uint32_t* __restrict__ maskD, const uint32_t* __restrict__ uOld,
                           const uint32_t* __restrict__ uHist, uint32_t* __restrict__ uNew, float* ring, HandoffC hand, uint32_t b,
                           uint32_t nx, uint32_t ny, uint32_t mws, float swf, bool skipIdle, int& changed, MultiWg mg)
{
    unsigned int* const error = mg.error;
    using rsrc_t = __amdgpu_buffer_rsrc_t;
    b = __builtin_amdgcn_readfirstlane(b);  // wave-uniform, see fill2d_band
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t y0 = 1 + kWave * b;
    const uint32_t nrow = min((uint32_t)kWave, (ny - 1) - y0);
    const uint32_t L = nrow - 1;
    const bool rowValid = lane < nrow;
    const uint32_t y = y0 + min(lane, L);
    const uint32_t C = nx - 2;
    const uint32_t xpEnd = C + L;
    float* ringRow = ring + lane * kCreepPitch;
    const float* ringBelow = ring + min(lane + 1, (uint32_t)kWave - 1) * kCreepPitch;
    const float left0 = f[(size_t)y * nx];
    const uint32_t* drow = maskD + (size_t)y * mws;
    const uint32_t* urow = uOld + (size_t)y * mws;
    const uint32_t* hrow = uHist ? uHist + (size_t)y * mws : nullptr;
    uint32_t* nrowU = uNew + (size_t)y * mws;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(f + (size_t)(y0 - 1) * nx, 0, (nrow + 2) * nx * 4u, 0x00020000);
    const bool hasBelow = y0 + nrow < ny - 1;
    const bool outGlobal = hasBelow && (b % kCreepWaves) == kCreepWaves - 1;
    const bool inGlobal = b > 0 && (b % kCreepWaves) == 0;
    const bool writeThrough = MULTI && outGlobal;  // several workgroups per slice: the band below runs on another CU (fill2d_band)
    const uint32_t kOob = 0xFFFFFFFFu;
    // weight of border column 0 of my row: skewed column = lane
    const float wLeft0 = ((drow[lane >> 5] >> (lane & 31)) & 1u) ? swf : 0.f;

    const uint32_t crow = lane / kCreepCh, ccol = lane % kCreepCh;
    float stage[kCreepCh];
    auto chunk_off = [&](uint32_t c, uint32_t it, bool store) -> uint32_t {
        const uint32_t row = kCreepRowsPerIt * it + crow;
        const int64_t x = (int64_t)c * kCreepCh + ccol - row;
        const bool ok = row < nrow && (store ? (x >= 1 && x <= (int64_t)C) : (x >= 0 && x <= (int64_t)nx - 1));
        return ok ? (uint32_t)(((row + 1) * nx + x) * 4u) : kOob;
    };
    // interior chunks: lean addressing, see fill2d_band
    auto interior = [&](uint32_t c) -> bool { return nrow == (uint32_t)kWave && c * kCreepCh >= (uint32_t)kWave && c * kCreepCh + kCreepCh - 1 <= C; };
    const uint32_t voffLane = ((crow + 1) * nx + ccol - crow) * 4u;           // row crow, chunk 0, column ccol - crow
    const uint32_t rowStep = (uint32_t)kCreepRowsPerIt * (nx - 1) * 4u;            // next row group: kCreepRowsPerIt rows down, as many columns back
    float* ringLane = ring + crow * kCreepPitch + ccol;
    auto load_chunk = [&](uint32_t c) {
        if (interior(c)) {
            const uint32_t s0 = c * kCreepCh * 4u;
#pragma unroll
            for (uint32_t it = 0; it < (uint32_t)kCreepCh; ++it)
                stage[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voffLane, s0 + it * rowStep, 0));
            return;
        }
#pragma unroll
        for (uint32_t it = 0; it < (uint32_t)kCreepCh; ++it)
            stage[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, chunk_off(c, it, false), 0, 0));
    };
    auto commit_chunk = [&](uint32_t c) {
        float* dst = ringLane + ((c * kCreepCh) & kCreepCh);
#pragma unroll
        for (uint32_t it = 0; it < (uint32_t)kCreepCh; ++it) dst[kCreepRowsPerIt * it * kCreepPitch] = stage[it];
    };
    // ring -> registers before the slot is refilled, registers -> global behind the event's loads (see fill2d_band)
    float v[kCreepCh];
    auto flush_read = [&](uint32_t c) {
        const float* src = ringLane + ((c * kCreepCh) & kCreepCh);
#pragma unroll
        for (uint32_t it = 0; it < (uint32_t)kCreepCh; ++it) v[it] = src[kCreepRowsPerIt * it * kCreepPitch];
    };
    auto flush_store = [&](uint32_t c) {
        if (interior(c)) {
            const uint32_t s0 = c * kCreepCh * 4u;
#pragma unroll
            for (uint32_t it = 0; it < (uint32_t)kCreepCh; ++it) {
                if (writeThrough) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[it]), rs, voffLane, s0 + it * rowStep, 17);
                else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[it]), rs, voffLane, s0 + it * rowStep, 0);
            }
            return;
        }
#pragma unroll
        for (uint32_t it = 0; it < (uint32_t)kCreepCh; ++it) {
            if (writeThrough) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[it]), rs, chunk_off(c, it, true), 0, 17);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[it]), rs, chunk_off(c, it, true), 0, 0);
        }
    };
    // this sweep's U word of my row: the band below reads the last row's words (weights of its "up" cells)
    auto store_u = [&](uint32_t word, uint32_t value) {
        if (writeThrough) __hip_atomic_store(&nrowU[word], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else nrowU[word] = value;
    };
    auto load_block = [&](uint32_t rowInBuf, uint32_t k) {
        const uint32_t col = 64 * k + lane;
        const uint32_t off = col <= nx - 1 ? (rowInBuf * nx + col) * 4u : kOob;
        if (MULTI && inGlobal && rowInBuf == 0) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 17));  // see fill2d_band
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
    };
    // weights of 64 columns of an unskewed row (row 0, the first row of the band below, row ny - 1)
    auto load_wblock = [&](uint32_t yRow, uint32_t k) -> float {
        const uint32_t col = min(64 * k + lane, nx - 1);
        const uint32_t d = maskD[(size_t)yRow * mws + (col >> 5)], u = uOld[(size_t)yRow * mws + (col >> 5)];
        return ((d >> (col & 31)) & 1u) ? swf : (float)((u >> (col & 31)) & 1u);
    };

    const uint32_t round = kCreepWaves * (MULTI ? mg.G : 1u);
    const uint32_t slotOut = (b % kCreepWaves) * 2 + ((b / round) & 1);
    const uint32_t slotIn = ((b - 1) % kCreepWaves) * 2 + (((b - 1) / round) & 1);
    float* handOut = hand.data + slotOut * kHandWC;
    unsigned char* handOutW = hand.wcode + slotOut * kHandWC;
    const float* handIn = hand.data + slotIn * kHandWC;
    const unsigned char* handInW = hand.wcode + slotIn * kHandWC;
    if (lane == 0) {
        __hip_atomic_store(&hand.produced[slotOut], hand_tag(b, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&hand.consumed[slotOut], hand_tag(b, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // every 16th boundary goes through global memory (see fill2d_band): values from the flushed row, weights from the
    // D mask and this sweep's U words of that row (skew 63), which the producing band stores at every event
    auto wait_above = [&](uint32_t cols) {  // chunk by chunk, see fill2d_band
        cols = min(cols, C + 1);
        if (MULTI) { wait_global_at_least(&mg.flags[b - 1], cols + 1, error); return; }
        wait_lds_at_least(&hand.produced[slotIn], hand_tag(b - 1, cols), error);
    };
    auto load_wblock_above = [&](uint32_t k) -> float {
        const uint32_t xs = min(64 * k + lane, nx - 1) + (kWave - 1);
        const uint32_t d = maskD[(size_t)(y0 - 1) * mws + (xs >> 5)];
        const uint32_t u = MULTI ? __hip_atomic_load(&uNew[(size_t)(y0 - 1) * mws + (xs >> 5)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : uNew[(size_t)(y0 - 1) * mws + (xs >> 5)];
        return ((d >> (xs & 31)) & 1u) ? swf : (float)((u >> (xs & 31)) & 1u);
    };
    // columns [xpc, xpc + kCreepCh) of the row above (values and weights) into lanes (column % 64)
    auto take_above = [&](uint32_t xpc, float& fv, float& wv) {
        wait_lds_at_least(&hand.produced[slotIn], hand_tag(b - 1, min(xpc + kCreepCh, C + 1)), error);
        const uint32_t col = (xpc & ~63u) + lane;
        const float v = handIn[col % kHandWC];
        const unsigned int code = handInW[col % kHandWC];
        if (lane - (xpc & 63u) < (uint32_t)kCreepCh) { fv = v; wv = (code == 2u) ? swf : (float)code; }
        if (lane == 0)
            lds_publish(&hand.consumed[slotIn], hand_tag(b - 1, xpc + kCreepCh));
    };

    load_chunk(0);
    commit_chunk(0);
    load_chunk(1);
    commit_chunk(1);
    load_chunk(2);
    float upCur = 0.f, upWCur = 0.f, upLd = 0.f, upWLd = 0.f;
    const bool fromGlobal = b == 0 || inGlobal;
    if (fromGlobal) {  // see fill2d_band: the chunk in work in lanes (column % 64), the next chunk's block one event ahead
        if (inGlobal) wait_above(kCreepCh);
        upCur = load_block(0, 0);
        upWCur = inGlobal ? load_wblock_above(0) : load_wblock(0, 0);
        if (inGlobal) wait_above(2 * kCreepCh);
        upLd = load_block(0, kCreepCh >> 6);
        upWLd = inGlobal ? load_wblock_above(kCreepCh >> 6) : load_wblock(0, kCreepCh >> 6);
    } else take_above(0, upCur, upWCur);
    const uint32_t yBelow = y0 + nrow;
    float downA = load_block(nrow + 1, 0), downB = downA, downLd = 0.f;
    float downWA = load_wblock(yBelow, 0), downWB = downWA, downWLd = 0.f;
    uint32_t downIssued = 0;
    bool downLdValid = false;
    const uint32_t wLast = mws - 1;
    uint32_t dw = drow[0], dwN = drow[1], dwLd = drow[min(2u, wLast)];
    uint32_t uw = urow[0], uwN = urow[1], uwLd = urow[min(2u, wLast)];
    uint32_t hw = hrow ? hrow[0] : 0u, hwN = hrow ? hrow[1] : 0u, hwLd = hrow ? hrow[min(2u, wLast)] : 0u;
    uint32_t ddw = lane_from_below(dw), ddwN = lane_from_below(dwN), duw = lane_from_below(uw), duwN = lane_from_below(uwN);
    uint32_t un = 0;
    float prevRes = 0.f, prevW = 0.f;
    float prevRight = ringRow[1];

    const uint32_t nChunks = xpEnd / kCreepCh + 1;
    for (uint32_t c = 0; c < nChunks; ++c) {
        const uint32_t xpc = c * kCreepCh;
        if (c > 0) {
            if (outGlobal) {
                if (xpc > L) {  // stores of the previous event have landed: columns < 16 (c - 1) - L of the last row, values and U bits
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0 && xpc - kCreepCh > L) {
                        if (MULTI) __hip_atomic_store(&mg.flags[b], xpc - kCreepCh - L + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else lds_publish(&hand.produced[slotOut], hand_tag(b, xpc - kCreepCh - L));
                    }
                }
                if (rowValid) store_u((c - 1) / kCreepChunksPerWord, un);  // the (partial) word of the chunk just finished
            }
            // small loads first, the chunk prefetch last (see fill2d_band)
            if ((c % kCreepChunksPerWord) == 0) {  // x' is a multiple of 32: the finished U word goes out, every lane switches words
                if (rowValid) store_u(c / kCreepChunksPerWord - 1, un);
                un = 0;
                const uint32_t nxt = min(c / kCreepChunksPerWord + 2, wLast);
                dw = dwN; dwN = dwLd; dwLd = drow[nxt];
                uw = uwN; uwN = uwLd; uwLd = urow[nxt];
                hw = hwN; hwN = hwLd; hwLd = hrow ? hrow[nxt] : 0u;
                ddw = ddwN; ddwN = lane_from_below(dwN);
                duw = duwN; duwN = lane_from_below(uwN);
            }
            if (downLdValid) { downB = downLd; downWB = downWLd; downLdValid = false; }
            if (xpc + 2 * kCreepCh > L && ((xpc + 2 * kCreepCh - L) >> 6) > downIssued) {  // block j + 1 is requested two chunks before the last lane
                // reaches it and lands in downB at the next event: after block j has moved on to downA, never skipping one
                ++downIssued;
                downLd = load_block(nrow + 1, downIssued);
                downWLd = load_wblock(yBelow, downIssued);
                downLdValid = true;
            }
            if (fromGlobal) {
                if (lane - (xpc & 63u) < (uint32_t)kCreepCh) { upCur = upLd; upWCur = upWLd; }  // this chunk's columns, requested one event ago
                const uint32_t k = (xpc + kCreepCh) >> 6;
                if (inGlobal) wait_above(xpc + 2 * kCreepCh);
                upLd = load_block(0, k);
                upWLd = inGlobal ? load_wblock_above(k) : load_wblock(0, k);
            }
            flush_read(c - 1);
            commit_chunk(c + 1);
            load_chunk(c + 2);  // as soon as its registers are free, before the waits on the neighbouring bands (see fill2d_band)
            if (xpc > L && !outGlobal) {
                if (lane == 0)
                    lds_publish(&hand.produced[slotOut], hand_tag(b, xpc - L));
                if (hasBelow) {
                    const unsigned int limit = xpc + kCreepCh - L;
                    unsigned long long tSpin = 0;
                    for (unsigned int it = 0;; ++it) {
                        const unsigned int cns = lds_observe(&hand.consumed[slotOut]);
                        if (limit <= (cns & 0x7FFFFu) + kHandWC) break;
                        __builtin_amdgcn_s_sleep(1);
                        if ((it & 0xFFF) == 0xFFF && (spin_expired(tSpin) || launch_failed(error))) { fail_launch(error, 3); break; }
                    }
                }
            }
            if (!fromGlobal) take_above(xpc, upCur, upWCur);
            flush_store(c - 1);
        }
        const uint32_t xp0 = max(xpc, 1u), xp1 = min(xpc + kCreepCh - 1, xpEnd);
        // A chunk in which no row has a cell that may still change (undefined on entry and not yet updated `repeat` times:
        // neither D nor H) is passed over: nothing is computed, the state the next chunk and the band below need is taken
        // from the ring and the masks.  After the first sweeps that is most of the field.
        const uint32_t chunkBits = ((xp1 - xp0 + 1 >= 32) ? 0xFFFFFFFFu : ((1u << (xp1 - xp0 + 1)) - 1u)) << (xp0 & 31);
        if (skipIdle && !__any(rowValid && ((~dw & ~hw & chunkBits) != 0u))) {
            for (uint32_t xp = xp0; xp <= xp1; ++xp)
                if (xp > L && ((xp - L) & 63) == 0) { downA = downB; downWA = downWB; }
            un |= uw & chunkBits;  // U is carried over unchanged
            {   // the band below still needs this stretch of the last row: lanes 0..15 copy one column each
                const uint32_t dL = (uint32_t)__builtin_amdgcn_readlane((int)dw, (int)L), uL = (uint32_t)__builtin_amdgcn_readlane((int)uw, (int)L);
                const uint32_t xpk = xp0 + lane;
                const int64_t xk = (int64_t)xpk - L;
                if (xpk <= xp1 && xk >= 1 && xk <= (int64_t)C) {
                    handOut[(uint32_t)xk % kHandWC] = ring[L * kCreepPitch + (xpk & (kCreepRingW - 1))];
                    handOutW[(uint32_t)xk % kHandWC] = ((dL >> (xpk & 31)) & 1u) ? 2 : ((uL >> (xpk & 31)) & 1u);
                }
            }
            prevRes = ringRow[xp1 & (kCreepRingW - 1)];
            prevW = ((dw >> (xp1 & 31)) & 1u) ? swf : (float)((uw >> (xp1 & 31)) & 1u);
            prevRight = ringRow[(xp1 + 1) & (kCreepRingW - 1)];
            continue;
        }
        if (interior(c) && xpc > (uint32_t)kWave) {
            // ---- every lane is at an interior cell with x >= 2: unrolled, no range tests, mask bits as (kCreepCh + 1)-bit windows
            const uint32_t half = xpc & kCreepCh;
            float* rc = ringRow + half;
            const float* rb = ringBelow + half;
            const uint32_t rNext = (half ^ kCreepCh);
            const uint32_t sh0 = xpc & 31, up0 = xpc & 63;
            const uint32_t kSwitch = (L - xpc) & 63;
            const bool switches = kSwitch < (uint32_t)kCreepCh;
            const int dBase = (int)((xpc - L) & 63);
            // bit k: the cell of step k, bit k + 1: its right neighbour (own row) / the cell below (row of lane + 1)
            // (kCreepCh + 1)-bit windows of the word pairs, as 64-bit values: the right neighbour of the chunk's last column is bit kCreepCh
            const uint64_t d17 = (((uint64_t)dwN << 32) | dw) >> sh0, u17 = (((uint64_t)uwN << 32) | uw) >> sh0;
            const uint64_t dd17 = (((uint64_t)ddwN << 32) | ddw) >> sh0, du17 = (((uint64_t)duwN << 32) | duw) >> sh0;
            const uint32_t h16 = hw >> sh0;
            uint32_t newBits = 0;
            // the windows as two 32-bit halves: bit tests at compile-time positions stay 32-bit operations
            const uint32_t dLo = (uint32_t)d17, dHi = (uint32_t)(d17 >> 32), uLo = (uint32_t)u17, uHi = (uint32_t)(u17 >> 32);
            const uint32_t ddLo = (uint32_t)dd17, ddHi = (uint32_t)(dd17 >> 32), duLo = (uint32_t)du17, duHi = (uint32_t)(du17 >> 32);
            auto bit = [](uint32_t lo, uint32_t hi, int i) -> bool { return i < 32 ? ((lo >> i) & 1u) != 0 : ((hi >> (i - 32)) & 1u) != 0; };
#pragma unroll
            for (int k = 0; k < kCreepCh; ++k) {
                const bool cD = bit(dLo, dHi, k), cU = bit(uLo, uHi, k), cH = (h16 >> k) & 1u;
                const float wr = bit(dLo, dHi, k + 1) ? swf : (bit(uLo, uHi, k + 1) ? 1.f : 0.f);
                float wd = bit(ddLo, ddHi, k + 1) ? swf : (bit(duLo, duHi, k + 1) ? 1.f : 0.f);
                const float right = (k < kCreepCh - 1) ? rc[k + 1] : ringRow[rNext];
                float down = (k < kCreepCh - 1) ? rb[k + 1] : ringBelow[rNext];
                const float center = prevRight;
                const float up = lane_from_above_or(prevRes, lane_value(upCur, (int)(up0 + k)));
                const float wu = lane_from_above_or(prevW, lane_value(upWCur, (int)(up0 + k)));
                const bool after = switches && (uint32_t)k >= kSwitch;
                const float dsel = after ? downB : downA, dwsel = after ? downWB : downWA;
                const float downLast = lane_value(dsel, (dBase + k) & 63), wdLast = lane_value(dwsel, (dBase + k) & 63);
                if (lane == L) { down = downLast; wd = wdLast; }
                const float wsum = ((wr + prevW) + wd) + wu;                                        // :1445
                const bool act = !cD && !cH && wsum != 0.f;                                         // :1443, :1446
                float v = center + (((wr * right + prevW * prevRes) + wd * down) + wu * up);        // :1451
                v = v / (1.f + wsum);                                                               // :1452
                const float res = act ? v : center;
                const bool newU = cU || act;
                rc[k] = res;
                if (act) changed = 1;
                newBits |= (newU ? 1u : 0u) << k;
                prevRes = res;
                prevW = cD ? swf : (newU ? 1.f : 0.f);
                prevRight = right;
            }
            if (switches) { downA = downB; downWA = downWB; }
            un |= newBits << sh0;
            {
                const uint32_t dL = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d17, (int)L), nL = (uint32_t)__builtin_amdgcn_readlane((int)newBits, (int)L);
                if (lane < (uint32_t)kCreepCh) {
                    const uint32_t xk = xpc + lane - L;
                    handOut[xk % kHandWC] = ring[L * kCreepPitch + ((xpc + lane) & (kCreepRingW - 1))];
                    handOutW[xk % kHandWC] = ((dL >> lane) & 1u) ? 2 : ((nL >> lane) & 1u);
                }
            }
            continue;
        }
        for (uint32_t xp = xp0; xp <= xp1; ++xp) {
            if (xp > L && ((xp - L) & 63) == 0) { downA = downB; downWA = downWB; }
            const int64_t x = (int64_t)xp - lane;
            const bool inRange = rowValid && x >= 1 && x <= (int64_t)C;
            const uint32_t sh = xp & 31;
            // bit 0: this cell, bit 1: the cell to the right (own row) / the cell below (row of lane + 1)
            const uint32_t dPair = __builtin_amdgcn_alignbit(dwN, dw, sh), uPair = __builtin_amdgcn_alignbit(uwN, uw, sh);
            const uint32_t ddPair = __builtin_amdgcn_alignbit(ddwN, ddw, sh), duPair = __builtin_amdgcn_alignbit(duwN, duw, sh);
            const bool cD = dPair & 1u, cU = uPair & 1u, cH = (hw >> sh) & 1u;
            const float wr = (dPair & 2u) ? swf : ((uPair & 2u) ? 1.f : 0.f);
            float wd = (ddPair & 2u) ? swf : ((duPair & 2u) ? 1.f : 0.f);
            const uint32_t rp = (xp + 1) & (kCreepRingW - 1);
            const float right = ringRow[rp];
            float down = ringBelow[rp];
            const float center = prevRight;
            const float up = lane_from_above_or(prevRes, lane_value(upCur, (int)(xp & 63)));
            const float wu = lane_from_above_or(prevW, lane_value(upWCur, (int)(xp & 63)));
            const int dIdx = (int)((xp >= L) ? ((xp - L) & 63) : 0);
            const float downLast = lane_value(downA, dIdx), wdLast = lane_value(downWA, dIdx);
            if (lane == L) { down = downLast; wd = wdLast; }
            const float left = (x == 1) ? left0 : prevRes;
            const float wl = (x == 1) ? wLeft0 : prevW;
            const float wsum = ((wr + wl) + wd) + wu;                          // :1445, small integers: exact
            const bool act = inRange && !cD && !cH && wsum != 0.f;             // :1443, :1446
            float v = center + (((wr * right + wl * left) + wd * down) + wu * up);  // :1451
            v = v / (1.f + wsum);                                              // :1452
            const float res = act ? v : center;
            const bool newU = cU || act;
            if (inRange) {
                if (act) { ringRow[xp & (kCreepRingW - 1)] = res; changed = 1; }
                if (lane == L) {
                    handOut[(uint32_t)x % kHandWC] = res;
                    handOutW[(uint32_t)x % kHandWC] = cD ? 2 : (newU ? 1 : 0);
                }
            }
            un |= (newU ? 1u : 0u) << sh;
            prevRes = res;
            prevW = cD ? swf : (newU ? 1.f : 0.f);
            prevRight = right;
        }
    }
    flush_read(nChunks - 1);
    flush_store(nChunks - 1);
    if (rowValid) store_u((nChunks - 1) / kCreepChunksPerWord, un);
    if (outGlobal) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        if (MULTI && outGlobal) __hip_atomic_store(&mg.flags[b], C + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_publish(&hand.produced[slotOut], hand_tag(b, C + 1));
    }
}

__global__ void __launch_bounds__(kCreepThreads) creepfill_kernel_v2(CreepV2Args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rings = smem;
    HandoffC hand;
    hand.data = smem + kCreepWaves * kWave * kCreepPitch;
    hand.wcode = reinterpret_cast<unsigned char*>(hand.data + kCreepWaves * 2 * kHandWC);
    hand.produced = reinterpret_cast<unsigned int*>(hand.wcode + kCreepWaves * 2 * kHandWC);
    hand.consumed = hand.produced + kCreepWaves * 2;
    const uint32_t nx = a.nx, ny = a.ny, mws = a.mws;
    const size_t total = (size_t)nx * ny;
    const size_t maskWords = (size_t)ny * mws;
    float* f = a.field + (size_t)blockIdx.x * total;
    uint32_t* maskD = a.maskD + (size_t)blockIdx.x * maskWords;
    uint32_t* maskU = a.maskU + (size_t)blockIdx.x * a.gens * maskWords;
    SliceStats* st = a.stats + blockIdx.x;
    const uint32_t wave = threadIdx.x / kWave;

    // sum, first guess and the D mask were made by fill_stats_kernel / first_guess_kernel
    if (st->skip) return;  // :1384-1386, :1515
    const unsigned long long nDef = st->sweepBound;  // the loop's bound (:1430)
    const uint32_t repeat = a.repeat;
    const float swf = (float)a.setWeight;

    const uint32_t nxm1 = nx - 1, nym1 = ny - 1;
    const uint32_t nBands = (ny - 2 + kWave - 1) / kWave;
    float* ring = rings + wave * kWave * kCreepPitch;
    unsigned long long l = 0;
    int changedInLoop = 1;
    while (repeat > 0 && changedInLoop && l < nDef) {  // :1430 (nothing has r < repeat when repeat is 0)
        l++;
        if (threadIdx.x < kCreepWaves * 2) { hand.produced[threadIdx.x] = 0; hand.consumed[threadIdx.x] = 0; }
        __syncthreads();
        const uint32_t* uOld = maskU + (size_t)((l - 1) % a.gens) * maskWords;
        const uint32_t* uHist = (l > repeat) ? maskU + (size_t)((l - repeat) % a.gens) * maskWords : nullptr;
        uint32_t* uNew = maskU + (size_t)(l % a.gens) * maskWords;
        int mine = 0;
        for (uint32_t b = wave; b < nBands; b += kCreepWaves)
            creep_band<false>(f, maskD, uOld, uHist, uNew, ring, hand, b, nx, ny, mws, swf, a.skipIdle != 0, mine, MultiWg{0u, 1u, 0u, nullptr, a.error});
        changedInLoop = __syncthreads_or(mine);
    }
    // borders (:1464-1489): undefined border cells have r = 0 < repeat in every round, defined ones never change
    const uint32_t* uFin = maskU + (size_t)(l % a.gens) * maskWords;
    auto defined = [&](uint32_t y, uint32_t x) -> bool {
        const uint32_t sk = (y == 0 || y == nym1) ? 0u : ((y - 1) & (kWave - 1));
        return (maskD[(size_t)y * mws + ((x + sk) >> 5)] >> ((x + sk) & 31)) & 1u;
    };
    auto w_interior = [&](uint32_t y, uint32_t x) -> int {  // final weight of an interior cell
        const uint32_t sk = (y - 1) & (kWave - 1);
        if (defined(y, x)) return a.setWeight;
        return (uFin[(size_t)y * mws + ((x + sk) >> 5)] >> ((x + sk) & 31)) & 1u;
    };
    for (uint32_t k = 0; k < repeat; ++k) {
        for (uint32_t y = 1 + threadIdx.x; y < nym1; y += kCreepThreads) {
            const size_t row = (size_t)y * nx;
            if (!defined(y, 0)) {
                const int wn = w_interior(y, 1);
                f[row] += f[row + 1] * wn;
                f[row] /= (float)(1 + wn);
            }
            if (!defined(y, nxm1)) {
                const int wn = w_interior(y, nx - 2);
                f[row + nxm1] += f[row + nx - 2] * wn;
                f[row + nxm1] /= (float)(1 + wn);
            }
        }
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < nx; x += kCreepThreads) {
            const size_t bo = (size_t)nym1 * nx + x;
            const bool edge = (x == 0 || x == nxm1);  // the neighbour is a border cell of the column loop above: w = 1 if it was undefined
            if (!defined(0, x)) {
                const int wn = edge ? (defined(1, x) ? a.setWeight : 1) : w_interior(1, x);
                f[x] += f[nx + x] * wn;
                f[x] /= (float)(1 + wn);
            }
            if (!defined(nym1, x)) {
                const int wn = edge ? (defined(nym1 - 1, x) ? a.setWeight : 1) : w_interior(nym1 - 1, x);
                f[bo] += f[bo - nx] * wn;
                f[bo] /= (float)(1 + wn);
            }
        }
        __syncthreads();
    }
}

// The sweeps of creepfill_kernel_v2 with the bands of a slice dealt to a.groups workgroups (MultiWg, fill2d_kernel_v3); the
// border rounds that follow the sweeps are little work and stay with the slice's first workgroup.
__global__ void __launch_bounds__(kCreepThreads) creepfill_kernel_v3(CreepV2Args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rings = smem;
    HandoffC hand;
    hand.data = smem + kCreepWaves * kWave * kCreepPitch;
    hand.wcode = reinterpret_cast<unsigned char*>(hand.data + kCreepWaves * 2 * kHandWC);
    hand.produced = reinterpret_cast<unsigned int*>(hand.wcode + kCreepWaves * 2 * kHandWC);
    hand.consumed = hand.produced + kCreepWaves * 2;
    const uint32_t G = a.groups;
    const uint32_t slice = (blockIdx.x % kXcds) + kXcds * (blockIdx.x / (kXcds * G));
    const uint32_t g = (blockIdx.x / kXcds) % G;
    if (slice >= a.nz) return;
    const uint32_t nx = a.nx, ny = a.ny, mws = a.mws;
    const size_t total = (size_t)nx * ny;
    const size_t maskWords = (size_t)ny * mws;
    float* f = a.field + (size_t)slice * total;
    uint32_t* maskD = a.maskD + (size_t)slice * maskWords;
    uint32_t* maskU = a.maskU + (size_t)slice * a.gens * maskWords;
    SliceStats* st = a.stats + slice;
    unsigned int* sync = a.sync + (size_t)slice * a.syncStride;
    const MultiWg mg{g, G, 0u, sync + 4, a.error};
    const uint32_t wave = threadIdx.x / kWave;
    if (st->skip) return;  // :1384-1386, :1515
    const unsigned long long nDef = st->sweepBound;  // the loop's bound (:1430)
    const uint32_t repeat = a.repeat;
    const float swf = (float)a.setWeight;
    const uint32_t nxm1 = nx - 1, nym1 = ny - 1;
    const uint32_t nBands = (ny - 2 + kWave - 1) / kWave;
    float* ring = rings + wave * kWave * kCreepPitch;
    unsigned long long l = 0;
    unsigned int barriers = 0;
    int changedInLoop = 1;
    while (repeat > 0 && changedInLoop && l < nDef) {  // :1430
        l++;
        if (threadIdx.x < kCreepWaves * 2) { hand.produced[threadIdx.x] = 0; hand.consumed[threadIdx.x] = 0; }
        __syncthreads();
        const uint32_t* uOld = maskU + (size_t)((l - 1) % a.gens) * maskWords;
        const uint32_t* uHist = (l > repeat) ? maskU + (size_t)((l - repeat) % a.gens) * maskWords : nullptr;
        uint32_t* uNew = maskU + (size_t)(l % a.gens) * maskWords;
        int mine = 0;
        for (uint32_t b = g * kCreepWaves + wave; b < nBands; b += G * kCreepWaves)
            creep_band<true>(f, maskD, uOld, uHist, uNew, ring, hand, b, nx, ny, mws, swf, a.skipIdle != 0, mine, mg);
        unsigned int* changedWord = sync + 1 + (unsigned int)(l % 3);
        if (__syncthreads_or(mine) && threadIdx.x == 0) __hip_atomic_fetch_or(changedWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        slice_barrier(sync, ++barriers * G, a.error);
        if (launch_failed(a.error)) return;
        changedInLoop = __hip_atomic_load(changedWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        // the word of the sweep after next (nobody adds to it before the next barrier, which this workgroup has yet to reach),
        // and this workgroup's progress words for the next sweep
        if (g == 0 && threadIdx.x == 0) __hip_atomic_store(sync + 1 + (unsigned int)((l + 2) % 3), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t b = g * kCreepWaves + kCreepWaves - 1 + threadIdx.x * G * kCreepWaves; b < nBands; b += kCreepThreads * G * kCreepWaves)
            __hip_atomic_store(mg.flags + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (g != 0) return;
    // borders (:1464-1489), as in creepfill_kernel_v2: everything the other workgroups wrote is visible behind the last barrier
    const uint32_t* uFin = maskU + (size_t)(l % a.gens) * maskWords;
    auto defined = [&](uint32_t y, uint32_t x) -> bool {
        const uint32_t sk = (y == 0 || y == nym1) ? 0u : ((y - 1) & (kWave - 1));
        return (maskD[(size_t)y * mws + ((x + sk) >> 5)] >> ((x + sk) & 31)) & 1u;
    };
    auto w_interior = [&](uint32_t y, uint32_t x) -> int {
        const uint32_t sk = (y - 1) & (kWave - 1);
        if (defined(y, x)) return a.setWeight;
        return (uFin[(size_t)y * mws + ((x + sk) >> 5)] >> ((x + sk) & 31)) & 1u;
    };
    for (uint32_t k = 0; k < repeat; ++k) {
        for (uint32_t y = 1 + threadIdx.x; y < nym1; y += kCreepThreads) {
            const size_t row = (size_t)y * nx;
            if (!defined(y, 0)) {
                const int wn = w_interior(y, 1);
                f[row] += f[row + 1] * wn;
                f[row] /= (float)(1 + wn);
            }
            if (!defined(y, nxm1)) {
                const int wn = w_interior(y, nx - 2);
                f[row + nxm1] += f[row + nx - 2] * wn;
                f[row + nxm1] /= (float)(1 + wn);
            }
        }
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < nx; x += kCreepThreads) {
            const size_t bo = (size_t)nym1 * nx + x;
            const bool edge = (x == 0 || x == nxm1);
            if (!defined(0, x)) {
                const int wn = edge ? (defined(1, x) ? a.setWeight : 1) : w_interior(1, x);
                f[x] += f[nx + x] * wn;
                f[x] /= (float)(1 + wn);
            }
            if (!defined(nym1, x)) {
                const int wn = edge ? (defined(nym1 - 1, x) ? a.setWeight : 1) : w_interior(nym1 - 1, x);
                f[bo] += f[bo - nx] * wn;
                f[bo] /= (float)(1 + wn);
            }
        }
        __syncthreads();
    }
}

// A grid whose workgroups wait for each other: all of them have to be resident at once.  hipLaunchCooperativeKernel checks the grid
// against the occupancy query and then launches like any other launch -- plain, cooperative and graph launches give identical
// residency (MI355X_MICROARCH.md, residency and cooperative launch).  The same check is made here and the launch is a plain one:
// 15-19 us less per call, and a process that is being profiled no longer dies in its exit handlers (rocprofv3 7.2 ends with
// SIGSEGV inside exit() after any cooperative launch, after its output is complete: profiles/r03_fill2d_nz16_abnormal_exit.txt).
// FILL_COOP=1 (tuning build) brings the cooperative launch back.  false: not every workgroup would be resident.
bool launch_resident(const void* kernel, dim3 grid, dim3 block, void** params, size_t ldsBytes, hipStream_t stream)
{
    if (tuning("FILL_COOP", 0) != 0) {
        if (hipLaunchCooperativeKernel(kernel, grid, block, params, (unsigned int)ldsBytes, stream) == hipSuccess) return true;
        (void)hipGetLastError();
        return false;
    }
    int perCu = 0, dev = 0, cus = 0;
    FA_HIP(hipGetDevice(&dev));
    FA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, (int)block.x, ldsBytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if ((size_t)perCu * (size_t)cus < (size_t)grid.x * grid.y * grid.z) return false;
    FA_HIP(hipLaunchKernel(kernel, grid, block, params, ldsBytes, stream));
    return true;
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

namespace {

// The sweeps over whole slices [nz][ny][nx].  d_defaults / d_devs (device, per slice): first guess and convergence criterion given
// instead of computed; couple > 0: slices i, i + couple, ... end their sweeps together (fill2d by rectangles, see run_fill2d) --
// false where that cannot be launched (nothing has been touched then).
bool run_fill2d_whole(size_t nx, size_t ny, size_t nz, float* d_field, float relaxCrit, float corrEff, size_t maxLoop,
                      size_t* h_nChanged, hipStream_t stream, const double* d_defaults, const double* d_devs, uint32_t couple)
{
    if (nx * ny == 0 || nz == 0) return true;  // :1248
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
        a.sumAlgo = tuning("SUM_ALGO", 1);
        if (couple > 0 && tuning("FILL_MULTI", 1) == 0) return false;
        launch_fill_prologue(false, d_field, stats.get(), nx, ny, nz, maskS.get(), mws, mbRows.get(), mbCols.get(), d_defaults == nullptr, d_defaults != nullptr, 0.f,
                             relaxCrit, stream, d_defaults, nullptr, d_devs);
        // small batches and short calls: 16 waves x 16 columns; from FILL_WIDE_NZ slices on: 8 waves x 32 columns
        const int geometry = tuning("FILL_GEOMETRY", 0);  // 0: by batch size, 1: 16 x 16, 2: 8 x 32
        const bool wide = geometry == 2 || (geometry == 0 && nz >= (size_t)tuning("FILL_WIDE_NZ", 8));
        const int waves = wide ? 8 : 16, ch = wide ? 32 : 16;
        const size_t ldsBytes = (size_t)waves * (kWave + 1) * (2 * ch + 1) * sizeof(float) + (size_t)waves * 2 * kHandW * sizeof(float) +
                                (size_t)waves * 4 * sizeof(unsigned int);
        DeviceArray<unsigned int> error(1);
        FA_HIP(hipMemsetAsync(error.get(), 0, sizeof(unsigned int), stream));
        a.error = error.get();
        // Small batches leave most of the chip idle at one workgroup per slice: deal the bands of a slice to several
        // workgroups (one per CU: the rings fill the LDS), as many as there are groups of `waves` bands and as fit the XCD
        // the slice's workgroups share with the slices of the same i % 8.
        // (four waves per workgroup there, one per SIMD: a band's time is the time of its dependent instruction chain, and a
        // wave that shares its SIMD with three others runs that chain at a quarter of the speed; the critical path of a
        // sweep -- every band starts ~130 columns behind the one above -- is what a small batch waits for)
        int mwaves = tuning("FILL_MULTI_WAVES", 4), mch = tuning("FILL_MULTI_CH", 16) == 32 ? 32 : 16;
        if (!(mwaves == 4 || mwaves == 8 || mwaves == 16)) mwaves = 4;
        if (mwaves == 16) mch = 16;  // 16 rings of 32 columns do not fit the LDS
        const size_t bandGroups = ceil_div(nBands, (size_t)mwaves);
        const size_t perXcd = ceil_div(nz, (size_t)kXcds);  // slices whose workgroups meet on one XCD
        int cus = 0, dev = 0;
        FA_HIP(hipGetDevice(&dev));
        FA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const size_t cusPerXcd = std::max(1, cus / kXcds);
        size_t groups = std::min(bandGroups, perXcd ? cusPerXcd / perXcd : (size_t)1);
        if (tuning("FILL_MULTI", 1) == 0 || groups < 2) groups = 1;
        if (couple > 0) groups = std::max<size_t>(groups, 1);
        DeviceArray<unsigned int> sync;
        if (groups > 1 || couple > 0) {
            a.syncStride = (uint32_t)(4 + nBands);
            a.groups = (uint32_t)groups;
            a.nz = (uint32_t)nz;
            a.couple = couple;
            sync.allocate(nz * a.syncStride);
            FA_HIP(hipMemsetAsync(sync.get(), 0, sync.bytes(), stream));
            a.sync = sync.get();
            a.experiment = (uint32_t)tuning("FILL_EXPERIMENT", 0);
            DeviceArray<unsigned long long> prof(8);
            FA_HIP(hipMemsetAsync(prof.get(), 0, 8 * sizeof(unsigned long long), stream));
            a.prof = prof.get();
            struct ProfDump {
                DeviceArray<unsigned long long>& p; hipStream_t st; uint32_t on;
                ~ProfDump() {
                    if (on != 4) return;
                    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    (void)hipStreamSynchronize(st);
                    (void)hipMemcpy(h, p.get(), sizeof(h), hipMemcpyDeviceToHost);
                    std::fprintf(stderr, "fill2d profile: bands %llu, cycles per band: events %.0f (waiting for the band above %.0f, for the band below %.0f), steps %.0f\n",
                                 h[2], h[2] ? (double)h[0] / h[2] : 0.0, h[2] ? (double)h[3] / h[2] : 0.0, h[2] ? (double)h[4] / h[2] : 0.0,
                                 h[2] ? (double)h[1] / h[2] : 0.0);
                    std::fprintf(stderr, "fill2d profile: band 0 (waits for nobody above): events %.0f, steps %.0f cycles per sweep\n", h[7] ? (double)h[5] / h[7] : 0.0,
                                 h[7] ? (double)h[6] / h[7] : 0.0);
                }
            } profDump{prof, stream, a.experiment};
            // more LDS than half a CU has, so that no two of these workgroups share a CU (and its SIMDs)
            const size_t mlds = std::max<size_t>((size_t)mwaves * (kWave + 1) * (2 * mch + 1) * sizeof(float) + (size_t)mwaves * 2 * kHandW * sizeof(float) +
                                                     (size_t)mwaves * 4 * sizeof(unsigned int), 84 * 1024);
            const void* kernel = mwaves == 16  ? reinterpret_cast<const void*>(&fill2d_kernel_v3<16, 16>)
                                 : mwaves == 8 ? (mch == 32 ? reinterpret_cast<const void*>(&fill2d_kernel_v3<32, 8>) : reinterpret_cast<const void*>(&fill2d_kernel_v3<16, 8>))
                                               : (mch == 32 ? reinterpret_cast<const void*>(&fill2d_kernel_v3<32, 4>) : reinterpret_cast<const void*>(&fill2d_kernel_v3<16, 4>));
            allow_dynamic_lds(kernel, mlds);
            void* params[] = {&a};
            const dim3 grid((uint32_t)(kXcds * groups * perXcd));
            // every workgroup of the grid resident (they wait for each other), or one workgroup per slice does the work
            if (!launch_resident(kernel, grid, dim3(mwaves * kWave), params, mlds, stream)) {
                // (the prologue has filled the first guess in: a coupled run is of copies, the caller drops them)
                if (couple > 0) return false;
                groups = 0;
            } else if (groups == 1) groups = 2;  // launched: not again below
        }
        if (groups <= 1) {
            auto launch = [&](auto kernel) {
                allow_dynamic_lds(reinterpret_cast<const void*>(kernel), ldsBytes);
                kernel<<<dim3((uint32_t)nz), waves * kWave, ldsBytes, stream>>>(a);
            };
            if (wide) launch(&fill2d_kernel_v2<32, 8>);
            else launch(&fill2d_kernel_v2<16, 16>);
        }
        FA_HIP(hipGetLastError());
        unsigned int failed = 0;
        FA_HIP(hipMemcpyAsync(&failed, error.get(), sizeof(failed), hipMemcpyDeviceToHost, stream));
        collect_stats(stats, nz, h_nChanged, stream, "fill2d");  // synchronises the stream
        FA_REQUIRE(failed == 0, "fill2d: a hand-off between waves or workgroups did not arrive (wait " + std::to_string(failed) +
                                    " gave up); the field is not valid");
        return true;
    }
    if (couple > 0 || d_defaults) return false;
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
    a.sumAlgo = tuning("SUM_ALGO", 1);
    fill2d_kernel<<<dim3((uint32_t)nz), kFillBlock, 0, stream>>>(a);
    FA_HIP(hipGetLastError());
    collect_stats(stats, nz, h_nChanged, stream, "fill2d");
    return true;
}

}  // namespace

namespace {

// one run of the sweeps over whole slices [nz][ny][nx]; d_defaults (device, per slice) replaces the first guess
void run_creepfill_whole(size_t nx, size_t ny, size_t nz, float* d_field, bool useDefault, float defaultVal,
                         unsigned short repeat, char setWeight, size_t* h_nChanged, hipStream_t stream, const double* d_defaults,
                         const unsigned long long* d_bounds = nullptr)
{
    if (nx * ny == 0 || nz == 0) return;  // :1380
    FA_REQUIRE(nx <= 0x7FFFFFFFu && ny <= 0x7FFFFFFFu && nz <= 0x7FFFFFFFu, "creepfill: slice too large");
    DeviceArray<SliceStats> stats(nz);
    FA_HIP(hipMemsetAsync(stats.get(), 0, nz * sizeof(SliceStats), stream));
    const char* what = useDefault ? "creepfillval2d" : "creepfill2d";
    const size_t nBands = ny > 2 ? (ny - 2 + kWave - 1) / kWave : 0;
    const uint32_t mws = (uint32_t)((nx + kWave + 31) / 32 + 2);
    const size_t gens = (size_t)repeat + 1;
    const size_t uWords = nz * gens * ny * mws;
    // the systolic kernel keeps repeat + 1 generations of the "updated" mask; very long repeats take the counter kernel
    if (tuning("CREEP_V2", 1) != 0 && nx >= 4 && ny >= 4 && nBands < (size_t)kMaxBands && nx < (1u << 19) &&
        (size_t)(kWave + 2) * nx * 4 < 0xFFFFFFFFull && setWeight >= 0 && uWords * 4 <= ((size_t)8 << 30)) {
        DeviceArray<uint32_t> maskD(nz * ny * mws), maskU(uWords);
        FA_HIP(hipMemsetAsync(maskU.get(), 0, uWords * sizeof(uint32_t), stream));
        CreepV2Args a{};
        a.field = d_field;
        a.maskD = maskD.get();
        a.maskU = maskU.get();
        a.stats = stats.get();
        a.nx = (uint32_t)nx;
        a.ny = (uint32_t)ny;
        a.mws = mws;
        a.gens = (uint32_t)gens;
        a.useDefault = useDefault ? 1 : 0;
        a.defaultVal = defaultVal;
        a.repeat = repeat;
        a.setWeight = (int)setWeight;
        a.sumAlgo = tuning("SUM_ALGO", 1);
        a.skipIdle = tuning("CREEP_SKIP", 1);
        constexpr size_t ldsBytes = (size_t)kCreepWaves * kWave * kCreepPitch * sizeof(float) + (size_t)kCreepWaves * 2 * kHandWC * (sizeof(float) + 1) +
                                    (size_t)kCreepWaves * 4 * sizeof(unsigned int);
        launch_fill_prologue(true, d_field, stats.get(), nx, ny, nz, maskD.get(), mws, nullptr, nullptr, false, useDefault, defaultVal, 0.f, stream, d_defaults, d_bounds);
        DeviceArray<unsigned int> error(1);
        FA_HIP(hipMemsetAsync(error.get(), 0, sizeof(unsigned int), stream));
        a.error = error.get();
        // small batches: the bands of a slice on several workgroups (see run_fill2d)
        const size_t bandGroups = ceil_div(nBands, (size_t)kCreepWaves);
        const size_t perXcd = ceil_div(nz, (size_t)kXcds);
        int cus = 0, dev = 0;
        FA_HIP(hipGetDevice(&dev));
        FA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        size_t groups = std::min(bandGroups, perXcd ? (size_t)std::max(1, cus / kXcds) / perXcd : (size_t)1);
        if (tuning("FILL_MULTI", 1) == 0 || groups < 2) groups = 1;
        DeviceArray<unsigned int> sync;
        if (groups > 1) {
            a.syncStride = (uint32_t)(4 + nBands);
            a.groups = (uint32_t)groups;
            a.nz = (uint32_t)nz;
            sync.allocate(nz * a.syncStride);
            FA_HIP(hipMemsetAsync(sync.get(), 0, sync.bytes(), stream));
            a.sync = sync.get();
            const void* kernel = reinterpret_cast<const void*>(&creepfill_kernel_v3);
            allow_dynamic_lds(kernel, ldsBytes);
            void* params[] = {&a};
            if (!launch_resident(kernel, dim3((uint32_t)(kXcds * groups * perXcd)), dim3(kCreepThreads), params, ldsBytes, stream))
                groups = 1;  // not every workgroup would be resident: one workgroup per slice
        }
        if (groups <= 1) {
            allow_dynamic_lds(reinterpret_cast<const void*>(&creepfill_kernel_v2), ldsBytes);
            creepfill_kernel_v2<<<dim3((uint32_t)nz), kCreepThreads, ldsBytes, stream>>>(a);
        }
        FA_HIP(hipGetLastError());
        unsigned int failed = 0;
        FA_HIP(hipMemcpyAsync(&failed, error.get(), sizeof(failed), hipMemcpyDeviceToHost, stream));
        collect_stats(stats, nz, h_nChanged, stream, what);  // synchronises the stream
        FA_REQUIRE(failed == 0, std::string(what) + ": a hand-off between waves did not arrive (wait " + std::to_string(failed) +
                                    " gave up); the field is not valid");
        return;
    }
    DeviceArray<signed char> w(nx * ny * nz);
    DeviceArray<unsigned short> r(nx * ny * nz);
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
    a.sumAlgo = tuning("SUM_ALGO", 1);
    a.defaults = d_defaults;
    a.bounds = d_bounds;
    creepfill_kernel<<<dim3((uint32_t)nz), kFillBlock, 0, stream>>>(a);
    FA_HIP(hipGetLastError());
    collect_stats(stats, nz, h_nChanged, stream, what);
}

// ---- creep fill by rectangles -----------------------------------------------------------------------------------------------
// A cell that is defined on entry never changes and is read with the same weight (setWeight) whether it lies on a border or
// inside (src/interpolation.c:1408-1461).  Rows and columns that are defined throughout therefore cut the field into rectangles
// whose sweeps do not see each other, and a sweep that changes nothing ends a rectangle's loop without touching the others'
// results (further sweeps over a finished region are no-ops).  The reference sweeps the whole field until nothing changes
// anywhere: a region outside the source domain that lies ABOVE defined cells is filled one row per sweep (in-place, row-major:
// values travel down and right within a sweep, up and left one cell per sweep) -- 185 sweeps over 3000 x 3000 cells for the
// configs[4] field, of which a tenth of the field needs more than 22.  Here every rectangle (bounding box of a run of rows
// with undefined cells x a run of columns with undefined cells inside those rows, plus the defined ring around it, or the
// field's own border) is copied out, filled with the whole slice's first guess as a field of its own, and copied back.
using creep_rects::Rect;
using creep_rects::slice_rects;

// one wave per row: bit x of the row's words = cell x is undefined
// rowCount[row] = undefined cells of the row
// rowSpecial (may be null): defined cells of the row that hold -0.0 or an infinity
__global__ void __launch_bounds__(kBlock) nan_bitmap_kernel(const float* __restrict__ field, uint32_t nx, size_t rows, uint32_t words,
                                                            uint32_t* __restrict__ bits, uint32_t* __restrict__ rowCount, uint32_t* __restrict__ rowSpecial = nullptr)
{
    const size_t row = (size_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (row >= rows) return;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const float* f = field + row * nx;
    uint32_t* out = bits + row * words;
    uint32_t count = 0, special = 0;
    for (uint32_t base = 0; base < words * 32; base += kWave) {
        const uint32_t x = base + lane;
        const float v = x < nx ? f[x] : 0.f;
        const unsigned long long m = __ballot(x < nx && isnan(v));
        special += (uint32_t)__popcll(__ballot(x < nx && (__float_as_uint(v) == 0x80000000u || isinf(v))));
        count += (uint32_t)__popcll(m);
        if (lane == 0) {
            out[base / 32] = (uint32_t)m;
            if (base / 32 + 1 < words) out[base / 32 + 1] = (uint32_t)(m >> 32);
        }
    }
    if (lane == 0) {
        rowCount[row] = count;
        if (rowSpecial) rowSpecial[row] = special;
    }
}

struct RectCopyArgs {
    float* field;      // [nz][ny][nx], first slice of the group
    float* box;        // [count][boxH][boxW]
    size_t total;      // nx * ny
    uint32_t nx, w, h, xa, ya;
    int back;
    uint32_t boxW, boxH, ox, oy;  // the rectangle sits at (ox, oy) of its box (fill2d pads rectangles to one size)
};
__global__ void __launch_bounds__(kBlock) rect_copy_kernel(RectCopyArgs a)
{
    const uint32_t y = blockIdx.x % a.h, s = blockIdx.x / a.h;
    float* src = a.field + (size_t)s * a.total + (size_t)(a.ya + y) * a.nx + a.xa;
    float* box = a.box + ((size_t)s * a.boxH + a.oy + y) * a.boxW + a.ox;
    for (uint32_t x = threadIdx.x; x < a.w; x += kBlock) {
        if (a.back) src[x] = box[x];
        else box[x] = src[x];
    }
}
// boxes [count][cells] filled with one value per box
__global__ void __launch_bounds__(kBlock) box_fill_kernel(float* __restrict__ box, size_t cells, const double* __restrict__ values)
{
    const float v = (float)values[blockIdx.y];
    float* b = box + (size_t)blockIdx.y * cells;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < cells; i += (size_t)gridDim.x * kBlock) b[i] = v;
}

}  // namespace

void run_creepfill(size_t nx, size_t ny, size_t nz, float* d_field, bool useDefault, float defaultVal,
                   unsigned short repeat, char setWeight, size_t* h_nChanged, hipStream_t stream)
{
    if (nx * ny == 0 || nz == 0) return;  // :1380
    const size_t total = nx * ny;
    // worth looking for rectangles: large slices (the decomposition costs a pass over the data and a host round trip)
    if (tuning("CREEP_RECTS", 1) == 0 || nx < 64 || ny < 64 || nx > 0x7FFFFFFFu || ny > 0x7FFFFFFFu || total * nz > ((size_t)1 << 33)) {
        run_creepfill_whole(nx, ny, nz, d_field, useDefault, defaultVal, repeat, setWeight, h_nChanged, stream, nullptr);
        return;
    }
    const uint32_t words = (uint32_t)(ceil_div(nx, (size_t)64) * 2);
    DeviceArray<uint32_t> d_bits(nz * ny * words), d_rowCount(nz * ny);
    nan_bitmap_kernel<<<dim3((uint32_t)ceil_div(nz * ny, (size_t)(kBlock / kWave))), kBlock, 0, stream>>>(d_field, (uint32_t)nx, nz * ny, words, d_bits.get(), d_rowCount.get());
    FA_HIP(hipGetLastError());
    // first the rows' counts (a few KB): holes scattered over (nearly) all rows of a slice leave nothing to cut -- the usual case pays a pass
    // over the data on the device and this copy, not the bitmap's
    std::vector<uint32_t> rowCount(nz * ny);
    FA_HIP(hipMemcpyAsync(rowCount.data(), d_rowCount.get(), rowCount.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    bool all = true;
    size_t withHoles = 0;
    std::vector<unsigned char> consider(nz, 0);  // slices with undefined AND defined cells (the others are left alone, :1384-1386)
    for (size_t z = 0; z < nz && all; ++z) {
        size_t dirtyRows = 0, undefined = 0;
        for (size_t y = 0; y < ny; ++y) { dirtyRows += rowCount[z * ny + y] != 0; undefined += rowCount[z * ny + y]; }
        if (undefined == 0 || undefined == total) continue;
        consider[z] = 1;
        withHoles++;
        if (dirtyRows * 10 > ny * 9) all = false;
    }
    std::vector<std::vector<Rect>> rects(nz);
    std::vector<uint32_t> bits;
    if (all && withHoles != 0) {
        bits.resize(nz * ny * words);
        FA_HIP(hipMemcpyAsync(bits.data(), d_bits.get(), bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        FA_HIP(hipStreamSynchronize(stream));
        for (size_t z = 0; z < nz && all; ++z)
            if (consider[z]) all = slice_rects(bits.data() + z * ny * words, (uint32_t)nx, (uint32_t)ny, words, rects[z]);
    }
    if (!all || withHoles == 0) {
        FA_REQUIRE(tuning("CREEP_RECTS", 1) != 2 || withHoles == 0, "creepfill: CREEP_RECTS=2 (tests) asks for a field that can be cut into rectangles");
        run_creepfill_whole(nx, ny, nz, d_field, useDefault, defaultVal, repeat, setWeight, h_nChanged, stream, nullptr);
        return;
    }
    // the whole slices' statistics: the first guess (mean of the defined cells in scan order, :1502-1516) and *nChanged
    DeviceArray<SliceStats> stats(nz);
    FA_HIP(hipMemsetAsync(stats.get(), 0, nz * sizeof(SliceStats), stream));
    {
        FillStatsArgs fs{};
        fs.field = d_field;
        fs.stats = stats.get();
        fs.total = total;
        fs.useDefault = useDefault ? 1 : 0;
        fs.defaultVal = defaultVal;
        fs.sumAlgo = tuning("SUM_ALGO", 1);
        if (fs.sumAlgo > 1) fs.sumAlgo = 1;
        fill_stats_kernel<<<dim3((uint32_t)nz), kFillBlock, 0, stream>>>(fs);
        FA_HIP(hipGetLastError());
    }
    std::vector<SliceStats> h_stats(nz);
    FA_HIP(hipMemcpyAsync(h_stats.data(), stats.get(), nz * sizeof(SliceStats), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    std::vector<double> defaults(nz);
    std::vector<unsigned long long> bounds(nz);  // the loop of a rectangle ends where the whole slice's would (:1430)
    for (size_t z = 0; z < nz; ++z) {
        if (h_nChanged) h_nChanged[z] = (size_t)h_stats[z].nUndef;
        defaults[z] = h_stats[z].average;
        bounds[z] = h_stats[z].sweepBound;
    }
    // groups of consecutive slices with the same rectangles (masks usually do not change from slice to slice)
    for (size_t z0 = 0; z0 < nz;) {
        size_t z1 = z0 + 1;
        while (z1 < nz && rects[z1] == rects[z0] && h_stats[z1].skip == h_stats[z0].skip) ++z1;
        if (!h_stats[z0].skip) {  // (skip: nothing defined or nothing undefined, :1384-1386)
            const size_t count = z1 - z0;
            // rectangles of one size go through the sweeps together, as further slices of one run
            std::vector<char> done(rects[z0].size(), 0);
            for (size_t i = 0; i < rects[z0].size(); ++i) {
                if (done[i]) continue;
                const Rect& r = rects[z0][i];
                const size_t w = r.xb - r.xa + 1, h = r.yb - r.ya + 1;
                std::vector<size_t> same;
                for (size_t j = i; j < rects[z0].size(); ++j) {
                    const Rect& q = rects[z0][j];
                    if (!done[j] && q.xb - q.xa + 1 == w && q.yb - q.ya + 1 == h) { same.push_back(j); done[j] = 1; }
                }
                const size_t boxes = same.size() * count;
                FA_REQUIRE(boxes * h <= 0x7FFFFFFFull, "creepfill: too many rows for one copy");
                DeviceArray<float> box(boxes * w * h);
                std::vector<double> hd(boxes);
                std::vector<unsigned long long> hb(boxes);
                for (size_t k = 0; k < same.size(); ++k)
                    for (size_t c = 0; c < count; ++c) { hd[k * count + c] = defaults[z0 + c]; hb[k * count + c] = bounds[z0 + c]; }
                DeviceArray<double> d_def(boxes);
                DeviceArray<unsigned long long> d_bnd(boxes);
                FA_HIP(hipMemcpyAsync(d_def.get(), hd.data(), boxes * sizeof(double), hipMemcpyHostToDevice, stream));
                FA_HIP(hipMemcpyAsync(d_bnd.get(), hb.data(), boxes * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
                auto copy = [&](int back) {
                    for (size_t k = 0; k < same.size(); ++k) {
                        const Rect& q = rects[z0][same[k]];
                        if (kTuningBuild && !back && tuning("CREEP_RECTS", 1) == 3)
                            fprintf(stderr, "creepfill: slices %zu..%zu rectangle x %u..%u y %u..%u\n", z0, z1 - 1, q.xa, q.xb, q.ya, q.yb);
                        RectCopyArgs c{d_field + z0 * total, box.get() + k * count * w * h, total, (uint32_t)nx, (uint32_t)w, (uint32_t)h, q.xa, q.ya, back, (uint32_t)w, (uint32_t)h, 0u, 0u};
                        rect_copy_kernel<<<dim3((uint32_t)(count * h)), kBlock, 0, stream>>>(c);
                        FA_HIP(hipGetLastError());
                    }
                };
                copy(0);
                run_creepfill_whole(w, h, boxes, box.get(), true, 0.f, repeat, setWeight, nullptr, stream, d_def.get(), d_bnd.get());  // synchronises
                copy(1);
                FA_HIP(hipStreamSynchronize(stream));  // box, hd, hb are released at the end of the iteration
            }
        }
        z0 = z1;
    }
    FA_HIP(hipStreamSynchronize(stream));
}

void run_scan_sum(const float* d_values, size_t n, int mode, double average, int algo, double* h_sum, size_t* h_nUndefined, hipStream_t stream)
{
    FA_REQUIRE(mode >= 0 && mode <= 2 && algo >= 0 && algo <= 2, "scan_sum: mode 0..2, algo 0..2");
    DeviceArray<double> d_sum(1);
    DeviceArray<unsigned long long> d_undef(1);
    if (algo == 2 && n > 0) {
        const SumBuffers buffers(n, 1);
        SumJob j{d_values, n, mode, nullptr, average};
        StitchOut o{};
        o.sum = d_sum.get();
        o.nUndef = d_undef.get();
        launch_chip_sum(j, buffers, 1, o, stream);
        FA_HIP(hipStreamSynchronize(stream));
    } else {
        ScanSumArgs a{d_values, n, mode, algo == 2 ? 1 : algo, average, d_sum.get(), d_undef.get()};
        scan_sum_kernel<<<1, kFillBlock, 0, stream>>>(a);
        FA_HIP(hipGetLastError());
    }
    unsigned long long u = 0;
    FA_HIP(hipMemcpyAsync(h_sum, d_sum.get(), sizeof(double), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipMemcpyAsync(&u, d_undef.get(), sizeof(u), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    if (h_nUndefined) *h_nUndefined = (size_t)u;
}

// fill2d by rectangles.  The same cut as for the creep fills (rows and columns that are defined throughout never change: w = 0,
// src/interpolation.c:1288-1315), with two differences.  The sweeps of the reference end by a criterion over the WHOLE field
// (:1338-1359), so the rectangles of a slice sweep in lock-step: they are padded to one size with defined cells (which change
// nothing), run as slices of ONE launch and end together by the criterion over all of them (Fill2dV2Args::couple).  And the
// reference's sweep adds e * 0 to every defined cell (:1327): that turns a -0.0 into +0.0 and, next to an infinity, a value into
// NaN -- fields with such defined cells are not cut.
void run_fill2d(size_t nx, size_t ny, size_t nz, float* d_field, float relaxCrit, float corrEff, size_t maxLoop,
                size_t* h_nChanged, hipStream_t stream)
{
    if (nx * ny == 0 || nz == 0) return;  // :1248
    const size_t total = nx * ny;
    auto whole = [&]() { (void)run_fill2d_whole(nx, ny, nz, d_field, relaxCrit, corrEff, maxLoop, h_nChanged, stream, nullptr, nullptr, 0); };
    if (tuning("FILL_RECTS", 1) == 0 || nx < 64 || ny < 64 || nx > 0x7FFFFFFFu || ny > 0x7FFFFFFFu || total * nz > ((size_t)1 << 33) || maxLoop == 0) {
        whole();
        return;
    }
    const bool required = tuning("FILL_RECTS", 1) == 2;  // tests: fail instead of falling back
    const uint32_t words = (uint32_t)(ceil_div(nx, (size_t)64) * 2);
    DeviceArray<uint32_t> d_bits(nz * ny * words), d_rowCount(2 * nz * ny);
    nan_bitmap_kernel<<<dim3((uint32_t)ceil_div(nz * ny, (size_t)(kBlock / kWave))), kBlock, 0, stream>>>(d_field, (uint32_t)nx, nz * ny, words, d_bits.get(), d_rowCount.get(),
                                                                                                     d_rowCount.get() + nz * ny);
    FA_HIP(hipGetLastError());
    std::vector<uint32_t> rowCount(2 * nz * ny);
    FA_HIP(hipMemcpyAsync(rowCount.data(), d_rowCount.get(), rowCount.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    bool all = true;
    size_t withHoles = 0;
    std::vector<unsigned char> consider(nz, 0);
    for (size_t z = 0; z < nz && all; ++z) {
        size_t dirtyRows = 0, undefined = 0, special = 0;
        for (size_t y = 0; y < ny; ++y) { dirtyRows += rowCount[z * ny + y] != 0; undefined += rowCount[z * ny + y]; special += rowCount[(nz + z) * ny + y]; }
        if (undefined == 0 || undefined == total) continue;  // :1266-1269
        consider[z] = 1;
        withHoles++;
        if (dirtyRows * 10 > ny * 9 || special != 0) all = false;
    }
    std::vector<std::vector<Rect>> rects(nz);
    if (all && withHoles != 0) {
        std::vector<uint32_t> bits(nz * ny * words);
        FA_HIP(hipMemcpyAsync(bits.data(), d_bits.get(), bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        FA_HIP(hipStreamSynchronize(stream));
        for (size_t z = 0; z < nz && all; ++z) {
            if (!consider[z]) continue;
            all = slice_rects(bits.data() + z * ny * words, (uint32_t)nx, (uint32_t)ny, words, rects[z]);
            // padded to one size: what the boxes of this slice cover
            size_t mw = 0, mh = 0;
            for (const Rect& r : rects[z]) { mw = std::max<size_t>(mw, r.xb - r.xa + 1); mh = std::max<size_t>(mh, r.yb - r.ya + 1); }
            if (all && rects[z].size() * mw * mh * 2 > total) all = false;
        }
    }
    if (!all || withHoles == 0) {
        FA_REQUIRE(!required || withHoles == 0, "fill2d: FILL_RECTS=2 (tests) asks for a field that can be cut into rectangles");
        whole();
        return;
    }
    // the whole slices' statistics: first guess and criterion (:1281-1305), *nChanged
    DeviceArray<SliceStats> stats(nz);
    FA_HIP(hipMemsetAsync(stats.get(), 0, nz * sizeof(SliceStats), stream));
    {
        FillStatsArgs fs{};
        fs.field = d_field;
        fs.stats = stats.get();
        fs.total = total;
        fs.wantDeviation = 1;
        fs.relaxCrit = relaxCrit;
        fs.sumAlgo = 1;
        fill_stats_kernel<<<dim3((uint32_t)nz), kFillBlock, 0, stream>>>(fs);
        FA_HIP(hipGetLastError());
    }
    std::vector<SliceStats> h_stats(nz);
    FA_HIP(hipMemcpyAsync(h_stats.data(), stats.get(), nz * sizeof(SliceStats), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    // the boxes are filled before anything of the field is written: where a group cannot be launched, the whole call takes the other path
    struct Group { size_t z0, z1, mw, mh; DeviceArray<float> box; };
    std::vector<Group> groups;
    bool ok = true;
    int cus = 0, dev = 0;
    FA_HIP(hipGetDevice(&dev));
    FA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    for (size_t z0 = 0; z0 < nz && ok;) {
        size_t z1 = z0 + 1;
        // (the coupled boxes of a launch wait for each other: one workgroup per box at least, a CU each -- long batches go in several launches)
        const size_t most = std::max<size_t>(1, (size_t)cus / std::max<size_t>(1, rects[z0].size()));
        while (z1 < nz && z1 - z0 < most && rects[z1] == rects[z0] && h_stats[z1].skip == h_stats[z0].skip) ++z1;
        if (!h_stats[z0].skip && !rects[z0].empty()) {
            Group gr{z0, z1, 0, 0, {}};
            for (const Rect& r : rects[z0]) { gr.mw = std::max<size_t>(gr.mw, r.xb - r.xa + 1); gr.mh = std::max<size_t>(gr.mh, r.yb - r.ya + 1); }
            const size_t count = z1 - z0, nr = rects[z0].size(), boxes = nr * count, cells = gr.mw * gr.mh;
            if (boxes * gr.mh > 0x7FFFFFFFull || boxes > 65535) { ok = false; break; }
            // a rectangle that spans the field from border to border must be as wide (high) as its box
            for (const Rect& q : rects[z0])
                if ((q.xa == 0 && q.xb == nx - 1 && gr.mw != nx) || (q.ya == 0 && q.yb == ny - 1 && gr.mh != ny)) ok = false;
            if (!ok) break;
            gr.box.allocate(boxes * cells);
            std::vector<double> hd(boxes), hv(boxes);
            for (size_t k = 0; k < nr; ++k)
                for (size_t c = 0; c < count; ++c) { hd[k * count + c] = h_stats[z0 + c].average; hv[k * count + c] = h_stats[z0 + c].meanAbsDev; }
            DeviceArray<double> d_def(boxes), d_dev(boxes);
            FA_HIP(hipMemcpyAsync(d_def.get(), hd.data(), boxes * sizeof(double), hipMemcpyHostToDevice, stream));
            FA_HIP(hipMemcpyAsync(d_dev.get(), hv.data(), boxes * sizeof(double), hipMemcpyHostToDevice, stream));
            box_fill_kernel<<<dim3((uint32_t)std::min<size_t>(ceil_div(cells, (size_t)kBlock), 1024), (uint32_t)boxes), kBlock, 0, stream>>>(gr.box.get(), cells, d_def.get());
            FA_HIP(hipGetLastError());
            auto copy = [&](int back) {
                for (size_t k = 0; k < nr; ++k) {
                    const Rect& q = rects[z0][k];
                    const size_t w = q.xb - q.xa + 1, h = q.yb - q.ya + 1;
                    // a side on the field's border stays on the box's border (its cells are the ones :1363-1370 work on)
                    const uint32_t ox = (q.xb == nx - 1 && q.xa != 0) ? (uint32_t)(gr.mw - w) : 0u, oy = (q.yb == ny - 1 && q.ya != 0) ? (uint32_t)(gr.mh - h) : 0u;
                    if (kTuningBuild && !back && tuning("FILL_RECTS", 1) == 3)
                        fprintf(stderr, "fill2d: slices %zu..%zu rectangle x %u..%u y %u..%u in boxes of %zu x %zu\n", z0, z1 - 1, q.xa, q.xb, q.ya, q.yb, gr.mw, gr.mh);
                    RectCopyArgs c{d_field + z0 * total, gr.box.get() + k * count * cells, total, (uint32_t)nx, (uint32_t)w, (uint32_t)h, q.xa, q.ya, back,
                                   (uint32_t)gr.mw, (uint32_t)gr.mh, ox, oy};
                    rect_copy_kernel<<<dim3((uint32_t)(count * h)), kBlock, 0, stream>>>(c);
                    FA_HIP(hipGetLastError());
                }
            };
            copy(0);
            ok = run_fill2d_whole(gr.mw, gr.mh, boxes, gr.box.get(), relaxCrit, corrEff, maxLoop, nullptr, stream, d_def.get(), d_dev.get(), (uint32_t)count);  // synchronises
            if (!ok) break;
            groups.push_back(std::move(gr));
            // (copied back below, once every group has run: the field is untouched until then)
            (void)copy;
        }
        z0 = z1;
    }
    if (!ok) {
        FA_REQUIRE(!required, "fill2d: FILL_RECTS=2 (tests): a group of rectangles could not be launched");
        whole();
        return;
    }
    for (Group& gr : groups) {
        const size_t count = gr.z1 - gr.z0, cells = gr.mw * gr.mh;
        for (size_t k = 0; k < rects[gr.z0].size(); ++k) {
            const Rect& q = rects[gr.z0][k];
            const size_t w = q.xb - q.xa + 1, h = q.yb - q.ya + 1;
            const uint32_t ox = (q.xb == nx - 1 && q.xa != 0) ? (uint32_t)(gr.mw - w) : 0u, oy = (q.yb == ny - 1 && q.ya != 0) ? (uint32_t)(gr.mh - h) : 0u;
            RectCopyArgs c{d_field + gr.z0 * total, gr.box.get() + k * count * cells, total, (uint32_t)nx, (uint32_t)w, (uint32_t)h, q.xa, q.ya, 1,
                           (uint32_t)gr.mw, (uint32_t)gr.mh, ox, oy};
            rect_copy_kernel<<<dim3((uint32_t)(count * h)), kBlock, 0, stream>>>(c);
            FA_HIP(hipGetLastError());
        }
    }
    FA_HIP(hipStreamSynchronize(stream));
    for (size_t z = 0; z < nz; ++z)
        if (h_nChanged) h_nChanged[z] = (size_t)h_stats[z].nUndef;
}

}  // namespace fimex_amd
