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
__device__ double serial_sum(const float* __restrict__ f, size_t total, int mode, double average, float* lds,
                             unsigned long long* nUndefOut)
{
    double sum = 0;
    unsigned long long nUndef = 0;
    for (size_t base = 0; base < total; base += kTile) {
        const size_t len = (total - base < (size_t)kTile) ? total - base : (size_t)kTile;
        __syncthreads();
        for (size_t i = threadIdx.x; i < len; i += kFillBlock) lds[i] = f[base + i];
        __syncthreads();
        if (threadIdx.x < kWave) {
            for (size_t i = 0; i < len; ++i) {
                const float v = lds[i];
                const bool undef = isnan(v);
                nUndef += undef;
                if (!undef) {
                    if (mode == 0) sum += v;
                    else sum += fabs(v - average);
                }
            }
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
    __shared__ float lds[kTile];
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
    __shared__ float lds[kTile];
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
    DeviceArray<float> w(nx * ny * nz);
    DeviceArray<SliceStats> stats(nz);
    FA_HIP(hipMemsetAsync(stats.get(), 0, nz * sizeof(SliceStats), stream));
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
