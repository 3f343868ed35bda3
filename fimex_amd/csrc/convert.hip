// The edges of the path: fill value <-> NaN, and coordinate -> fractional axis index.
//
// Replaces mifi_bad2nanf / mifi_nanf2bad (src/interpolation.c:1775-1793) and
// mifi_points2position (src/interpolation.c:104-217).
#include "plan.hpp"

namespace fimex_amd {

namespace {

constexpr double kPi = 3.1415926535897932384626433832795;  // MIFI_PI, include/fimex/mifi_constants.h:42

__device__ __forceinline__ float undefined_f() { return __uint_as_float(0x7fc00000u); }

template <bool TO_NAN>
__device__ __forceinline__ float4 replace4(float4 v, float bad)
{
    if (TO_NAN) {
        v.x = (v.x == bad) ? undefined_f() : v.x;  // :1778
        v.y = (v.y == bad) ? undefined_f() : v.y;
        v.z = (v.z == bad) ? undefined_f() : v.z;
        v.w = (v.w == bad) ? undefined_f() : v.w;
    } else {
        v.x = isnan(v.x) ? bad : v.x;              // :1788
        v.y = isnan(v.y) ? bad : v.y;
        v.z = isnan(v.z) ? bad : v.z;
        v.w = isnan(v.w) ? bad : v.w;
    }
    return v;
}

template <bool TO_NAN>
__global__ void __launch_bounds__(kBlock) replace_kernel(float* __restrict__ d, size_t n, float bad)
{
    // grid-stride over float4 groups, four independent 16-byte loads in flight per lane, scalar tail
    const size_t n4 = n / 4;
    float4* d4 = reinterpret_cast<float4*>(d);
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a = d4[i], b = d4[i + stride], c = d4[i + 2 * stride], e = d4[i + 3 * stride];
        d4[i] = replace4<TO_NAN>(a, bad);
        d4[i + stride] = replace4<TO_NAN>(b, bad);
        d4[i + 2 * stride] = replace4<TO_NAN>(c, bad);
        d4[i + 3 * stride] = replace4<TO_NAN>(e, bad);
    }
    for (; i < n4; i += stride) d4[i] = replace4<TO_NAN>(d4[i], bad);
    for (size_t j = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const float v = d[j];
        d[j] = TO_NAN ? ((v == bad) ? undefined_f() : v) : (isnan(v) ? bad : v);
    }
}

template <bool TO_NAN>
__global__ void __launch_bounds__(kBlock) replace_scalar_kernel(float* __restrict__ d, size_t n, float bad)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = d[i];
        d[i] = TO_NAN ? ((v == bad) ? undefined_f() : v) : (isnan(v) ? bad : v);
    }
}

template <bool TO_NAN>
void launch_replace(float* d, size_t n, float bad, hipStream_t stream)
{
    if (n == 0 || bad != bad) return;  // a NaN fill value leaves the data untouched (:1776, :1786)
    const size_t groups = ceil_div(n, (size_t)4 * kBlock);
    const uint32_t blocks = (uint32_t)(groups < 256 * 8 ? (groups ? groups : 1) : 256 * 8);
    if (reinterpret_cast<uintptr_t>(d) % 16 == 0) replace_kernel<TO_NAN><<<blocks, kBlock, 0, stream>>>(d, n, bad);
    else replace_scalar_kernel<TO_NAN><<<blocks, kBlock, 0, stream>>>(d, n, bad);
    FA_HIP(hipGetLastError());
}

// axis order compare, src/interpolation.c:104-117: dir = +1 ascending, -1 descending
__device__ __forceinline__ int axis_compare(double key, double elem, int dir)
{
    const int c = (key > elem) ? 1 : ((key == elem) ? 0 : -1);
    return dir * c;
}

struct P2PArgs {
    double* points;
    size_t n;
    const double* axis;
    int num;
    int dir;
    int lonShift;   // 0 none, 1: points > pi -= 2pi, 2: points < 0 += 2pi  (:155-167)
    int circular;   // :168-179
};

__global__ void __launch_bounds__(kBlock) points2position_kernel(P2PArgs a)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        double p = a.points[i];
        if (a.lonShift == 1) { if (p > kPi) p -= 2 * kPi; }
        else if (a.lonShift == 2) { if (p < 0) p += 2 * kPi; }
        if (!isfinite(p)) { a.points[i] = -999.; continue; }  // :183-186
        // binary search, same probe sequence as bsearchDoubleIndex (:124-146)
        int lo = 0, hi = a.num - 1, mid = 0, c = 0;
        while (lo <= hi) {
            mid = (lo + hi) / 2;
            c = axis_compare(p, a.axis[mid], a.dir);
            if (c > 0) lo = mid + 1;
            else if (c < 0) hi = mid - 1;
            else break;
        }
        if (c == 0) { a.points[i] = (double)mid; continue; }
        int np = (c > 0) ? mid + 1 : mid;  // insertion point (:144-145, :192)
        if (np == a.num) np--;             // extrapolate to the right
        else if (np == 0) np++;            // extrapolate to the left
        const double slope = a.axis[np] - a.axis[np - 1];   // :199
        const double offset = a.axis[np] - (slope * np);    // :200
        double ap = (p - offset) / slope;                   // :201
        if (a.circular && ap <= -0.5) ap += a.num;          // :202-204
        if (a.circular && ap > (a.num - 0.5)) ap -= a.num;  // :205-207
        a.points[i] = ap;
    }
}

}  // namespace

void launch_bad2nan(float* d, size_t n, float bad, hipStream_t stream) { launch_replace<true>(d, n, bad, stream); }
void launch_nan2bad(float* d, size_t n, float bad, hipStream_t stream) { launch_replace<false>(d, n, bad, stream); }

void launch_points2position(double* d_points, size_t n, const double* axis, int num, int axisType, hipStream_t stream)
{
    if (n == 0) return;
    FA_REQUIRE(num >= 2, "points2position needs an axis of at least 2 values");
    P2PArgs a{};
    a.points = d_points;
    a.n = n;
    a.num = num;
    a.dir = (axis[0] < axis[num - 1]) ? 1 : -1;  // :152-153
    if (axisType == FIMEX_AMD_LONGITUDE) {
        a.lonShift = (axis[0] < 0 || axis[num - 1] < 0) ? 1 : 2;   // :157-167
        double next = axis[num - 1] + (axis[1] - axis[0]) * 1.01;  // :168
        if (a.dir > 0) { next -= 2 * kPi; if (next >= axis[0]) a.circular = 1; }
        else { next += 2 * kPi; if (next <= axis[0]) a.circular = 1; }
    }
    DeviceArray<double> d_axis((size_t)num);
    FA_HIP(hipMemcpyAsync(d_axis.get(), axis, (size_t)num * sizeof(double), hipMemcpyHostToDevice, stream));
    a.axis = d_axis.get();
    const size_t want = ceil_div(n, kBlock);
    const uint32_t blocks = (uint32_t)(want < 256 * 8 ? want : 256 * 8);
    points2position_kernel<<<blocks, kBlock, 0, stream>>>(a);
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(stream));  // d_axis is released on return
}

}  // namespace fimex_amd
