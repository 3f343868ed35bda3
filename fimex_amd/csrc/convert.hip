// The edges of the path: fill value <-> NaN, and coordinate -> fractional axis index.
//
// Replaces mifi_bad2nanf / mifi_nanf2bad (src/interpolation.c:1775-1793) and
// mifi_points2position (src/interpolation.c:104-217).
#include "plan.hpp"
#include "typed_convert.hpp"

#include <cmath>
#include <string>
#include <type_traits>

namespace fimex_amd {

namespace {

constexpr double kPi = 3.1415926535897932384626433832795;  // MIFI_PI, include/fimex/mifi_constants.h:42

__device__ __forceinline__ float undefined_f() { return __uint_as_float(0x7fc00000u); }

template <bool TO_NAN>
__device__ __forceinline__ bool hit(float v, float bad) { return TO_NAN ? (v == bad) : isnan(v); }  // :1778 / :1788

// replaces in place; returns whether anything changed (unchanged groups are not written back: fill values are rare, so
// the pass is mostly a read)
template <bool TO_NAN>
__device__ __forceinline__ bool replace4(float4& v, float bad)
{
    const float to = TO_NAN ? undefined_f() : bad;
    const bool hx = hit<TO_NAN>(v.x, bad), hy = hit<TO_NAN>(v.y, bad), hz = hit<TO_NAN>(v.z, bad), hw = hit<TO_NAN>(v.w, bad);
    v.x = hx ? to : v.x;
    v.y = hy ? to : v.y;
    v.z = hz ? to : v.z;
    v.w = hw ? to : v.w;
    return hx || hy || hz || hw;
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
        if (replace4<TO_NAN>(a, bad)) d4[i] = a;
        if (replace4<TO_NAN>(b, bad)) d4[i + stride] = b;
        if (replace4<TO_NAN>(c, bad)) d4[i + 2 * stride] = c;
        if (replace4<TO_NAN>(e, bad)) d4[i + 3 * stride] = e;
    }
    for (; i < n4; i += stride) {
        float4 a = d4[i];
        if (replace4<TO_NAN>(a, bad)) d4[i] = a;
    }
    for (size_t j = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const float v = d[j];
        if (hit<TO_NAN>(v, bad)) d[j] = TO_NAN ? undefined_f() : bad;
    }
}

template <bool TO_NAN>
__global__ void __launch_bounds__(kBlock) replace_scalar_kernel(float* __restrict__ d, size_t n, float bad)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float v = d[i];
        if (hit<TO_NAN>(v, bad)) d[i] = TO_NAN ? undefined_f() : bad;
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

// ---- typed edges of a slice: data2InterpolationArray / interpolationArray2Data, src/CDMInterpolator.cc:115-124
// T -> float is Data::asFloat() = static_cast<float> per element (src/DataImpl.h:99,132,384-389; include/fimex/Utils.h:94-116),
// fused with mifi_bad2nanf on the fill value; four elements per lane, one vector load and one float4 store.
template <typename T>
using Vec4 = T __attribute__((ext_vector_type(4)));

template <typename T, bool VEC>
__global__ void __launch_bounds__(kBlock) to_float_kernel(const T* __restrict__ in, float* __restrict__ out, size_t n, float bad, bool hasBad)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    if (VEC) {
        const size_t n4 = n / 4;
        const Vec4<T>* in4 = reinterpret_cast<const Vec4<T>*>(in);
        float4* out4 = reinterpret_cast<float4*>(out);
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
            const Vec4<T> v = in4[i];
            out4[i] = make_float4(as_float_nan(v.x, bad, hasBad), as_float_nan(v.y, bad, hasBad), as_float_nan(v.z, bad, hasBad),
                                  as_float_nan(v.w, bad, hasBad));
        }
        for (size_t j = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) out[j] = as_float_nan(in[j], bad, hasBad);
    } else {
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = as_float_nan(in[i], bad, hasBad);
    }
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(kBlock) from_float_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n, T fill)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    if (VEC) {
        const size_t n4 = n / 4;
        const float4* in4 = reinterpret_cast<const float4*>(in);
        Vec4<T>* out4 = reinterpret_cast<Vec4<T>*>(out);
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
            const float4 v = in4[i];
            Vec4<T> r;
            r.x = from_float_fill<T>(v.x, fill);
            r.y = from_float_fill<T>(v.y, fill);
            r.z = from_float_fill<T>(v.z, fill);
            r.w = from_float_fill<T>(v.w, fill);
            out4[i] = r;
        }
        for (size_t j = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) out[j] = from_float_fill<T>(in[j], fill);
    } else {
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = from_float_fill<T>(in[i], fill);
    }
}

uint32_t stream_blocks(size_t n)
{
    const size_t groups = ceil_div(n, (size_t)4 * kBlock);
    return (uint32_t)(groups < 256 * 8 ? (groups ? groups : 1) : 256 * 8);
}

template <typename T>
void launch_to_float_t(const void* d_in, size_t n, double badValue, float* d_out, hipStream_t stream)
{
    const T* in = static_cast<const T*>(d_in);
    const float bad = (float)badValue;  // the double fill value narrows to mifi_bad2nanf's float parameter (CDMInterpolator.cc:117)
    const bool hasBad = !(bad != bad);
    const bool vec = reinterpret_cast<uintptr_t>(in) % (4 * sizeof(T)) == 0 && reinterpret_cast<uintptr_t>(d_out) % 16 == 0;
    if (vec) to_float_kernel<T, true><<<stream_blocks(n), kBlock, 0, stream>>>(in, d_out, n, bad, hasBad);
    else to_float_kernel<T, false><<<stream_blocks(n), kBlock, 0, stream>>>(in, d_out, n, bad, hasBad);
    FA_HIP(hipGetLastError());
}

template <typename T>
void launch_from_float_t(const float* d_in, size_t n, double badValue, void* d_out, hipStream_t stream)
{
    T* out = static_cast<T*>(d_out);
    const T fill = static_cast<T>(badValue);  // ScaleValue's newFill_ (Utils.h:456)
    const bool vec = reinterpret_cast<uintptr_t>(out) % (4 * sizeof(T)) == 0 && reinterpret_cast<uintptr_t>(d_in) % 16 == 0;
    if (vec) from_float_kernel<T, true><<<stream_blocks(n), kBlock, 0, stream>>>(d_in, out, n, fill);
    else from_float_kernel<T, false><<<stream_blocks(n), kBlock, 0, stream>>>(d_in, out, n, fill);
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

// ---- 1-D blends between two fields (time / vertical interpolation), src/interpolation.c:1030-1156
namespace {

template <typename T>
__global__ void __launch_bounds__(kBlock) blend_kernel(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ out, size_t n, T f)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const T iA = A[i], iB = B[i];
        out[i] = iA + f * (iB - iA);  // :1046 / :1078
    }
}

__global__ void __launch_bounds__(kBlock) undefined_kernel(float* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = undefined_f();
}

template <typename T>
void copy_field(const T* src, T* dst, size_t n, hipStream_t stream)  // the reference's memcpy: no 0 * NaN side effects
{
    if (src != dst) FA_HIP(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToDevice, stream));
}

template <typename T>
void blend(const T* A, const T* B, T* out, size_t n, T f, hipStream_t stream)
{
    blend_kernel<T><<<stream_blocks(n), kBlock, 0, stream>>>(A, B, out, n, f);
    FA_HIP(hipGetLastError());
}

// mifi_get_values_linear_f, :1050-1063
void linear_f(const float* A, const float* B, float* out, size_t n, double a, double b, double x, hipStream_t stream)
{
    const float f = (a == b) ? 0 : ((x - a) / (b - a));
    if (f == 0) copy_field(A, out, n, stream);
    else if (f == 1) copy_field(B, out, n, stream);
    else blend(A, B, out, n, f, stream);
}

// mifi_get_values_linear_conf_extrapol_f, :1085-1104
void linear_conf_extrapol_f(float left, float right, const float* A, const float* B, float* out, size_t n, double a, double b, double x,
                            hipStream_t stream)
{
    const float f = (a == b) ? 0 : ((x - a) / (b - a));
    if (f == 0) copy_field(A, out, n, stream);
    else if (f == 1) copy_field(B, out, n, stream);
    else if ((f >= left) && (f <= right)) blend(A, B, out, n, f, stream);
    else {
        undefined_kernel<<<stream_blocks(n), kBlock, 0, stream>>>(out, n);
        FA_HIP(hipGetLastError());
    }
}

}  // namespace

// returns false where the reference returns MIFI_ERROR (non-positive coordinates of the log blends)
bool launch_get_values_1d_f(int kind, const float* A, const float* B, float* out, size_t n, double a, double b, double x, hipStream_t stream)
{
    switch (kind) {
    case FIMEX_AMD_1D_NEAREST: if (n) copy_field(A, out, n, stream); return true;  // :1030-1034
    case FIMEX_AMD_1D_LINEAR: if (n) linear_f(A, B, out, n, a, b, x, stream); return true;
    case FIMEX_AMD_1D_LINEAR_WEAK_EXTRAPOL: if (n) linear_conf_extrapol_f(-1.f, 2.f, A, B, out, n, a, b, x, stream); return true;  // :1106-1109
    case FIMEX_AMD_1D_LINEAR_NO_EXTRAPOL: if (n) linear_conf_extrapol_f(0.f, 1.f, A, B, out, n, a, b, x, stream); return true;     // :1110-1113
    case FIMEX_AMD_1D_LINEAR_CONST_EXTRAPOL: {  // :1115-1126
        const float f = (a == b) ? 0 : ((x - a) / (b - a));
        if (n == 0) return true;
        if (f >= 1) copy_field(B, out, n, stream);
        else if (f <= 0) copy_field(A, out, n, stream);
        else blend(A, B, out, n, f, stream);
        return true;
    }
    case FIMEX_AMD_1D_LOG:  // :1134-1145; the three logarithms are taken on the host, by the same libm as the reference's
        if (a <= 0 || b <= 0 || x <= 0) return false;
        if (n) linear_f(A, B, out, n, std::log(a), std::log(b), std::log(x), stream);
        return true;
    case FIMEX_AMD_1D_LOG_LOG: {  // :1147-1156
        if (a <= 0 || b <= 0 || x <= 0) return false;
        const double la = std::log(a + M_E), lb = std::log(b + M_E), lx = std::log(x + M_E);
        if (!(la <= 0 || lb <= 0 || lx <= 0) && n) linear_f(A, B, out, n, std::log(la), std::log(lb), std::log(lx), stream);
        return true;  // the reference drops the inner status
    }
    default: throw Error("unknown 1-D blend " + std::to_string(kind));
    }
}

// mifi_get_values_linear_d, :1065-1083
void launch_get_values_linear_d(const double* A, const double* B, double* out, size_t n, double a, double b, double x, hipStream_t stream)
{
    if (n == 0) return;
    const double f = (a == b) ? 0 : ((x - a) / (b - a));
    if (f == 0) copy_field(A, out, n, stream);
    else if (f == 1) copy_field(B, out, n, stream);
    else blend(A, B, out, n, f, stream);
}

size_t cdm_type_size(int cdmType)
{
    switch (cdmType) {
    case FIMEX_AMD_CDM_CHAR: case FIMEX_AMD_CDM_UCHAR: return 1;
    case FIMEX_AMD_CDM_SHORT: case FIMEX_AMD_CDM_USHORT: return 2;
    case FIMEX_AMD_CDM_INT: case FIMEX_AMD_CDM_UINT: case FIMEX_AMD_CDM_FLOAT: return 4;
    case FIMEX_AMD_CDM_DOUBLE: case FIMEX_AMD_CDM_INT64: case FIMEX_AMD_CDM_UINT64: return 8;
    default: throw Error("data type " + std::to_string(cdmType) + " cannot be regridded (CDM_STRING / CDM_NAT have no float form)");
    }
}

void launch_data2interpolation(const void* d_in, int cdmType, size_t n, double badValue, float* d_out, hipStream_t stream)
{
    if (n == 0) return;
    switch (cdmType) {
    case FIMEX_AMD_CDM_CHAR: launch_to_float_t<signed char>(d_in, n, badValue, d_out, stream); break;  // char is signed on the reference's platforms
    case FIMEX_AMD_CDM_SHORT: launch_to_float_t<short>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_INT: launch_to_float_t<int>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_FLOAT: launch_to_float_t<float>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_DOUBLE: launch_to_float_t<double>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_UCHAR: launch_to_float_t<unsigned char>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_USHORT: launch_to_float_t<unsigned short>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_UINT: launch_to_float_t<unsigned int>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_INT64: launch_to_float_t<long long>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_UINT64: launch_to_float_t<unsigned long long>(d_in, n, badValue, d_out, stream); break;
    default: (void)cdm_type_size(cdmType);
    }
}

void launch_interpolation2data(const float* d_in, size_t n, int cdmType, double badValue, void* d_out, hipStream_t stream)
{
    if (n == 0) return;
    switch (cdmType) {
    case FIMEX_AMD_CDM_CHAR: launch_from_float_t<signed char>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_SHORT: launch_from_float_t<short>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_INT: launch_from_float_t<int>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_FLOAT: launch_from_float_t<float>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_DOUBLE: launch_from_float_t<double>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_UCHAR: launch_from_float_t<unsigned char>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_USHORT: launch_from_float_t<unsigned short>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_UINT: launch_from_float_t<unsigned int>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_INT64: launch_from_float_t<long long>(d_in, n, badValue, d_out, stream); break;
    case FIMEX_AMD_CDM_UINT64: launch_from_float_t<unsigned long long>(d_in, n, badValue, d_out, stream); break;
    default: (void)cdm_type_size(cdmType);
    }
}

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
