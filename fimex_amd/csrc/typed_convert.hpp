// Element conversions at the edges of a slice, shared by the conversion passes (convert.hip) and the regrid kernels that
// read and write a variable's stored type directly (regrid.hip): data2InterpolationArray / interpolationArray2Data,
// src/CDMInterpolator.cc:115-124.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace fimex_amd {

__device__ __forceinline__ float undefined_value_f() { return __uint_as_float(0x7fc00000u); }

// T -> float is Data::asFloat() = static_cast<float> per element (src/DataImpl.h:99,132,384-389; include/fimex/Utils.h:94-116),
// then mifi_bad2nanf with the fill value narrowed to float (src/interpolation.c:1775-1783; a NaN fill value changes nothing)
template <typename T>
__device__ __forceinline__ float as_float_nan(T v, float bad, bool hasBad)
{
    const float f = (float)v;
    return (hasBad && f == bad) ? undefined_value_f() : f;
}

// MetNoFimex::round(double) (include/fimex/Utils.h:72-75): lround, then long -> int.  Outside the range of long the
// reference is unspecified; LONG_MIN (what glibc/x86-64 yields) is kept (DESIGN.md divergence D6).
__device__ __forceinline__ int mifi_round(double num)
{
    const long long r = (fabs(num) < 9223372036854775808.0) ? llround(num) : (-9223372036854775807LL - 1);
    return (int)r;
}

// float -> T is ScaleValue<float, T>(NaN, 1, 0, fill, 1, 0) (include/fimex/Utils.h:444-464): NaN -> fill, else
// data_caster<T, double>(1.0 * v + 0.0): through mifi_round for integer T, a plain cast otherwise
template <typename T>
__device__ __forceinline__ T from_float_fill(float v, T fill)
{
    if (isnan(v)) return fill;
    if (std::is_integral<T>::value && fabsf(v) < 2147483648.f) {
        // lround of a float inside the int range, without the detour through double: the fraction v - trunc(v) is exact
        // in float, and from 2^23 on v is an integer already; then int -> T as the reference's static_cast
        const float t = truncf(v);
        const float r = t + ((fabsf(v - t) >= 0.5f) ? copysignf(1.f, v) : 0.f);
        return (T)(int)r;
    }
    const double d = 1.0 * (double)v + 0.0;  // turns -0.0 into +0.0, as the reference does
    if (std::is_integral<T>::value) return (T)mifi_round(d);
    return (T)d;
}

}  // namespace fimex_amd
