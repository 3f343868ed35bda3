// Per-cell 2x2 rotation of vector components and rotation of direction angles for gfx950.
//
// Replaces mifi_vector_reproject_values_by_matrix_f (src/interpolation.c:790-812) and
// mifi_vector_reproject_direction_by_matrix_f (:814-835) as called by CachedVectorReprojection
// (src/CachedVectorReprojection.cc:35-55).
//
// Pure streaming kernels: each lane owns 4 consecutive cells, keeps their (cos, sin) pairs -- read
// once from a compact double2 array instead of 16 of every 32 bytes of the reference's 4-double
// records -- in registers, and walks the z slices with 16-byte loads and stores of u and v.
// Arithmetic: float -> double products and one double add/sub, rounded once to float on store, as
// the reference does; -ffp-contract=off keeps the products un-fused.
#include "plan.hpp"

#include <vector>

namespace fimex_amd {

namespace {

constexpr double kRadToDeg = 57.29577951308232;  // PROJ.4 RAD_TO_DEG, used by interpolation.c:827

struct VecArgs {
    float* u;
    float* v;
    const double2* cs;
    const double* phi;
    uint32_t layer;      // cells per slice
    uint32_t nz;
    uint32_t zPerBlock;
};

__device__ __forceinline__ void rotate(float& u, float& v, double c, double s)
{
    const double un = (double)u * c - (double)v * s;  // :804
    const double vn = (double)u * s + (double)v * c;  // :805
    u = (float)un;
    v = (float)vn;
}

// layer % 4 == 0 and 16-byte aligned bases: float4 path
__global__ void __launch_bounds__(kBlock) rotate_values_vec4(VecArgs a)
{
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;  // group of 4 cells
    if (q * 4 >= a.layer) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    double2 cs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cs[k] = a.cs[q * 4 + k];
    float4* up = reinterpret_cast<float4*>(a.u + (size_t)z0 * a.layer) + q;
    float4* vp = reinterpret_cast<float4*>(a.v + (size_t)z0 * a.layer) + q;
    const size_t stride = a.layer / 4;
    uint32_t z = z0;
    for (; z + 2 <= z1; z += 2) {  // two slices in flight
        float4 u0 = up[0], v0 = vp[0], u1 = up[stride], v1 = vp[stride];
        rotate(u0.x, v0.x, cs[0].x, cs[0].y); rotate(u0.y, v0.y, cs[1].x, cs[1].y);
        rotate(u0.z, v0.z, cs[2].x, cs[2].y); rotate(u0.w, v0.w, cs[3].x, cs[3].y);
        rotate(u1.x, v1.x, cs[0].x, cs[0].y); rotate(u1.y, v1.y, cs[1].x, cs[1].y);
        rotate(u1.z, v1.z, cs[2].x, cs[2].y); rotate(u1.w, v1.w, cs[3].x, cs[3].y);
        up[0] = u0; vp[0] = v0; up[stride] = u1; vp[stride] = v1;
        up += 2 * stride;
        vp += 2 * stride;
    }
    for (; z < z1; ++z, up += stride, vp += stride) {
        float4 u0 = up[0], v0 = vp[0];
        rotate(u0.x, v0.x, cs[0].x, cs[0].y); rotate(u0.y, v0.y, cs[1].x, cs[1].y);
        rotate(u0.z, v0.z, cs[2].x, cs[2].y); rotate(u0.w, v0.w, cs[3].x, cs[3].y);
        up[0] = u0; vp[0] = v0;
    }
}

__global__ void __launch_bounds__(kBlock) rotate_values_scalar(VecArgs a)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.layer) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const double2 cs = a.cs[i];
    for (uint32_t z = z0; z < z1; ++z) {
        float u = a.u[(size_t)z * a.layer + i], v = a.v[(size_t)z * a.layer + i];
        rotate(u, v, cs.x, cs.y);
        a.u[(size_t)z * a.layer + i] = u;
        a.v[(size_t)z * a.layer + i] = v;
    }
}

__device__ __forceinline__ float rotate_angle(float ang, double phiDeg)
{
    double an = (double)ang - phiDeg;  // :827 (phiDeg = RAD_TO_DEG * m[3], same product as the reference)
    if (an < 0) an += 360;             // :829
    if (an > 360) an -= 360;           // :830
    return (float)an;
}

__global__ void __launch_bounds__(kBlock) rotate_direction(VecArgs a)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.layer) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const double phiDeg = kRadToDeg * a.phi[i];
    float* p = a.u + (size_t)z0 * a.layer + i;
    uint32_t z = z0;
    for (; z + 4 <= z1; z += 4, p += (size_t)4 * a.layer) {  // four slices in flight
        const float a0 = p[0], a1 = p[a.layer], a2 = p[(size_t)2 * a.layer], a3 = p[(size_t)3 * a.layer];
        p[0] = rotate_angle(a0, phiDeg);
        p[a.layer] = rotate_angle(a1, phiDeg);
        p[(size_t)2 * a.layer] = rotate_angle(a2, phiDeg);
        p[(size_t)3 * a.layer] = rotate_angle(a3, phiDeg);
    }
    for (; z < z1; ++z, p += a.layer) *p = rotate_angle(*p, phiDeg);
}

// layer % 4 == 0 and a 16-byte aligned base: four cells per lane
__global__ void __launch_bounds__(kBlock) rotate_direction_vec4(VecArgs a)
{
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q * 4 >= a.layer) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    double pd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pd[k] = kRadToDeg * a.phi[q * 4 + k];
    float4* p = reinterpret_cast<float4*>(a.u + (size_t)z0 * a.layer) + q;
    const size_t stride = a.layer / 4;
    uint32_t z = z0;
    for (; z + 2 <= z1; z += 2, p += 2 * stride) {
        float4 v0 = p[0], v1 = p[stride];
        v0.x = rotate_angle(v0.x, pd[0]); v0.y = rotate_angle(v0.y, pd[1]); v0.z = rotate_angle(v0.z, pd[2]); v0.w = rotate_angle(v0.w, pd[3]);
        v1.x = rotate_angle(v1.x, pd[0]); v1.y = rotate_angle(v1.y, pd[1]); v1.z = rotate_angle(v1.z, pd[2]); v1.w = rotate_angle(v1.w, pd[3]);
        p[0] = v0;
        p[stride] = v1;
    }
    for (; z < z1; ++z, p += stride) {
        float4 v0 = p[0];
        v0.x = rotate_angle(v0.x, pd[0]); v0.y = rotate_angle(v0.y, pd[1]); v0.z = rotate_angle(v0.z, pd[2]); v0.w = rotate_angle(v0.w, pd[3]);
        p[0] = v0;
    }
}

// CDMProcessor's direction rotation (src/CDMProcessor.cc:621-636): packed angles are unpacked with ScaleOffset<float>
// (scale * a + offset in double, stored to the float array), rotated, and packed again with UnScaleOffset<float>
// ((1 / scale) * (a - offset)) -- three passes there, one here
__global__ void __launch_bounds__(kBlock) rotate_direction_scaled(VecArgs a, double scale, double invscale, double offset)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.layer) return;
    const uint32_t z0 = blockIdx.y * a.zPerBlock;
    const uint32_t z1 = min(a.nz, z0 + a.zPerBlock);
    const double phiDeg = kRadToDeg * a.phi[i];
    float* p = a.u + (size_t)z0 * a.layer + i;
    for (uint32_t z = z0; z < z1; ++z, p += a.layer) {
        const float unpacked = (float)(scale * (double)*p + offset);   // :631
        const float rotated = rotate_angle(unpacked, phiDeg);          // :632
        *p = (float)(invscale * ((double)rotated - offset));           // :633
    }
}

VecArgs make_args(const fimex_amd_vector_plan& plan, float* u, float* v, size_t oz, uint32_t cellsPerLane, dim3& grid)
{
    VecArgs a{};
    a.u = u;
    a.v = v;
    a.cs = plan.cossin.get();
    a.phi = plan.phi.get();
    a.layer = (uint32_t)(plan.ox * plan.oy);
    a.nz = (uint32_t)oz;
    const size_t blocksX = ceil_div(ceil_div((size_t)a.layer, cellsPerLane), kBlock);
    // enough workgroups to fill the chip, long enough z runs to amortise the matrix read
    size_t chunks = ceil_div((size_t)256 * 8 * 2, blocksX);
    if (chunks > oz) chunks = oz;
    if (chunks < 1) chunks = 1;
    a.zPerBlock = (uint32_t)ceil_div(oz, chunks);
    chunks = ceil_div(oz, (size_t)a.zPerBlock);
    FA_REQUIRE(chunks <= 65535, "too many z chunks for one launch");
    grid = dim3((uint32_t)blocksX, (uint32_t)chunks, 1);
    return a;
}

}  // namespace

void build_vector_plan(fimex_amd_vector_plan& plan, const double* h_matrix)
{
    const size_t n = plan.ox * plan.oy;
    FA_REQUIRE(n > 0 && n <= 0xFFFFFFF0u, "rotation grid must have between 1 and 2^32-16 cells");
    std::vector<double2> cs(n);
    std::vector<double> phi(n);
    for (size_t i = 0; i < n; ++i) {
        cs[i] = make_double2(h_matrix[4 * i], h_matrix[4 * i + 1]);
        phi[i] = h_matrix[4 * i + 3];
    }
    plan.cossin.allocate(n);
    plan.phi.allocate(n);
    FA_HIP(hipMemcpy(plan.cossin.get(), cs.data(), n * sizeof(double2), hipMemcpyHostToDevice));
    FA_HIP(hipMemcpy(plan.phi.get(), phi.data(), n * sizeof(double), hipMemcpyHostToDevice));
}

void launch_vector_values(const fimex_amd_vector_plan& plan, float* d_u, float* d_v, size_t oz, hipStream_t stream)
{
    if (oz == 0) return;
    FA_REQUIRE(oz <= 0xFFFFFFFFu, "too many slices");
    const size_t layer = plan.ox * plan.oy;
    const bool vec4 = (layer % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_u) | reinterpret_cast<uintptr_t>(d_v)) % 16 == 0);
    dim3 grid;
    const VecArgs a = make_args(plan, d_u, d_v, oz, vec4 ? 4 : 1, grid);
    if (vec4) rotate_values_vec4<<<grid, kBlock, 0, stream>>>(a);
    else rotate_values_scalar<<<grid, kBlock, 0, stream>>>(a);
    FA_HIP(hipGetLastError());
}

void launch_vector_direction(const fimex_amd_vector_plan& plan, float* d_angles, size_t oz, hipStream_t stream)
{
    if (oz == 0) return;
    FA_REQUIRE(oz <= 0xFFFFFFFFu, "too many slices");
    const size_t layer = plan.ox * plan.oy;
    const bool vec4 = (layer % 4 == 0) && (reinterpret_cast<uintptr_t>(d_angles) % 16 == 0);
    dim3 grid;
    const VecArgs a = make_args(plan, d_angles, nullptr, oz, vec4 ? 4 : 1, grid);
    if (vec4) rotate_direction_vec4<<<grid, kBlock, 0, stream>>>(a);
    else rotate_direction<<<grid, kBlock, 0, stream>>>(a);
    FA_HIP(hipGetLastError());
}

void launch_vector_direction_scaled(const fimex_amd_vector_plan& plan, float* d_angles, size_t oz, double scale, double offset, hipStream_t stream)
{
    if (oz == 0) return;
    FA_REQUIRE(oz <= 0xFFFFFFFFu, "too many slices");
    dim3 grid;
    const VecArgs a = make_args(plan, d_angles, nullptr, oz, 1, grid);
    rotate_direction_scaled<<<grid, kBlock, 0, stream>>>(a, scale, 1 / scale, offset);  // UnScaleOffset's invscale_ = 1 / scale (:464)
    FA_HIP(hipGetLastError());
}

}  // namespace fimex_amd
