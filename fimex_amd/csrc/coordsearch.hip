// Plan building for the coordinate-based nearest-neighbour methods:
//   MIFI_INTERPOL_COORD_NN     fastTranslatePointsToClosestInputCell + getGridDistance   src/CDMInterpolator.cc:1069-1217
//   MIFI_INTERPOL_COORD_NN_KD  flannTranslatePointsToClosestInputCell (nanoflann kd-tree)  src/CDMInterpolator.cc:991-1067
// Both find, for every target point (lon, lat in rad), the closest cell of a source grid that is described by 2-D
// longitude / latitude fields, within a radius; the result (ix, iy as doubles) is the plan of row a1 (nearest).
//
// The reference walks a latitude-sorted list (COORD_NN) or a kd-tree (COORD_NN_KD).  Here both share one search
// structure: the source cells as points on the unit sphere, binned into cubes of edge >= the search radius and sorted by
// cube (rocPRIM radix sort); a query inspects the 27 cubes around its own, three of which are one contiguous key range.
// What is computed per candidate is the reference's own expression, so that the chosen cell is the reference's:
//   COORD_NN     cos_d = cos(lat1) cos(lat0) cos(lon1 - lon0) + sin(lat1) sin(lat0)  > cos(ROI), maximal   (:1180, :1199)
//   COORD_NN_KD  d2 = dx dx + dy dy + dz dz of the unit vectors  < (maxDist / R)^2, minimal                 (:966-972, :1039)
// Exactly equal candidates are taken in the order of the reference's containers there (unspecified: unstable sort /
// tree order); here the lowest source index wins.  Source cells with NaN coordinates never match.
#include <cstring>  // rocprim's texture iterator needs memset declared first

#include "plan.hpp"

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <vector>

namespace fimex_amd {

namespace {

constexpr double kPi = 3.1415926535897932384626433832795;  // MIFI_PI
constexpr double kEarthRadius = 6371000.;                    // MIFI_EARTH_RADIUS_M, include/fimex/CDMconstants.h:113
constexpr uint64_t kNoKey = ~0ull;

struct Grid {
    uint32_t g;      // cubes per axis over [-1, 1]
    double inv;      // g / 2
};

__device__ __forceinline__ uint32_t cube_of(double v, const Grid& gr)
{
    const double c = floor((v + 1.0) * gr.inv);
    return (uint32_t)fmin(fmax(c, 0.0), (double)(gr.g - 1));
}
__device__ __forceinline__ uint64_t key_of(uint32_t cx, uint32_t cy, uint32_t cz, const Grid& gr)
{
    return ((uint64_t)cx * gr.g + cy) * gr.g + cz;
}

struct Point {  // one source cell, in sorted order
    double x, y, z;          // unit vector (:1012-1014)
    double lon, sinLat, cosLat;
    uint32_t index;          // ix + iy * orgXDimSize
    uint32_t pad;
};

__global__ void __launch_bounds__(kBlock) source_keys_kernel(const double* __restrict__ lon, const double* __restrict__ lat, size_t n, Grid gr,
                                                             uint64_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double lo = lon[i], la = lat[i];
        uint64_t k = kNoKey;
        if (!(isnan(lo) || isnan(la))) {  // :1007
            const double cosLat = cos(la);
            k = key_of(cube_of(cosLat * cos(lo), gr), cube_of(cosLat * sin(lo), gr), cube_of(sin(la), gr), gr);
        }
        keys[i] = k;
        idx[i] = (uint32_t)i;
    }
}

__global__ void __launch_bounds__(kBlock) source_points_kernel(const double* __restrict__ lon, const double* __restrict__ lat,
                                                               const uint32_t* __restrict__ sortedIdx, size_t n, Point* __restrict__ pts)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t k = sortedIdx[i];
        const double lo = lon[k], la = lat[k];
        Point p;
        p.sinLat = sin(la);   // :1008-1011
        p.cosLat = cos(la);
        const double sinLon = sin(lo), cosLon = cos(lo);
        p.x = p.cosLat * cosLon;
        p.y = p.cosLat * sinLon;
        p.z = p.sinLat;
        p.lon = lo;
        p.index = k;
        p.pad = 0;
        pts[i] = p;
    }
}

__device__ __forceinline__ size_t lower_bound_key(const uint64_t* __restrict__ keys, size_t n, uint64_t key)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// KD == true: squared chord distance below limit, minimal; KD == false: cos_d above limit, maximal
template <bool KD>
__global__ void __launch_bounds__(kBlock) nearest_kernel(double* __restrict__ qx, double* __restrict__ qy, size_t nq,
                                                         const uint64_t* __restrict__ keys, const Point* __restrict__ pts, size_t nValid,
                                                         Grid gr, double limit, uint32_t orgX, double noMatch)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nq; i += stride) {
        const double lon0 = qx[i], lat0 = qy[i];
        const double sinLat = sin(lat0), cosLat = cos(lat0);   // :1044-1047
        const double sinLon = sin(lon0), cosLon = cos(lon0);
        const double x = cosLat * cosLon, y = cosLat * sinLon, z = sinLat;
        double best = limit;
        uint32_t bestIdx = 0xFFFFFFFFu;
        if (!(isnan(lon0) || isnan(lat0))) {
            const int cx = (int)cube_of(x, gr), cy = (int)cube_of(y, gr), cz = (int)cube_of(z, gr);
            const int g = (int)gr.g;
            for (int dx = -1; dx <= 1; ++dx) {
                const int ux = cx + dx;
                if (ux < 0 || ux >= g) continue;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int uy = cy + dy;
                    if (uy < 0 || uy >= g) continue;
                    const uint64_t k0 = key_of((uint32_t)ux, (uint32_t)uy, (uint32_t)max(cz - 1, 0), gr);
                    const uint64_t k1 = key_of((uint32_t)ux, (uint32_t)uy, (uint32_t)min(cz + 1, g - 1), gr);
                    for (size_t j = lower_bound_key(keys, nValid, k0); j < nValid && keys[j] <= k1; ++j) {
                        const Point p = pts[j];
                        bool closer;
                        double m;
                        if (KD) {
                            const double d0 = x - p.x, d1 = y - p.y, d2 = z - p.z;   // kdtree_distance, :966-972
                            m = d0 * d0 + d1 * d1 + d2 * d2;
                            closer = m < best || (m == best && bestIdx != 0xFFFFFFFFu && p.index < bestIdx);  // addPoint: dist < radius
                        } else {
                            const double dlon = p.lon - lon0;                        // :1175
                            m = p.cosLat * cosLat * cos(dlon) + p.sinLat * sinLat;    // :1180
                            closer = m > best || (m == best && bestIdx != 0xFFFFFFFFu && p.index < bestIdx);  // :1181 cos_d > min_cos_d
                        }
                        if (closer) { best = m; bestIdx = p.index; }
                    }
                }
            }
        }
        if (bestIdx != 0xFFFFFFFFu) {
            qx[i] = (double)(bestIdx % orgX);   // :1054-1058
            qy[i] = (double)(bestIdx / orgX);
        } else {
            qx[i] = noMatch;                     // :1061-1062 (-1000) / LL_POINT's x = y = -1 (:1152)
            qy[i] = noMatch;
        }
    }
}

// getGridDistance, :1069-1141: for every sample cell the largest cos_d to any other cell
__global__ void __launch_bounds__(kBlock) grid_distance_kernel(const double* __restrict__ lon, const double* __restrict__ lat, size_t n,
                                                               size_t stepSize, double* __restrict__ maxCos)
{
    __shared__ double red[kBlock];
    const size_t samplePos = (size_t)blockIdx.x * stepSize;
    const double lon0 = lon[samplePos], lat0 = lat[samplePos];
    double best = -2;  // :1091
    if (!(isnan(lon0) || isnan(lat0))) {
        const double c0 = cos(lat0), s0 = sin(lat0);
        for (size_t pos = threadIdx.x; pos < n; pos += kBlock) {
            if (pos == samplePos) continue;
            const double lon1 = lon[pos], lat1 = lat[pos];
            if (isnan(lon1) || isnan(lat1)) continue;
            const double dlon = lon0 - lon1;
            const double cos_d = c0 * cos(lat1) * cos(dlon) + s0 * sin(lat1);  // :1103
            if (cos_d > best) best = cos_d;
        }
    }
    red[threadIdx.x] = best;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) maxCos[blockIdx.x] = (isnan(lon0) || isnan(lat0)) ? nan("") : red[0];
}

uint32_t blocks_for(size_t n)
{
    const size_t want = ceil_div(n, (size_t)kBlock);
    return (uint32_t)(want < 256 * 8 ? (want ? want : 1) : 256 * 8);
}

struct SearchIndex {
    Grid grid;
    DeviceArray<uint64_t> keys;
    DeviceArray<Point> points;
    size_t nValid = 0;
};

// chord: Euclidean search radius on the unit sphere the cubes have to cover
void build_index(SearchIndex& s, const double* d_lon, const double* d_lat, size_t n, double chord, hipStream_t stream)
{
    double g = std::floor(2.0 / chord);
    if (!(g >= 1)) g = 1;
    if (g > 1048576.0) g = 1048576.0;  // 60-bit keys
    s.grid.g = (uint32_t)g;
    s.grid.inv = g / 2.0;
    DeviceArray<uint64_t> keysIn(n);
    DeviceArray<uint32_t> idxIn(n), idxOut(n);
    s.keys.allocate(n);
    source_keys_kernel<<<blocks_for(n), kBlock, 0, stream>>>(d_lon, d_lat, n, s.grid, keysIn.get(), idxIn.get());
    FA_HIP(hipGetLastError());
    size_t tmpBytes = 0;
    FA_HIP(rocprim::radix_sort_pairs(nullptr, tmpBytes, keysIn.get(), s.keys.get(), idxIn.get(), idxOut.get(), n, 0, 64, stream));
    DeviceArray<unsigned char> tmp(tmpBytes ? tmpBytes : 1);
    FA_HIP(rocprim::radix_sort_pairs(tmp.get(), tmpBytes, keysIn.get(), s.keys.get(), idxIn.get(), idxOut.get(), n, 0, 64, stream));
    s.points.allocate(n);
    source_points_kernel<<<blocks_for(n), kBlock, 0, stream>>>(d_lon, d_lat, idxOut.get(), n, s.points.get());
    FA_HIP(hipGetLastError());
    // cells with NaN coordinates sort to the end (key ~0): find where they start
    std::vector<uint64_t> probe(1);
    size_t lo = 0, hi = n;
    while (lo < hi) {  // host-side binary search over the device keys, ~24 tiny copies
        const size_t mid = (lo + hi) / 2;
        FA_HIP(hipMemcpyAsync(probe.data(), s.keys.get() + mid, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        FA_HIP(hipStreamSynchronize(stream));
        if (probe[0] != kNoKey) lo = mid + 1;
        else hi = mid;
    }
    s.nValid = lo;
    FA_HIP(hipStreamSynchronize(stream));  // temporaries are released on return
}

}  // namespace

double grid_distance(const double* d_lon, const double* d_lat, size_t orgX, size_t orgY, hipStream_t stream)
{
    const size_t n = orgX * orgY;
    size_t steps, stepSize;
    if (n > 1000) { steps = 53; stepSize = n / steps; }  // :1077-1083
    else { stepSize = 1; steps = n; }
    DeviceArray<double> d_max(steps);
    grid_distance_kernel<<<(uint32_t)steps, kBlock, 0, stream>>>(d_lon, d_lat, n, stepSize, d_max.get());
    FA_HIP(hipGetLastError());
    std::vector<double> h(steps);
    FA_HIP(hipMemcpyAsync(h.data(), d_max.get(), steps * sizeof(double), hipMemcpyDeviceToHost, stream));
    FA_HIP(hipStreamSynchronize(stream));
    double minCos = 2;
    bool any = false;
    for (double v : h) if (!std::isnan(v)) { any = true; if (v < minCos) minCos = v; }  // min_element of the samples, :1136
    if (!any) throw Error("coord_nearestneighbor: every sampled source cell has undefined coordinates");
    double d = std::acos(minCos);
    d *= 1.414;             // :1137
    if (d > kPi) d = kPi;   // :1138
    return d;
}

// fastTranslatePointsToClosestInputCell, :1158-1217
void launch_coord_nearest(double* d_pointsX, double* d_pointsY, size_t nPoints, const double* d_lon, const double* d_lat, size_t orgX,
                          size_t orgY, hipStream_t stream)
{
    const size_t n = orgX * orgY;
    if (nPoints == 0) return;
    FA_REQUIRE(n > 0 && n <= 0xFFFFFFF0u, "coord_nearestneighbor: source grid must have between 1 and 2^32-16 cells");
    const double maxGridD = grid_distance(d_lon, d_lat, orgX, orgY, stream);
    const double minGridCosD = std::cos(maxGridD);  // :1165
    SearchIndex s;
    build_index(s, d_lon, d_lat, n, maxGridD, stream);  // chord <= arc
    nearest_kernel<false><<<blocks_for(nPoints), kBlock, 0, stream>>>(d_pointsX, d_pointsY, nPoints, s.keys.get(), s.points.get(), s.nValid, s.grid,
                                                                      minGridCosD, (uint32_t)orgX, -1.);
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(stream));
}

// flannTranslatePointsToClosestInputCell, :991-1067
void launch_coord_kdtree(double maxDist, double* d_pointsX, double* d_pointsY, size_t nPoints, const double* d_lon, const double* d_lat,
                         size_t orgX, size_t orgY, hipStream_t stream)
{
    const size_t n = orgX * orgY;
    if (nPoints == 0) return;
    FA_REQUIRE(n > 0 && n <= 0xFFFFFFF0u, "coord_kdtree: source grid must have between 1 and 2^32-16 cells");
    FA_REQUIRE(maxDist > 0, "coord_kdtree: the maximum distance must be positive (assert at :998)");
    const double r = maxDist / kEarthRadius;  // :1001
    SearchIndex s;
    build_index(s, d_lon, d_lat, n, r, stream);
    nearest_kernel<true><<<blocks_for(nPoints), kBlock, 0, stream>>>(d_pointsX, d_pointsY, nPoints, s.keys.get(), s.points.get(), s.nValid, s.grid,
                                                                     r * r, (uint32_t)orgX, -1000.);  // :1039 squared radius
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(stream));
}

}  // namespace fimex_amd
