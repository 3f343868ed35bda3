// Plan building on the device: the coordinate transforms the reference obtains from PROJ.4's
// pj_transform -- mifi_project_values / mifi_project_axes (src/interpolation.c:1158-1244) and the local rotation
// matrix mifi_get_vector_reproject_matrix (:719-788 -> :441-521 -> :330-438) -- for 4 M-cell target grids.
//
// PROJ.4 is a third-party library that is not part of the reference tree; the projections are implemented from their
// published closed forms on the sphere (Snyder, "Map Projections - A Working Manual", USGS PP 1395) with PROJ.4's
// conventions at the pj_transform boundary: geographic coordinates in radians, projected x = a * x' + x_0, longitudes
// relative to lon_0 wrapped to [-pi, pi].  Supported: latlong/longlat, stere, lcc, merc, ob_tran + o_proj=longlat.
// Ellipsoids are not implemented (spherical strings only); anything else fails loudly.
#include "plan.hpp"

#include <cmath>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace fimex_amd {

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kHalfPi = kPi / 2, kFortPi = kPi / 4;
constexpr double kSpi = 3.14159265359;  // PROJ.4's adjlon threshold
constexpr double kEps10 = 1e-10;
constexpr double kDegToRad = .0174532925199432958;  // proj_api.h DEG_TO_RAD

enum ProjKind { kLatLong = 0, kStere, kLcc, kMerc, kObTran };
enum StereMode { kNorth = 0, kSouth, kOblique, kEquatorial };

struct ProjParams {
    int kind, mode, oblique;
    double a, lam0, phi0, x0, y0, k0;
    double akm1, sinph0, cosph0;  // stere
    double n, c, rho0;            // lcc
    double lamp, sphip, cphip;    // ob_tran
};

bool geographic_name(const std::string& n) { return n == "latlong" || n == "longlat" || n == "latlon" || n == "lonlat"; }

ProjParams parse_proj4(const char* text)
{
    FA_REQUIRE(text != nullptr, "NULL projection string");
    const std::string proj4(text);
    std::map<std::string, std::string> par;
    std::istringstream in(proj4);
    std::string tok;
    while (in >> tok) {
        while (!tok.empty() && tok[0] == '+') tok.erase(0, 1);
        if (tok.empty()) continue;
        const size_t eq = tok.find('=');
        if (eq == std::string::npos) par[tok] = "";
        else par[tok.substr(0, eq)] = tok.substr(eq + 1);
    }
    auto has = [&](const char* k) { return par.count(k) != 0; };
    auto num = [&](const char* k, double d) {
        auto it = par.find(k);
        if (it == par.end()) return d;
        try { return std::stod(it->second); } catch (...) { throw Error("projection parameter +" + std::string(k) + " is not a number: " + proj4); }
    };
    auto rad = [&](const char* k, double d) { return has(k) ? num(k, 0) * kPi / 180.0 : d; };
    if (!has("proj")) throw Error("projection string without +proj: " + proj4);
    const std::string name = par["proj"];
    ProjParams p{};
    p.a = 1;
    if (has("R")) p.a = num("R", 1);
    else if (has("a")) {
        const bool sphere = (!has("e") || num("e", 0) == 0.0) && !has("b") && !has("rf") && !has("f") && (!has("ellps") || par["ellps"] == "sphere");
        if (!sphere) throw Error("ellipsoidal projections are not implemented: " + proj4);
        p.a = num("a", 1);
    } else if (has("ellps") && par["ellps"] == "sphere") p.a = 6370997.0;
    else if (!geographic_name(name)) throw Error("ellipsoidal projections are not implemented: " + proj4);
    p.lam0 = rad("lon_0", 0);
    p.phi0 = rad("lat_0", 0);
    p.x0 = num("x_0", 0);
    p.y0 = num("y_0", 0);
    p.k0 = has("k_0") ? num("k_0", 1) : num("k", 1);
    if (geographic_name(name)) {
        p.kind = kLatLong;
    } else if (name == "stere") {
        p.kind = kStere;
        const double phits = has("lat_ts") ? std::fabs(rad("lat_ts", kHalfPi)) : kHalfPi;
        const double t = std::fabs(p.phi0);
        if (std::fabs(t - kHalfPi) < kEps10) p.mode = p.phi0 < 0 ? kSouth : kNorth;
        else p.mode = t > kEps10 ? kOblique : kEquatorial;
        if (p.mode == kNorth || p.mode == kSouth) {
            p.akm1 = (std::fabs(phits - kHalfPi) >= kEps10) ? std::cos(phits) / std::tan(kFortPi - .5 * phits) : 2. * p.k0;
        } else {
            p.sinph0 = std::sin(p.phi0);
            p.cosph0 = std::cos(p.phi0);
            p.akm1 = 2. * p.k0;
        }
    } else if (name == "lcc") {
        p.kind = kLcc;
        const double phi1 = rad("lat_1", 0);
        const double phi2 = has("lat_2") ? rad("lat_2", phi1) : phi1;
        if (!has("lat_0")) p.phi0 = phi1;
        const double cosphi = std::cos(phi1);
        p.n = std::sin(phi1);
        if (std::fabs(phi1 - phi2) >= kEps10)
            p.n = std::log(cosphi / std::cos(phi2)) / std::log(std::tan(kFortPi + .5 * phi2) / std::tan(kFortPi + .5 * phi1));
        p.c = cosphi * std::pow(std::tan(kFortPi + .5 * phi1), p.n) / p.n;
        p.rho0 = (std::fabs(std::fabs(p.phi0) - kHalfPi) < kEps10) ? 0. : p.c * std::pow(std::tan(kFortPi + .5 * p.phi0), -p.n);
    } else if (name == "merc") {
        p.kind = kMerc;
        if (has("lat_ts")) p.k0 = std::cos(std::fabs(rad("lat_ts", 0)));
    } else if (name == "ob_tran") {
        p.kind = kObTran;
        if (!has("o_proj") || !geographic_name(par["o_proj"]) || !has("o_lat_p"))
            throw Error("ob_tran is implemented for +o_proj=longlat +o_lat_p only: " + proj4);
        p.lamp = rad("o_lon_p", 0);
        const double phip = rad("o_lat_p", kHalfPi);
        p.oblique = std::fabs(phip - kHalfPi) > kEps10;
        p.sphip = std::sin(phip);
        p.cphip = std::cos(phip);
    } else {
        throw Error("projection not implemented: " + name);
    }
    return p;
}

__device__ __forceinline__ double adjlon(double lon)
{
    if (fabs(lon) <= kSpi) return lon;
    lon += kPi;
    lon -= 2 * kPi * floor(lon / (2 * kPi));
    return lon - kPi;
}

// geographic (rad) -> projected
__device__ void proj_forward(const ProjParams& p, double lon, double lat, double& x, double& y)
{
    if (p.kind == kLatLong) { x = lon; y = lat; return; }
    double lam = adjlon(lon - p.lam0), phi = lat;
    double px = 0, py = 0;
    if (p.kind == kStere) {
        const double sinlam = sin(lam);
        double coslam = cos(lam);
        if (p.mode == kNorth || p.mode == kSouth) {
            if (p.mode == kNorth) { coslam = -coslam; phi = -phi; }
            py = p.akm1 * tan(kFortPi + .5 * phi);
            px = sinlam * py;
            py *= coslam;
        } else {
            const double sinphi = sin(phi), cosphi = cos(phi);
            if (p.mode == kEquatorial) {
                const double k = p.akm1 / (1. + cosphi * coslam);
                px = k * cosphi * sinlam;
                py = k * sinphi;
            } else {
                const double k = p.akm1 / (1. + p.sinph0 * sinphi + p.cosph0 * cosphi * coslam);
                px = k * cosphi * sinlam;
                py = k * (p.cosph0 * sinphi - p.sinph0 * cosphi * coslam);
            }
        }
    } else if (p.kind == kLcc) {
        const double rho = (fabs(fabs(phi) - kHalfPi) < kEps10) ? 0. : p.c * pow(tan(kFortPi + .5 * phi), -p.n);
        lam *= p.n;
        px = p.k0 * (rho * sin(lam));
        py = p.k0 * (p.rho0 - rho * cos(lam));
    } else if (p.kind == kMerc) {
        px = p.k0 * lam;
        py = p.k0 * log(tan(kFortPi + .5 * phi));
    } else {  // ob_tran + longlat: radians stay radians
        if (p.oblique) {
            const double coslam = cos(lam), sinphi = sin(phi), cosphi = cos(phi);
            px = adjlon(atan2(cosphi * sin(lam), p.sphip * cosphi * coslam + p.cphip * sinphi) + p.lamp);
            double s = p.sphip * sinphi - p.cphip * cosphi * coslam;
            s = s > 1 ? 1 : (s < -1 ? -1 : s);
            py = asin(s);
        } else {
            px = adjlon(lam + p.lamp);
            py = phi;
        }
        x = px + p.x0;
        y = py + p.y0;
        return;
    }
    x = p.a * px + p.x0;
    y = p.a * py + p.y0;
}

// projected -> geographic (rad)
__device__ void proj_inverse(const ProjParams& p, double x, double y, double& lon, double& lat)
{
    if (p.kind == kLatLong) { lon = x; lat = y; return; }
    double xs, ys;
    if (p.kind == kObTran) { xs = x - p.x0; ys = y - p.y0; }
    else { xs = (x - p.x0) / p.a; ys = (y - p.y0) / p.a; }
    double lam = 0, phi = 0;
    if (p.kind == kStere) {
        const double rh = hypot(xs, ys);
        const double c = 2. * atan(rh / p.akm1);
        const double sinc = sin(c), cosc = cos(c);
        if (p.mode == kNorth) {
            ys = -ys;
            phi = (fabs(rh) <= kEps10) ? p.phi0 : asin(cosc);
            lam = (xs == 0. && ys == 0.) ? 0. : atan2(xs, ys);
        } else if (p.mode == kSouth) {
            phi = (fabs(rh) <= kEps10) ? p.phi0 : asin(-cosc);
            lam = (xs == 0. && ys == 0.) ? 0. : atan2(xs, ys);
        } else if (p.mode == kEquatorial) {
            phi = (fabs(rh) <= kEps10) ? 0. : asin(ys * sinc / rh);
            lam = (cosc != 0. || xs != 0.) ? atan2(xs * sinc, cosc * rh) : 0.;
        } else {
            phi = (fabs(rh) <= kEps10) ? p.phi0 : asin(cosc * p.sinph0 + ys * sinc * p.cosph0 / rh);
            const double cc = cosc - p.sinph0 * sin(phi);
            lam = (cc != 0. || xs != 0.) ? atan2(xs * sinc * p.cosph0, cc * rh) : 0.;
        }
    } else if (p.kind == kLcc) {
        xs /= p.k0;
        ys = p.rho0 - ys / p.k0;
        double rho = hypot(xs, ys);
        if (rho != 0.) {
            if (p.n < 0.) { rho = -rho; xs = -xs; ys = -ys; }
            phi = 2. * atan(pow(p.c / rho, 1. / p.n)) - kHalfPi;
            lam = atan2(xs, ys) / p.n;
        } else {
            lam = 0.;
            phi = p.n > 0. ? kHalfPi : -kHalfPi;
        }
    } else if (p.kind == kMerc) {
        lam = xs / p.k0;
        phi = kHalfPi - 2. * atan(exp(-ys / p.k0));
    } else {
        if (p.oblique) {
            const double lamr = xs - p.lamp;
            const double coslam = cos(lamr), sinphi = sin(ys), cosphi = cos(ys);
            double s = p.sphip * sinphi + p.cphip * cosphi * coslam;
            s = s > 1 ? 1 : (s < -1 ? -1 : s);
            phi = asin(s);
            lam = atan2(cosphi * sin(lamr), p.sphip * cosphi * coslam - p.cphip * sinphi);
        } else {
            lam = xs - p.lamp;
            phi = ys;
        }
    }
    lon = adjlon(lam + p.lam0);
    lat = phi;
}

// pj_transform(src, dst) on one point (no datum shift: both sides are spheres)
__device__ __forceinline__ void transform_point(const ProjParams& src, const ProjParams& dst, double& x, double& y)
{
    double lon, lat;
    proj_inverse(src, x, y, lon, lat);
    proj_forward(dst, lon, lat, x, y);
}

__global__ void __launch_bounds__(kBlock) project_values_kernel(ProjParams src, ProjParams dst, double* __restrict__ x, double* __restrict__ y, size_t n)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double px = x[i], py = y[i];
        transform_point(src, dst, px, py);
        x[i] = px;
        y[i] = py;
    }
}

// the [iy][ix] mesh of two axes, transformed (interpolation.c:1226-1234)
__global__ void __launch_bounds__(kBlock) project_axes_kernel(ProjParams src, ProjParams dst, const double* __restrict__ xAxis,
                                                              const double* __restrict__ yAxis, uint32_t ix, uint32_t iy,
                                                              double* __restrict__ outX, double* __restrict__ outY)
{
    const size_t n = (size_t)ix * iy, stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double px = xAxis[i % ix], py = yAxis[i / ix];
        transform_point(src, dst, px, py);
        outX[i] = px;
        outY[i] = py;
    }
}

// interpolation.c:311-329
__device__ __forceinline__ double bearing(double lat0, double lon0, double lat1, double lon1)
{
    const double dlon = lon0 - lon1;
    return atan2(sin(dlon) * cos(lat1), cos(lat0) * sin(lat1) - sin(lat0) * cos(lat1) * cos(dlon));
}

// mifi_get_vector_reproject_matrix_points_proj_delta (interpolation.c:330-438): per point of the output mesh the angle of
// the input projection's x direction (and y direction) seen in the output projection, from two finite differences
// outX / outY: the two axes of the output mesh (axes != 0) or one value per point
__global__ void __launch_bounds__(kBlock) vector_matrix_kernel(ProjParams in, ProjParams out, const double* __restrict__ inX,
                                                               const double* __restrict__ inY, const double* __restrict__ outX,
                                                               const double* __restrict__ outY, int axes, uint32_t ox, size_t n,
                                                               double deltaX, double deltaY, int outIsLatLong,
                                                               double* __restrict__ matrix)
{
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double outXf = axes ? outX[i % ox] : outX[i], outYf = axes ? outY[i / ox] : outY[i];
        double ax = inX[i] + deltaX, ay = inY[i];  // (x + d, y), :343-355
        double bx = inX[i], by = inY[i] + deltaY;  // (x, y + d), :384-396
        transform_point(in, out, ax, ay);
        transform_point(in, out, bx, by);
        double phi0;
        if (outIsLatLong) {
            phi0 = bearing(outYf, outXf, by, bx);  // :408-412
            if (deltaY < 0) phi0 += kPi;
        } else {
            double phiy = atan2(ay - outYf, ax - outXf);  // :372-373
            if (deltaX < 0) phiy += kPi;
            double phix = -1 * atan2(bx - outXf, by - outYf);  // :414-415
            if (deltaY < 0) phix += kPi;
            phi0 = .5 * (phix + phiy);  // :424
        }
        const double c = cos(phi0), s = sin(phi0);
        matrix[4 * i + 0] = c;       // :429-432
        matrix[4 * i + 1] = s;
        matrix[4 * i + 2] = -1 * s;
        matrix[4 * i + 3] = phi0;
    }
}

uint32_t point_blocks(size_t n)
{
    const size_t want = ceil_div(n, (size_t)kBlock);
    return (uint32_t)(want < 256 * 8 ? (want ? want : 1) : 256 * 8);
}

}  // namespace

void launch_project_values(const char* projIn, const char* projOut, double* d_x, double* d_y, size_t n, hipStream_t stream)
{
    const ProjParams src = parse_proj4(projIn), dst = parse_proj4(projOut);
    if (n == 0) return;
    project_values_kernel<<<point_blocks(n), kBlock, 0, stream>>>(src, dst, d_x, d_y, n);
    FA_HIP(hipGetLastError());
}

void launch_project_axes(const char* projIn, const char* projOut, const double* h_xAxis, const double* h_yAxis, size_t ix, size_t iy,
                         double* d_outX, double* d_outY, hipStream_t stream)
{
    const ProjParams src = parse_proj4(projIn), dst = parse_proj4(projOut);
    if (ix * iy == 0) return;
    FA_REQUIRE(ix <= 0x7FFFFFFFu && iy <= 0x7FFFFFFFu, "axis too long");
    DeviceArray<double> d_axes(ix + iy);
    FA_HIP(hipMemcpyAsync(d_axes.get(), h_xAxis, ix * sizeof(double), hipMemcpyHostToDevice, stream));
    FA_HIP(hipMemcpyAsync(d_axes.get() + ix, h_yAxis, iy * sizeof(double), hipMemcpyHostToDevice, stream));
    project_axes_kernel<<<point_blocks(ix * iy), kBlock, 0, stream>>>(src, dst, d_axes.get(), d_axes.get() + ix, (uint32_t)ix, (uint32_t)iy, d_outX, d_outY);
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(stream));  // d_axes is released on return
}

// mifi_get_vector_reproject_matrix, interpolation.c:719-788 (axes of longitude / latitude type arrive in degrees, :740-745)
void launch_vector_reproject_matrix(const char* projIn, const char* projOut, const double* h_outXAxis, const double* h_outYAxis,
                                    int xAxisType, int yAxisType, size_t ox, size_t oy, double* d_matrix, hipStream_t stream)
{
    const ProjParams in = parse_proj4(projIn), out = parse_proj4(projOut);
    const size_t n = ox * oy;
    if (n == 0) return;
    FA_REQUIRE(ox <= 0x7FFFFFFFu && oy <= 0x7FFFFFFFu, "axis too long");
    std::vector<double> axes(ox + oy);
    for (size_t i = 0; i < ox; ++i) axes[i] = (xAxisType == FIMEX_AMD_LONGITUDE || xAxisType == FIMEX_AMD_LATITUDE) ? h_outXAxis[i] * kDegToRad : h_outXAxis[i];
    for (size_t i = 0; i < oy; ++i) axes[ox + i] = (yAxisType == FIMEX_AMD_LONGITUDE || yAxisType == FIMEX_AMD_LATITUDE) ? h_outYAxis[i] * kDegToRad : h_outYAxis[i];
    DeviceArray<double> d_axes(ox + oy), d_inX(n), d_inY(n);
    FA_HIP(hipMemcpyAsync(d_axes.get(), axes.data(), axes.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    // positions of the output mesh in the input projection (:773)
    project_axes_kernel<<<point_blocks(n), kBlock, 0, stream>>>(out, in, d_axes.get(), d_axes.get() + ox, (uint32_t)ox, (uint32_t)oy, d_inX.get(), d_inY.get());
    FA_HIP(hipGetLastError());
    // delta: 0.1 % of the distance between neighbouring cells, at the origin and in the middle of the mesh, both taken from
    // the x field as the reference does (:458-513)
    auto at = [&](size_t idx) {
        double v = 0;
        FA_HIP(hipMemcpyAsync(&v, d_inX.get() + idx, sizeof(double), hipMemcpyDeviceToHost, stream));
        FA_HIP(hipStreamSynchronize(stream));
        return v;
    };
    const double d = 1e-3;
    double delta;
    if (ox > 1 && oy > 1) {
        const size_t ox2 = ox / 2, oy2 = oy / 2;
        delta = d * (at(ox + 1) - at(0));
        delta += d * (at((oy2 + 1) * ox + ox2 + 1) - at(oy2 * ox + ox2));
        delta /= 2;
    } else if (ox > 1) {
        delta = d * (at(1) - at(0));
    } else if (oy > 1) {
        delta = d * (at(ox) - at(0));
    } else {
        const double v = at(0);
        delta = (v > 1) ? v * d : d;
    }
    if (std::fabs(delta) < 1e-9) delta = d;  // :514-518
    vector_matrix_kernel<<<point_blocks(n), kBlock, 0, stream>>>(in, out, d_inX.get(), d_inY.get(), d_axes.get(), d_axes.get() + ox, 1, (uint32_t)ox,
                                                                 n, delta, delta, out.kind == kLatLong ? 1 : 0, d_matrix);
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(stream));  // temporaries are released on return
}

namespace {
// the delta of mifi_get_vector_reproject_matrix_proj (:458-518) from the positions in the input projection
double mesh_delta(const double* inX, size_t ox, size_t oy)
{
    const double d = 1e-3;
    double delta;
    if (ox > 1 && oy > 1) {
        const size_t ox2 = ox / 2, oy2 = oy / 2;
        delta = d * (inX[ox + 1] - inX[0]);
        delta += d * (inX[(oy2 + 1) * ox + ox2 + 1] - inX[oy2 * ox + ox2]);
        delta /= 2;
    } else if (ox > 1) {
        delta = d * (inX[1] - inX[0]);
    } else if (oy > 1) {
        delta = d * (inX[ox] - inX[0]);
    } else {
        delta = (inX[0] > 1) ? inX[0] * d : d;
    }
    return std::fabs(delta) < 1e-9 ? d : delta;
}
}  // namespace

// mifi_get_vector_reproject_matrix_field, interpolation.c:657-717: the mesh is given in the INPUT projection
void launch_vector_reproject_matrix_field(const char* projIn, const char* projOut, const double* h_inX, const double* h_inY, size_t ox,
                                          size_t oy, double* d_matrix, hipStream_t stream)
{
    const ProjParams in = parse_proj4(projIn), out = parse_proj4(projOut);
    const size_t n = ox * oy;
    if (n == 0) return;
    DeviceArray<double> d_in(2 * n), d_out(2 * n);
    FA_HIP(hipMemcpyAsync(d_in.get(), h_inX, n * sizeof(double), hipMemcpyHostToDevice, stream));
    FA_HIP(hipMemcpyAsync(d_in.get() + n, h_inY, n * sizeof(double), hipMemcpyHostToDevice, stream));
    FA_HIP(hipMemcpyAsync(d_out.get(), d_in.get(), 2 * n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    project_values_kernel<<<point_blocks(n), kBlock, 0, stream>>>(in, out, d_out.get(), d_out.get() + n, n);  // :696
    FA_HIP(hipGetLastError());
    const double delta = mesh_delta(h_inX, ox, oy);
    vector_matrix_kernel<<<point_blocks(n), kBlock, 0, stream>>>(in, out, d_in.get(), d_in.get() + n, d_out.get(), d_out.get() + n, 0, (uint32_t)ox, n,
                                                                 delta, delta, out.kind == kLatLong ? 1 : 0, d_matrix);
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(stream));
}

// mifi_get_vector_reproject_matrix_points, interpolation.c:607-655: a list of points in the OUTPUT projection (m or rad)
void launch_vector_reproject_matrix_points(const char* projIn, const char* projOut, int inputIsMetric, const double* h_outX,
                                           const double* h_outY, size_t on, double* d_matrix, hipStream_t stream)
{
    const ProjParams in = parse_proj4(projIn), out = parse_proj4(projOut);
    if (on == 0) return;
    DeviceArray<double> d_in(2 * on), d_out(2 * on);
    FA_HIP(hipMemcpyAsync(d_out.get(), h_outX, on * sizeof(double), hipMemcpyHostToDevice, stream));
    FA_HIP(hipMemcpyAsync(d_out.get() + on, h_outY, on * sizeof(double), hipMemcpyHostToDevice, stream));
    FA_HIP(hipMemcpyAsync(d_in.get(), d_out.get(), 2 * on * sizeof(double), hipMemcpyDeviceToDevice, stream));
    project_values_kernel<<<point_blocks(on), kBlock, 0, stream>>>(out, in, d_in.get(), d_in.get() + on, on);  // :635
    FA_HIP(hipGetLastError());
    const double delta = inputIsMetric ? 100 : 0.00001;  // :641
    vector_matrix_kernel<<<point_blocks(on), kBlock, 0, stream>>>(in, out, d_in.get(), d_in.get() + on, d_out.get(), d_out.get() + on, 0, 1, on, delta,
                                                                  delta, out.kind == kLatLong ? 1 : 0, d_matrix);
    FA_HIP(hipGetLastError());
    FA_HIP(hipStreamSynchronize(stream));
}

int projection_is_degree(const char* proj)
{
    const ProjParams p = parse_proj4(proj);
    return (p.kind == kLatLong || p.kind == kObTran) ? 1 : 0;
}

}  // namespace fimex_amd
