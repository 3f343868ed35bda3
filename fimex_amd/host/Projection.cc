#include "Projection.h"

#include <cmath>
#include <sstream>

#include "CachedInterpolation.h"  // CDMException

namespace FimexAmd {

namespace {
const double kPi = 3.14159265358979323846;
const double kHalfPi = kPi / 2, kFortPi = kPi / 4;
const double kSpi = 3.14159265359;  // adjlon threshold of PROJ.4
const double kEps10 = 1e-10;

double adjlon(double lon)
{
    if (std::fabs(lon) <= kSpi) return lon;
    lon += kPi;
    lon -= 2 * kPi * std::floor(lon / (2 * kPi));
    return lon - kPi;
}

bool isGeographicName(const std::string& n) { return n == "latlong" || n == "longlat" || n == "latlon" || n == "lonlat"; }
}  // namespace

double Projection::num(const std::string& k, double dflt) const
{
    auto it = par_.find(k);
    return it == par_.end() ? dflt : std::stod(it->second);
}

double Projection::rad(const std::string& k, double dflt) const
{
    auto it = par_.find(k);
    return it == par_.end() ? dflt : std::stod(it->second) * kPi / 180.0;
}

Projection::Projection(const std::string& proj4) : proj4_(proj4)
{
    std::istringstream in(proj4);
    std::string tok;
    while (in >> tok) {
        while (!tok.empty() && tok[0] == '+') tok.erase(0, 1);
        if (tok.empty()) continue;
        const size_t eq = tok.find('=');
        if (eq == std::string::npos) par_[tok] = "";
        else par_[tok.substr(0, eq)] = tok.substr(eq + 1);
    }
    if (!has("proj")) throw CDMException("projection string without +proj: " + proj4);
    const std::string name = par_["proj"];
    // sphere only
    if (has("R")) a_ = num("R", 1);
    else if (has("a")) {
        const bool sphere = (!has("e") || num("e", 0) == 0.0) && !has("b") && !has("rf") && !has("f") &&
                            (!has("ellps") || par_["ellps"] == "sphere");
        if (!sphere) throw CDMException("ellipsoidal projections are not implemented: " + proj4);
        a_ = num("a", 1);
    } else if (has("ellps") && par_["ellps"] == "sphere") a_ = 6370997.0;
    else if (isGeographicName(name)) a_ = 1;
    else throw CDMException("ellipsoidal projections are not implemented: " + proj4);
    lam0_ = rad("lon_0", 0);
    phi0_ = rad("lat_0", 0);
    x0_ = num("x_0", 0);
    y0_ = num("y_0", 0);
    k0_ = has("k_0") ? num("k_0", 1) : num("k", 1);

    if (isGeographicName(name)) {
        kind_ = Kind::LatLong;
    } else if (name == "stere") {
        kind_ = Kind::Stere;
        double phits = has("lat_ts") ? std::fabs(rad("lat_ts", kHalfPi)) : kHalfPi;
        const double t = std::fabs(phi0_);
        if (std::fabs(t - kHalfPi) < kEps10) mode_ = phi0_ < 0 ? StereMode::South : StereMode::North;
        else mode_ = t > kEps10 ? StereMode::Oblique : StereMode::Equatorial;
        if (mode_ == StereMode::North || mode_ == StereMode::South) {
            akm1_ = (std::fabs(phits - kHalfPi) >= kEps10) ? std::cos(phits) / std::tan(kFortPi - .5 * phits) : 2. * k0_;
        } else {
            sinph0_ = std::sin(phi0_);
            cosph0_ = std::cos(phi0_);
            akm1_ = 2. * k0_;
        }
    } else if (name == "lcc") {
        kind_ = Kind::Lcc;
        const double phi1 = rad("lat_1", 0);
        const double phi2 = has("lat_2") ? rad("lat_2", phi1) : phi1;
        if (!has("lat_0")) phi0_ = phi1;
        const double cosphi = std::cos(phi1);
        n_ = std::sin(phi1);
        if (std::fabs(phi1 - phi2) >= kEps10)
            n_ = std::log(cosphi / std::cos(phi2)) / std::log(std::tan(kFortPi + .5 * phi2) / std::tan(kFortPi + .5 * phi1));
        c_ = cosphi * std::pow(std::tan(kFortPi + .5 * phi1), n_) / n_;
        rho0_ = (std::fabs(std::fabs(phi0_) - kHalfPi) < kEps10) ? 0. : c_ * std::pow(std::tan(kFortPi + .5 * phi0_), -n_);
    } else if (name == "merc") {
        kind_ = Kind::Merc;
        if (has("lat_ts")) k0_ = std::cos(std::fabs(rad("lat_ts", 0)));
    } else if (name == "ob_tran") {
        kind_ = Kind::ObTran;
        if (!has("o_proj") || !isGeographicName(par_["o_proj"]) || !has("o_lat_p"))
            throw CDMException("ob_tran is implemented for +o_proj=longlat +o_lat_p only: " + proj4);
        lamp_ = rad("o_lon_p", 0);
        const double phip = rad("o_lat_p", kHalfPi);
        oblique_ = std::fabs(phip - kHalfPi) > kEps10;
        sphip_ = std::sin(phip);
        cphip_ = std::cos(phip);
    } else {
        throw CDMException("projection not implemented: " + name);
    }
}

void Projection::forward(double lon, double lat, double& x, double& y) const
{
    if (kind_ == Kind::LatLong) { x = lon; y = lat; return; }
    double lam = adjlon(lon - lam0_), phi = lat;
    double px = 0, py = 0;
    switch (kind_) {
    case Kind::Stere: {
        double sinlam = std::sin(lam), coslam = std::cos(lam);
        if (mode_ == StereMode::North || mode_ == StereMode::South) {
            if (mode_ == StereMode::North) { coslam = -coslam; phi = -phi; }
            py = akm1_ * std::tan(kFortPi + .5 * phi);
            px = sinlam * py;
            py *= coslam;
        } else {
            const double sinphi = std::sin(phi), cosphi = std::cos(phi);
            if (mode_ == StereMode::Equatorial) {
                const double k = akm1_ / (1. + cosphi * coslam);
                px = k * cosphi * sinlam;
                py = k * sinphi;
            } else {
                const double k = akm1_ / (1. + sinph0_ * sinphi + cosph0_ * cosphi * coslam);
                px = k * cosphi * sinlam;
                py = k * (cosph0_ * sinphi - sinph0_ * cosphi * coslam);
            }
        }
        break;
    }
    case Kind::Lcc: {
        const double rho = (std::fabs(std::fabs(phi) - kHalfPi) < kEps10) ? 0. : c_ * std::pow(std::tan(kFortPi + .5 * phi), -n_);
        lam *= n_;
        px = k0_ * (rho * std::sin(lam));
        py = k0_ * (rho0_ - rho * std::cos(lam));
        break;
    }
    case Kind::Merc:
        px = k0_ * lam;
        py = k0_ * std::log(std::tan(kFortPi + .5 * phi));
        break;
    case Kind::ObTran: {
        if (oblique_) {
            const double coslam = std::cos(lam), sinphi = std::sin(phi), cosphi = std::cos(phi);
            px = adjlon(std::atan2(cosphi * std::sin(lam), sphip_ * cosphi * coslam + cphip_ * sinphi) + lamp_);
            double s = sphip_ * sinphi - cphip_ * cosphi * coslam;
            s = s > 1 ? 1 : (s < -1 ? -1 : s);
            py = std::asin(s);
        } else {
            px = adjlon(lam + lamp_);
            py = phi;
        }
        x = px + x0_;  // the linked longlat "projection" divides by a again: radians stay radians
        y = py + y0_;
        return;
    }
    default: break;
    }
    x = a_ * px + x0_;
    y = a_ * py + y0_;
}

void Projection::inverse(double x, double y, double& lon, double& lat) const
{
    if (kind_ == Kind::LatLong) { lon = x; lat = y; return; }
    double xs, ys;
    if (kind_ == Kind::ObTran) { xs = x - x0_; ys = y - y0_; }
    else { xs = (x - x0_) / a_; ys = (y - y0_) / a_; }
    double lam = 0, phi = 0;
    switch (kind_) {
    case Kind::Stere: {
        const double rh = std::hypot(xs, ys);
        const double c = 2. * std::atan(rh / akm1_);
        const double sinc = std::sin(c), cosc = std::cos(c);
        if (mode_ == StereMode::North) {
            ys = -ys;
            phi = (std::fabs(rh) <= kEps10) ? phi0_ : std::asin(cosc);
            lam = (xs == 0. && ys == 0.) ? 0. : std::atan2(xs, ys);
        } else if (mode_ == StereMode::South) {
            phi = (std::fabs(rh) <= kEps10) ? phi0_ : std::asin(-cosc);
            lam = (xs == 0. && ys == 0.) ? 0. : std::atan2(xs, ys);
        } else if (mode_ == StereMode::Equatorial) {
            phi = (std::fabs(rh) <= kEps10) ? 0. : std::asin(ys * sinc / rh);
            lam = (cosc != 0. || xs != 0.) ? std::atan2(xs * sinc, cosc * rh) : 0.;
        } else {
            phi = (std::fabs(rh) <= kEps10) ? phi0_ : std::asin(cosc * sinph0_ + ys * sinc * cosph0_ / rh);
            const double cc = cosc - sinph0_ * std::sin(phi);
            lam = (cc != 0. || xs != 0.) ? std::atan2(xs * sinc * cosph0_, cc * rh) : 0.;
        }
        break;
    }
    case Kind::Lcc: {
        xs /= k0_;
        ys = rho0_ - ys / k0_;
        double rho = std::hypot(xs, ys);
        if (rho != 0.) {
            if (n_ < 0.) { rho = -rho; xs = -xs; ys = -ys; }
            phi = 2. * std::atan(std::pow(c_ / rho, 1. / n_)) - kHalfPi;
            lam = std::atan2(xs, ys) / n_;
        } else {
            lam = 0.;
            phi = n_ > 0. ? kHalfPi : -kHalfPi;
        }
        break;
    }
    case Kind::Merc:
        lam = xs / k0_;
        phi = kHalfPi - 2. * std::atan(std::exp(-ys / k0_));
        break;
    case Kind::ObTran:
        if (oblique_) {
            const double lamr = xs - lamp_;
            const double coslam = std::cos(lamr), sinphi = std::sin(ys), cosphi = std::cos(ys);
            double s = sphip_ * sinphi + cphip_ * cosphi * coslam;
            s = s > 1 ? 1 : (s < -1 ? -1 : s);
            phi = std::asin(s);
            lam = std::atan2(cosphi * std::sin(lamr), sphip_ * cosphi * coslam - cphip_ * sinphi);
        } else {
            lam = xs - lamp_;
            phi = ys;
        }
        break;
    default: break;
    }
    lon = adjlon(lam + lam0_);
    lat = phi;
}

void transform(const Projection& src, const Projection& dst, double* x, double* y, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        double lon, lat;
        src.inverse(x[i], y[i], lon, lat);
        dst.forward(lon, lat, x[i], y[i]);
    }
}

void projectAxes(const Projection& in, const Projection& out, const std::vector<double>& xAxis, const std::vector<double>& yAxis,
                 std::vector<double>& outX, std::vector<double>& outY)
{
    const size_t ix = xAxis.size(), iy = yAxis.size();
    outX.resize(ix * iy);
    outY.resize(ix * iy);
    for (size_t y = 0; y < iy; ++y)
        for (size_t x = 0; x < ix; ++x) {
            outX[y * ix + x] = xAxis[x];
            outY[y * ix + x] = yAxis[y];
        }
    transform(in, out, outX.data(), outY.data(), ix * iy);
}

namespace {
// src/interpolation.c:311-329
double bearing(double lat0, double lon0, double lat1, double lon1)
{
    const double dlon = lon0 - lon1;
    return std::atan2(std::sin(dlon) * std::cos(lat1), std::cos(lat0) * std::sin(lat1) - std::sin(lat0) * std::cos(lat1) * std::cos(dlon));
}
}  // namespace

void vectorReprojectMatrix(const Projection& in, const Projection& out, const std::vector<double>& outXAxis,
                           const std::vector<double>& outYAxis, std::vector<double>& matrix)
{
    const size_t ox = outXAxis.size(), oy = outYAxis.size(), n = ox * oy;
    std::vector<double> outXf(n), outYf(n), inX, inY;
    for (size_t y = 0; y < oy; ++y)
        for (size_t x = 0; x < ox; ++x) { outXf[y * ox + x] = outXAxis[x]; outYf[y * ox + x] = outYAxis[y]; }
    inX = outXf;
    inY = outYf;
    transform(out, in, inX.data(), inY.data(), n);  // positions in the original projection (:773)

    // delta: 0.1 % of the distance between neighbouring cells, taken from the x field for both directions as the
    // reference does (:458-513)
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
    if (std::fabs(delta) < 1e-9) delta = d;

    std::vector<double> ax(n), ay(n), bx(n), by(n);
    for (size_t i = 0; i < n; ++i) { ax[i] = inX[i] + delta; ay[i] = inY[i]; bx[i] = inX[i]; by[i] = inY[i] + delta; }
    transform(in, out, ax.data(), ay.data(), n);  // (x + d, y) (:355)
    transform(in, out, bx.data(), by.data(), n);  // (x, y + d) (:396)
    const double sign = delta > 0 ? 1. : -1.;
    const bool latlon = out.isLatLong();
    matrix.resize(4 * n);
    for (size_t i = 0; i < n; ++i) {
        double phiy, phi0;
        if (latlon) {
            phiy = bearing(outYf[i], outXf[i], ay[i], ax[i]);  // :367 (not used further for lat/lon output)
            phi0 = bearing(outYf[i], outXf[i], by[i], bx[i]);  // :409
            if (sign < 0) phi0 += kPi;
            (void)phiy;
        } else {
            phiy = std::atan2(ay[i] - outYf[i], ax[i] - outXf[i]);  // :372-373
            if (sign < 0) phiy += kPi;
            double phix = -1 * std::atan2(bx[i] - outXf[i], by[i] - outYf[i]);  // :414-415
            if (sign < 0) phix += kPi;
            phi0 = .5 * (phix + phiy);  // :424
        }
        const double c = std::cos(phi0), s = std::sin(phi0);
        matrix[4 * i + 0] = c;
        matrix[4 * i + 1] = s;
        matrix[4 * i + 2] = -1 * s;
        matrix[4 * i + 3] = phi0;
    }
}

}  // namespace FimexAmd
