// Plan building on the device: the coordinate transforms the reference obtains from PROJ.4's
// pj_transform -- mifi_project_values / mifi_project_axes (src/interpolation.c:1158-1244) and the local rotation
// matrix mifi_get_vector_reproject_matrix (:719-788 -> :441-521 -> :330-438) -- for 4 M-cell target grids.
//
// PROJ.4 is a third-party library that is not part of the reference tree; the projections are implemented from their
// published closed forms on the sphere (Snyder, "Map Projections - A Working Manual", USGS PP 1395) with PROJ.4's
// conventions at the pj_transform boundary: geographic coordinates in radians, projected x = a * x' + x_0, longitudes
// relative to lon_0 wrapped to [-pi, pi].  Supported: latlong/longlat, stere, lcc, merc, tmerc, etmerc, utm, laea, aea, geos, omerc, sinu, cea, ortho, aeqd, nsper, ob_tran +
// o_proj=longlat; on the sphere and (except ob_tran and the equatorial stereographic, where PROJ.4 releases differ) on
// an ellipsoid given by +ellps / +datum=WGS84|NAD83 / +a with +b, +rf, +f, +e or +es, with the series PROJ.4 4.x uses
// (Snyder eq. 7-7, 7-9, 15-7..15-11, 21-33..21-40, 8-9..8-25, 3-21).  Geodetic coordinates pass unchanged between the
// two sides unless both name a datum (+datum, +towgs84) and the two differ: then pj_datum_transform's three- or
// seven-parameter shift is applied at height 0.  Everything else not implemented is refused (+units, +to_meter, +pm,
// +geoc, +over, grid shifts).  +units / +to_meter scale the projected coordinates as pj_fwd / pj_inv do, +pm shifts
// longitudes as pj_transform does.
#include "plan.hpp"

#include <cmath>
#include <cstdio>
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

enum ProjKind { kLatLong = 0, kStere, kLcc, kMerc, kObTran, kTmerc, kEtmerc, kLaea, kAea, kGeos, kOmerc, kSinu, kCea, kOrtho, kAeqd, kNsper };
enum StereMode { kNorth = 0, kSouth, kOblique, kEquatorial };

struct ProjParams {
    int kind, mode, oblique, datum;
    double a, es, e, lam0, phi0, x0, y0, k0;
    double akm1, sinph0, cosph0;  // stere (on an ellipsoid: sin, cos of the conformal latitude of the origin)
    double n, c, rho0;            // lcc
    double lamp, sphip, cphip;    // ob_tran
    double esp, ml0, en[5];       // tmerc
    double Qn, Zb, cgb[6], cbg[6], utg[6], gtu[6];  // etmerc
    double qp, rq, dd, xmf, ymf, sinb1, cosb1, apa[3];  // laea (aea: dd, and n, c, rho0 of lcc)
    double ec, n2;                // aea
    double radius_g, radius_g_1, radius_p, radius_p2, radius_p_inv2, C;  // geos (flip_axis in mode)
    double pn1, pp, rp, pfact;    // nsper (sinph0 / cosph0 and the aspect in mode as for stere; sinu: en; cea: k0, qp, apa)
    double oA, oB, oE, ArB, BrA, rB, singam, cosgam, sinrot, cosrot, v_pole_n, v_pole_s, u_0;  // omerc (no_rot in mode)
    double towgs84[7];            // pj_datum_set: dx dy dz (m), rx ry rz (rad), scale factor
    double toMeter, frMeter;      // pj_init: +units / +to_meter (1 for metres); pj_fwd multiplies by fr_meter, pj_inv by to_meter
    double fromGreenwich;         // pj_init: +pm, radians east of Greenwich
    int datumType, doShift;       // 0 unknown, 1 three parameters, 2 seven, 3 WGS84; doShift: set on both sides of a pair that needs pj_datum_transform
};

struct Ellipsoid { const char* name; double a; bool byB; double shape; };  // pj_ellps.c
constexpr Ellipsoid kEllipsoids[] = {
    {"sphere", 6370997.0, true, 6370997.0},  {"WGS84", 6378137.0, false, 298.257223563}, {"GRS80", 6378137.0, false, 298.257222101},
    {"WGS72", 6378135.0, false, 298.26},     {"GRS67", 6378160.0, false, 298.2471674270}, {"bessel", 6377397.155, false, 299.1528128},
    {"intl", 6378388.0, false, 297.},        {"clrk66", 6378206.4, true, 6356583.8},      {"clrk80", 6378249.145, false, 293.4663},
    {"krass", 6378245.0, false, 298.3},      {"airy", 6377563.396, true, 6356256.910},      {"evrstSS", 6377298.556, false, 300.8017},
};

// pj_tsfn / pj_msfn / pj_phi2 / pj_enfn / pj_mlfn / pj_inv_mlfn
__host__ __device__ inline double tsfn(double phi, double sinphi, double e)
{
    sinphi *= e;
    return tan(.5 * (kHalfPi - phi)) / pow((1. - sinphi) / (1. + sinphi), .5 * e);
}
__host__ __device__ inline double msfn(double sinphi, double cosphi, double es) { return cosphi / sqrt(1. - es * sinphi * sinphi); }
// pj_qsfn (Snyder eq. 3-12)
__host__ __device__ inline double qsfn(double sinphi, double e, double one_es)
{
    if (e >= 1e-7) {
        const double con = e * sinphi;
        return one_es * (sinphi / (1. - con * con) - (.5 / e) * log((1. - con) / (1. + con)));
    }
    return sinphi + sinphi;
}
__host__ __device__ inline double ssfn(double phit, double sinphi, double e)
{
    sinphi *= e;
    return tan(.5 * (kHalfPi + phit)) * pow((1. - sinphi) / (1. + sinphi), .5 * e);
}
__device__ inline double phi2(double ts, double e)
{
    const double eccnth = .5 * e;
    double phi = kHalfPi - 2. * atan(ts), dphi;
    int i = 15;
    do {
        const double con = e * sin(phi);
        dphi = kHalfPi - 2. * atan(ts * pow((1. - con) / (1. + con), eccnth)) - phi;
        phi += dphi;
    } while (fabs(dphi) > 1e-10 && --i);
    return i ? phi : NAN;  // pj_phi2 reports an error after 15 rounds
}
void enfn(double es, double* en)
{
    constexpr double C00 = 1., C02 = .25, C04 = .046875, C06 = .01953125, C08 = .01068115234375, C22 = .75, C44 = .46875,
                     C46 = .01302083333333333333, C48 = .00712076822916666666, C66 = .36458333333333333333,
                     C68 = .00569661458333333333, C88 = .3076171875;
    double t;
    en[0] = C00 - es * (C02 + es * (C04 + es * (C06 + es * C08)));
    en[1] = es * (C22 - es * (C04 + es * (C06 + es * C08)));
    en[2] = (t = es * es) * (C44 - es * (C46 + es * C48));
    en[3] = (t *= es) * (C66 - es * C68);
    en[4] = t * es * C88;
}
__host__ __device__ inline double mlfn(double phi, double sphi, double cphi, const double* en)
{
    cphi *= sphi;
    sphi *= sphi;
    return en[0] * phi - cphi * (en[1] + sphi * (en[2] + sphi * (en[3] + sphi * en[4])));
}
__device__ inline double inv_mlfn(double arg, double es, const double* en)
{
    const double k = 1. / (1. - es);
    double phi = arg;
    for (int i = 10; i; --i) {
        const double s = sin(phi);
        double t = 1. - es * s * s;
        phi -= t = (mlfn(phi, s, cos(phi), en) - arg) * (t * sqrt(t)) * k;
        if (fabs(t) < 1e-11) return phi;
    }
    return NAN;
}

// PJ_etmerc.c: Clenshaw summation of sum p[k] sin(2 (k+1) B) + B, of the real sine series, and of the complex one
__host__ __device__ inline double gatg(const double* p1, double B)
{
    const double cos2B = 2 * cos(2 * B);
    double h = 0, h1 = p1[5], h2 = 0;
    for (int k = 4; k >= 0; --k) {
        h = -h2 + cos2B * h1 + p1[k];
        h2 = h1;
        h1 = h;
    }
    return B + h * sin(2 * B);
}
inline double clens(const double* a, double argR)
{
    const double r = 2 * std::cos(argR);
    double hr = a[5], hr1 = 0, hr2;
    for (int k = 4; k >= 0; --k) {
        hr2 = hr1;
        hr1 = hr;
        hr = -hr2 + r * hr1 + a[k];
    }
    return std::sin(argR) * hr;
}
__device__ inline void clenS(const double* a, double argR, double argI, double& R, double& I)
{
    const double sinR = sin(argR), cosR = cos(argR), sinhI = sinh(argI), coshI = cosh(argI);
    double r = 2 * cosR * coshI, i = -2 * sinR * sinhI;
    double hr = a[5], hi = 0, hr1 = 0, hi1 = 0, hr2, hi2;
    for (int k = 4; k >= 0; --k) {
        hr2 = hr1;
        hi2 = hi1;
        hr1 = hr;
        hi1 = hi;
        hr = -hr2 + r * hr1 - i * hi1 + a[k];
        hi = -hi2 + i * hr1 + r * hi1;
    }
    r = sinR * coshI;
    i = cosR * sinhI;
    R = r * hr - i * hi;
    I = r * hi + i * hr;
}

// Engsager & Poder's series to the sixth power of the third flattening (PJ_etmerc.c setup)
void setup_etmerc(ProjParams& p)
{
    const double f = p.es / (1 + std::sqrt(1 - p.es));
    const double n = f / (2 - f);
    double np = n;
    p.cgb[0] = n * (2 + n * (-2 / 3.0 + n * (-2 + n * (116 / 45.0 + n * (26 / 45.0 + n * (-2854 / 675.0))))));
    p.cbg[0] = n * (-2 + n * (2 / 3.0 + n * (4 / 3.0 + n * (-82 / 45.0 + n * (32 / 45.0 + n * (4642 / 4725.0))))));
    np *= n;
    p.cgb[1] = np * (7 / 3.0 + n * (-8 / 5.0 + n * (-227 / 45.0 + n * (2704 / 315.0 + n * (2323 / 945.0)))));
    p.cbg[1] = np * (5 / 3.0 + n * (-16 / 15.0 + n * (-13 / 9.0 + n * (904 / 315.0 + n * (-1522 / 945.0)))));
    np *= n;
    p.cgb[2] = np * (56 / 15.0 + n * (-136 / 35.0 + n * (-1262 / 105.0 + n * (73814 / 2835.0))));
    p.cbg[2] = np * (-26 / 15.0 + n * (34 / 21.0 + n * (8 / 5.0 + n * (-12686 / 2835.0))));
    np *= n;
    p.cgb[3] = np * (4279 / 630.0 + n * (-332 / 35.0 + n * (-399572 / 14175.0)));
    p.cbg[3] = np * (1237 / 630.0 + n * (-12 / 5.0 + n * (-24832 / 14175.0)));
    np *= n;
    p.cgb[4] = np * (4174 / 315.0 + n * (-144838 / 6237.0));
    p.cbg[4] = np * (-734 / 315.0 + n * (109598 / 31185.0));
    np *= n;
    p.cgb[5] = np * (601676 / 22275.0);
    p.cbg[5] = np * (444337 / 155925.0);
    np = n * n;
    p.Qn = p.k0 / (1 + n) * (1 + np * (1 / 4.0 + np * (1 / 64.0 + np / 256.0)));
    p.utg[0] = n * (-0.5 + n * (2 / 3.0 + n * (-37 / 96.0 + n * (1 / 360.0 + n * (81 / 512.0 + n * (-96199 / 604800.0))))));
    p.gtu[0] = n * (0.5 + n * (-2 / 3.0 + n * (5 / 16.0 + n * (41 / 180.0 + n * (-127 / 288.0 + n * (7891 / 37800.0))))));
    p.utg[1] = np * (-1 / 48.0 + n * (-1 / 15.0 + n * (437 / 1440.0 + n * (-46 / 105.0 + n * (1118711 / 3870720.0)))));
    p.gtu[1] = np * (13 / 48.0 + n * (-3 / 5.0 + n * (557 / 1440.0 + n * (281 / 630.0 + n * (-1983433 / 1935360.0)))));
    np *= n;
    p.utg[2] = np * (-17 / 480.0 + n * (37 / 840.0 + n * (209 / 4480.0 + n * (-5569 / 90720.0))));
    p.gtu[2] = np * (61 / 240.0 + n * (-103 / 140.0 + n * (15061 / 26880.0 + n * (167603 / 181440.0))));
    np *= n;
    p.utg[3] = np * (-4397 / 161280.0 + n * (11 / 504.0 + n * (830251 / 7257600.0)));
    p.gtu[3] = np * (49561 / 161280.0 + n * (-179 / 168.0 + n * (6601661 / 7257600.0)));
    np *= n;
    p.utg[4] = np * (-4583 / 161280.0 + n * (108847 / 3991680.0));
    p.gtu[4] = np * (34729 / 80640.0 + n * (-3418889 / 1995840.0));
    np *= n;
    p.utg[5] = np * (-20648693 / 638668800.0);
    p.gtu[5] = np * (212378941 / 319334400.0);
    const double Z = gatg(p.cbg, p.phi0);
    p.Zb = -p.Qn * (Z + clens(p.gtu, 2 * Z));
}

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
    for (const char* k : {"geoc", "over", "vto_meter", "nadgrids", "geoidgrids", "R_A", "R_V", "R_a", "R_g", "R_h", "R_lat_a", "R_lat_g"})
        if (has(k)) throw Error("projection parameter +" + std::string(k) + " is not implemented: " + proj4);
    if (has("axis") && par["axis"] != "enu") throw Error("projection +axis other than enu is not implemented: " + proj4);
    ProjParams p{};
    // pj_init: +to_meter=<number>[/<number>] wins over +units=<name> (pj_units.c)
    p.toMeter = 1;
    if (has("to_meter") || has("units")) {
        std::string v;
        if (has("to_meter")) v = par["to_meter"];
        else {
            static const struct { const char* id; const char* toMeter; } kUnits[] = {
                {"km", "1000."}, {"m", "1."}, {"dm", "1/10"}, {"cm", "1/100"}, {"mm", "1/1000"}, {"kmi", "1852.0"}, {"in", "0.0254"},
                {"ft", "0.3048"}, {"yd", "0.9144"}, {"mi", "1609.344"}, {"fath", "1.8288"}, {"ch", "20.1168"}, {"link", "0.201168"},
                {"us-in", "1./39.37"}, {"us-ft", "0.304800609601219"}, {"us-yd", "0.914401828803658"}, {"us-ch", "20.11684023368047"},
                {"us-mi", "1609.347218694437"}, {"ind-yd", "0.91439523"}, {"ind-ft", "0.30479841"}, {"ind-ch", "20.11669506"}};
            for (const auto& u : kUnits)
                if (par["units"] == u.id) v = u.toMeter;
            if (v.empty()) throw Error("unknown +units: " + proj4);
        }
        try {
            size_t used = 0;
            p.toMeter = std::stod(v, &used);
            if (used < v.size() && v[used] == '/') p.toMeter /= std::stod(v.substr(used + 1));
        } catch (...) { throw Error("+to_meter is not a number: " + proj4); }
        if (!(p.toMeter > 0)) throw Error("invalid +to_meter: " + proj4);
    }
    p.frMeter = 1. / p.toMeter;
    if (has("pm")) {  // pj_init: a name of pj_prime_meridians or an angle (here: decimal degrees, positive east)
        static const struct { const char* id; double deg; } kMeridians[] = {
            {"greenwich", 0.}, {"lisbon", -(9 + 7 / 60. + 54.862 / 3600.)}, {"paris", 2 + 20 / 60. + 14.025 / 3600.},
            {"bogota", -(74 + 4 / 60. + 51.3 / 3600.)}, {"madrid", -(3 + 41 / 60. + 16.58 / 3600.)}, {"rome", 12 + 27 / 60. + 8.4 / 3600.},
            {"bern", 7 + 26 / 60. + 22.5 / 3600.}, {"jakarta", 106 + 48 / 60. + 27.79 / 3600.}, {"ferro", -(17 + 40 / 60.)},
            {"brussels", 4 + 22 / 60. + 4.71 / 3600.}, {"stockholm", 18 + 3 / 60. + 29.8 / 3600.}, {"athens", 23 + 42 / 60. + 58.815 / 3600.},
            {"oslo", 10 + 43 / 60. + 22.5 / 3600.}};
        bool named = false;
        for (const auto& m : kMeridians)
            if (par["pm"] == m.id) { p.fromGreenwich = m.deg * kPi / 180.0; named = true; }
        if (!named) {
            size_t used = 0;
            double deg = 0;
            try { deg = std::stod(par["pm"], &used); } catch (...) { used = 0; }
            if (used == 0 || used != par["pm"].size()) throw Error("+pm is neither a known meridian nor decimal degrees: " + proj4);
            p.fromGreenwich = deg * kPi / 180.0;
        }
    }
    // pj_ell_set: an explicit +a wins over the one +ellps implies; the shape is the first of +es +e +rf +f +b
    p.a = 1;
    p.datum = has("datum") || has("towgs84");
    if (has("datum") && !has("towgs84")) {  // pj_datums.c (entries without a grid)
        const std::string& d = par["datum"];
        if (d == "WGS84" || d == "NAD83") par["towgs84"] = "0,0,0";
        else if (d == "GGRS87") par["towgs84"] = "-199.87,74.79,246.62";
        else if (d == "potsdam") par["towgs84"] = "598.1,73.7,418.2,0.202,0.045,-2.455,6.7";
        else throw Error("datum not implemented: " + proj4);
    }
    if (has("towgs84")) {
        std::istringstream list(par["towgs84"]);
        std::string item;
        for (int i = 0; i < 7 && std::getline(list, item, ','); ++i) {
            try { p.towgs84[i] = std::stod(item); } catch (...) { throw Error("+towgs84 is not a list of numbers: " + proj4); }
        }
        if (p.towgs84[3] != 0 || p.towgs84[4] != 0 || p.towgs84[5] != 0 || p.towgs84[6] != 0) {
            p.datumType = 2;
            for (int i = 3; i < 6; ++i) p.towgs84[i] *= 4.84813681109535993589914102357e-6;  // SEC_TO_RAD
            p.towgs84[6] = p.towgs84[6] / 1000000.0 + 1;
        } else p.datumType = 1;
    }
    if (has("R")) p.a = num("R", 1);
    else {
        if (has("datum") && !has("ellps")) {
            if (par["datum"] == "WGS84") par["ellps"] = "WGS84";
            else if (par["datum"] == "NAD83" || par["datum"] == "GGRS87") par["ellps"] = "GRS80";
            else if (par["datum"] == "potsdam") par["ellps"] = "bessel";
            else throw Error("datum not implemented: " + proj4);
        }
        if (has("ellps")) {
            const Ellipsoid* ell = nullptr;
            for (const Ellipsoid& cand : kEllipsoids)
                if (par["ellps"] == cand.name) ell = &cand;
            if (!ell) throw Error("ellipsoid not implemented: " + proj4);
            char buf[64];
            if (!has("a")) { std::snprintf(buf, sizeof buf, "%.17g", ell->a); par["a"] = buf; }
            if (!has("es") && !has("e") && !has("rf") && !has("f") && !has("b")) {
                std::snprintf(buf, sizeof buf, "%.17g", ell->shape);
                par[ell->byB ? "b" : "rf"] = buf;
            }
        }
        if (has("a")) {
            p.a = num("a", 1);
            if (has("es")) p.es = num("es", 0);
            else if (has("e")) { p.es = num("e", 0); p.es *= p.es; }
            else if (has("rf")) { p.es = 1. / num("rf", 1); p.es = p.es * (2. - p.es); }
            else if (has("f")) { p.es = num("f", 0); p.es = p.es * (2. - p.es); }
            else if (has("b")) { const double b = num("b", 1); p.es = 1. - (b * b) / (p.a * p.a); }
            if (!(p.a > 0) || p.es < 0 || p.es >= 1) throw Error("invalid ellipsoid: " + proj4);
        } else if (!geographic_name(name)) {
            throw Error("projection string without an ellipsoid (+R, +a or +ellps): " + proj4);
        }
    }
    p.e = std::sqrt(p.es);
    if (p.datumType == 1 && p.towgs84[0] == 0 && p.towgs84[1] == 0 && p.towgs84[2] == 0 && p.a == 6378137.0 &&
        std::fabs(p.es - 0.006694379990) < 0.000000000050)
        p.datumType = 3;  // pj_init: PJD_WGS84
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
        if (p.es != 0) {
            if (p.mode == kEquatorial) throw Error("the equatorial stereographic projection is implemented on the sphere only: " + proj4);
            if (p.mode == kNorth || p.mode == kSouth) {
                if (std::fabs(phits - kHalfPi) < kEps10) p.akm1 = 2. * p.k0 / std::sqrt(std::pow(1 + p.e, 1 + p.e) * std::pow(1 - p.e, 1 - p.e));
                else {
                    double t = std::sin(phits);
                    p.akm1 = std::cos(phits) / tsfn(phits, t, p.e);
                    t *= p.e;
                    p.akm1 /= std::sqrt(1. - t * t);
                }
            } else {
                double t = std::sin(p.phi0);
                const double X = 2. * std::atan(ssfn(p.phi0, t, p.e)) - kHalfPi;
                t *= p.e;
                p.akm1 = 2. * p.k0 * std::cos(p.phi0) / std::sqrt(1. - t * t);
                p.sinph0 = std::sin(X);
                p.cosph0 = std::cos(X);
            }
        } else if (p.mode == kNorth || p.mode == kSouth) {
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
        if (std::fabs(phi1 + phi2) < kEps10) throw Error("lcc: lat_1 = -lat_2: " + proj4);
        const double sinphi = std::sin(phi1), cosphi = std::cos(phi1);
        const bool secant = std::fabs(phi1 - phi2) >= kEps10;
        p.n = sinphi;
        if (p.es != 0) {
            const double m1 = msfn(sinphi, cosphi, p.es), ml1 = tsfn(phi1, sinphi, p.e);
            if (secant) {
                const double sinphi2 = std::sin(phi2);
                p.n = std::log(m1 / msfn(sinphi2, std::cos(phi2), p.es));
                p.n /= std::log(ml1 / tsfn(phi2, sinphi2, p.e));
            }
            p.c = p.rho0 = m1 * std::pow(ml1, -p.n) / p.n;
            p.rho0 *= (std::fabs(std::fabs(p.phi0) - kHalfPi) < kEps10) ? 0. : std::pow(tsfn(p.phi0, std::sin(p.phi0), p.e), p.n);
        } else {
            if (secant) p.n = std::log(cosphi / std::cos(phi2)) / std::log(std::tan(kFortPi + .5 * phi2) / std::tan(kFortPi + .5 * phi1));
            p.c = cosphi * std::pow(std::tan(kFortPi + .5 * phi1), p.n) / p.n;
            p.rho0 = (std::fabs(std::fabs(p.phi0) - kHalfPi) < kEps10) ? 0. : p.c * std::pow(std::tan(kFortPi + .5 * p.phi0), -p.n);
        }
    } else if (name == "merc") {
        p.kind = kMerc;
        if (has("lat_ts")) {
            const double phits = std::fabs(rad("lat_ts", 0));
            if (phits >= kHalfPi) throw Error("merc: lat_ts >= 90: " + proj4);
            p.k0 = p.es != 0 ? msfn(std::sin(phits), std::cos(phits), p.es) : std::cos(phits);
        }
    } else if (name == "tmerc" || name == "utm" || name == "etmerc") {
        // utm is etmerc since PROJ.4 4.9.3 (the release debian_bionic/control builds against) and tmerc before; the two
        // differ by < 1 mm within 6 degrees of the meridian, but only etmerc inverts itself far from it
        p.kind = name == "tmerc" ? kTmerc : kEtmerc;
        if (name == "utm") {  // PJ_tmerc.c / PJ_etmerc.c, utm entry
            if (p.es == 0) throw Error("utm needs an ellipsoid: " + proj4);
            p.y0 = has("south") ? 10000000. : 0.;
            p.x0 = 500000.;
            long zone;
            if (has("zone")) {
                zone = (long)num("zone", 0);
                if (zone < 1 || zone > 60) throw Error("invalid UTM zone: " + proj4);
                --zone;
            } else {
                double l = p.lam0;
                if (std::fabs(l) > kSpi) { l += kPi; l -= 2 * kPi * std::floor(l / (2 * kPi)); l -= kPi; }
                zone = (long)std::floor((l + kPi) * 30. / kPi);
                zone = zone < 0 ? 0 : (zone >= 60 ? 59 : zone);
            }
            p.lam0 = (zone + .5) * kPi / 30. - kPi;
            p.k0 = 0.9996;
            p.phi0 = 0.;
        }
        if (p.kind == kEtmerc) {
            if (p.es == 0) throw Error("etmerc needs an ellipsoid: " + proj4);
            setup_etmerc(p);
        } else if (p.es != 0) {
            enfn(p.es, p.en);
            p.ml0 = mlfn(p.phi0, std::sin(p.phi0), std::cos(p.phi0), p.en);
            p.esp = p.es / (1. - p.es);
        } else {
            p.esp = p.k0;
            p.ml0 = .5 * p.esp;
        }
    } else if (name == "sinu") {  // PJ_gn_sinu.c
        p.kind = kSinu;
        if (p.es != 0) enfn(p.es, p.en);
    } else if (name == "cea") {  // PJ_cea.c
        p.kind = kCea;
        const double t = rad("lat_ts", 0);
        p.k0 = std::cos(t);
        if (p.k0 < 0) throw Error("cea: |lat_ts| > 90: " + proj4);
        if (p.es != 0) {
            const double st = std::sin(t);
            p.k0 /= std::sqrt(1. - p.es * st * st);
            double t2 = p.es * p.es;  // pj_authset
            p.apa[0] = p.es * .33333333333333333333 + t2 * .17222222222222222222;
            p.apa[1] = t2 * .06388888888888888888;
            t2 *= p.es;
            p.apa[0] += t2 * .10257936507936507936;
            p.apa[1] += t2 * .06640211640211640211;
            p.apa[2] = t2 * .01641501294219154443;
            p.qp = qsfn(1., p.e, 1. - p.es);
        }
    } else if (name == "ortho" || name == "aeqd" || name == "nsper") {  // PJ_ortho.c, PJ_aeqd.c (sphere), PJ_nsper.c
        p.kind = name == "ortho" ? kOrtho : (name == "aeqd" ? kAeqd : kNsper);
        if (p.es != 0) throw Error(name + " is implemented on the sphere only: " + proj4);
        const double t = std::fabs(p.phi0);
        if (std::fabs(t - kHalfPi) < kEps10) p.mode = p.phi0 < 0 ? kSouth : kNorth;
        else p.mode = t < kEps10 ? kEquatorial : kOblique;
        p.sinph0 = std::sin(p.phi0);
        p.cosph0 = std::cos(p.phi0);
        if (name == "nsper") {
            const double height = num("h", 0);
            if (!(height > 0)) throw Error("nsper needs +h > 0: " + proj4);
            p.pn1 = height / p.a;
            p.pp = 1. + p.pn1;
            p.rp = 1. / p.pp;
            p.pfact = (p.pp + 1.) / p.pn1;
        }
    } else if (name == "omerc") {  // PJ_omerc.c setup, central point and azimuth (Snyder's alternate B)
        p.kind = kOmerc;
        const bool alp = has("alpha"), gam = has("gamma");
        if (!alp && !gam) throw Error("omerc is implemented by central point and azimuth (+lonc +alpha / +gamma) only: " + proj4);
        p.mode = has("no_rot");
        const bool no_off = has("no_off") || has("no_uoff");
        double alpha_c = rad("alpha", 0), gamma = rad("gamma", 0), gamma0;
        const double lamc = rad("lonc", 0), com = std::sqrt(1. - p.es);
        double D, F;
        if (std::fabs(p.phi0) > kEps10) {
            const double sinph0 = std::sin(p.phi0), cosph0 = std::cos(p.phi0);
            const double con = 1. - p.es * sinph0 * sinph0;
            p.oB = cosph0 * cosph0;
            p.oB = std::sqrt(1. + p.es * p.oB * p.oB / (1. - p.es));
            p.oA = p.oB * p.k0 * com / con;
            D = p.oB * com / (cosph0 * std::sqrt(con));
            if ((F = D * D - 1.) <= 0.) F = 0.;
            else {
                F = std::sqrt(F);
                if (p.phi0 < 0.) F = -F;
            }
            p.oE = F += D;
            p.oE *= std::pow(tsfn(p.phi0, sinph0, p.e), p.oB);
        } else {
            p.oB = 1. / com;
            p.oA = p.k0;
            p.oE = D = F = 1.;
        }
        if (alp) {
            gamma0 = std::asin(std::sin(alpha_c) / D);
            if (!gam) gamma = alpha_c;
        } else {
            alpha_c = std::asin(D * std::sin(gamma0 = gamma));
        }
        const double con = std::fabs(alpha_c);
        if (con <= 1e-7 || std::fabs(con - kPi) <= 1e-7 || std::fabs(std::fabs(p.phi0) - kHalfPi) <= 1e-7)
            throw Error("omerc: alpha of 0 or 180 degrees, or lat_0 at a pole: " + proj4);
        p.lam0 = lamc - std::asin(.5 * (F - 1. / F) * std::tan(gamma0)) / p.oB;  // +lon_0 plays no role
        p.singam = std::sin(gamma0);
        p.cosgam = std::cos(gamma0);
        p.sinrot = std::sin(gamma);
        p.cosrot = std::cos(gamma);
        p.BrA = 1. / (p.ArB = p.oA * (p.rB = 1. / p.oB));
        if (no_off) p.u_0 = 0;
        else {
            p.u_0 = std::fabs(p.ArB * std::atan2(std::sqrt(D * D - 1.), std::cos(alpha_c)));
            if (p.phi0 < 0.) p.u_0 = -p.u_0;
        }
        p.v_pole_n = p.ArB * std::log(std::tan(kFortPi - .5 * gamma0));
        p.v_pole_s = p.ArB * std::log(std::tan(kFortPi + .5 * gamma0));
    } else if (name == "geos") {  // PJ_geos.c setup
        p.kind = kGeos;
        const double h = num("h", 0);
        if (!(h > 0)) throw Error("geos needs +h > 0: " + proj4);
        if (p.phi0 != 0) throw Error("geos: lat_0 must be 0: " + proj4);
        p.mode = 0;
        if (has("sweep")) {
            if (par["sweep"] != "x" && par["sweep"] != "y") throw Error("geos: +sweep must be x or y: " + proj4);
            p.mode = par["sweep"] == "x";
        }
        p.radius_g_1 = h / p.a;
        p.radius_g = 1. + p.radius_g_1;
        p.C = p.radius_g * p.radius_g - 1.0;
        if (p.es != 0) {
            p.radius_p = std::sqrt(1. - p.es);
            p.radius_p2 = 1. - p.es;
            p.radius_p_inv2 = 1. / (1. - p.es);
        } else {
            p.radius_p = p.radius_p2 = p.radius_p_inv2 = 1.;
        }
    } else if (name == "aea") {  // PJ_aea.c setup
        p.kind = kAea;
        const double phi1 = rad("lat_1", 0), phi2 = rad("lat_2", 0);
        if (std::fabs(phi1 + phi2) < kEps10) throw Error("aea: lat_1 = -lat_2: " + proj4);
        double sinphi = std::sin(phi1), cosphi = std::cos(phi1);
        p.n = sinphi;
        const bool secant = std::fabs(phi1 - phi2) >= kEps10;
        if (p.es != 0) {
            const double one_es = 1. - p.es;
            const double m1 = msfn(sinphi, cosphi, p.es), ml1 = qsfn(sinphi, p.e, one_es);
            if (secant) {
                sinphi = std::sin(phi2);
                cosphi = std::cos(phi2);
                const double m2 = msfn(sinphi, cosphi, p.es), ml2 = qsfn(sinphi, p.e, one_es);
                p.n = (m1 * m1 - m2 * m2) / (ml2 - ml1);
            }
            p.ec = 1. - .5 * one_es * std::log((1. - p.e) / (1. + p.e)) / p.e;
            p.c = m1 * m1 + p.n * ml1;
            p.dd = 1. / p.n;
            p.rho0 = p.dd * std::sqrt(p.c - p.n * qsfn(std::sin(p.phi0), p.e, one_es));
        } else {
            if (secant) p.n = .5 * (p.n + std::sin(phi2));
            p.n2 = p.n + p.n;
            p.c = cosphi * cosphi + p.n2 * sinphi;
            p.dd = 1. / p.n;
            p.rho0 = p.dd * std::sqrt(p.c - p.n2 * std::sin(p.phi0));
        }
    } else if (name == "laea") {  // PJ_laea.c setup
        p.kind = kLaea;
        const double t = std::fabs(p.phi0);
        if (std::fabs(t - kHalfPi) < kEps10) p.mode = p.phi0 < 0 ? kSouth : kNorth;
        else p.mode = t < kEps10 ? kEquatorial : kOblique;
        if (p.es != 0) {
            const double one_es = 1. - p.es;
            p.qp = qsfn(1., p.e, one_es);
            double t2 = p.es * p.es;  // pj_authset
            p.apa[0] = p.es * .33333333333333333333 + t2 * .17222222222222222222;
            p.apa[1] = t2 * .06388888888888888888;
            t2 *= p.es;
            p.apa[0] += t2 * .10257936507936507936;
            p.apa[1] += t2 * .06640211640211640211;
            p.apa[2] = t2 * .01641501294219154443;
            if (p.mode == kNorth || p.mode == kSouth) p.dd = 1.;
            else if (p.mode == kEquatorial) {
                p.rq = std::sqrt(.5 * p.qp);
                p.dd = 1. / p.rq;
                p.xmf = 1.;
                p.ymf = .5 * p.qp;
            } else {
                p.rq = std::sqrt(.5 * p.qp);
                const double sinphi = std::sin(p.phi0);
                p.sinb1 = qsfn(sinphi, p.e, one_es) / p.qp;
                p.cosb1 = std::sqrt(1. - p.sinb1 * p.sinb1);
                p.dd = std::cos(p.phi0) / (std::sqrt(1. - p.es * sinphi * sinphi) * p.rq * p.cosb1);
                p.ymf = (p.xmf = p.rq) / p.dd;
                p.xmf *= p.dd;
            }
        } else if (p.mode == kOblique) {
            p.sinb1 = std::sin(p.phi0);
            p.cosb1 = std::cos(p.phi0);
        }
    } else if (name == "ob_tran") {
        p.kind = kObTran;
        if (p.toMeter != 1) throw Error("+units / +to_meter with ob_tran are not implemented: " + proj4);
        if (p.es != 0) throw Error("ob_tran is implemented on the sphere only: " + proj4);
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
    if (p.kind == kStere && p.es != 0) {  // PJ_stere.c e_forward
        const double sinlam = sin(lam);
        double coslam = cos(lam), sinphi = sin(phi);
        if (p.mode == kOblique) {
            const double X = 2. * atan(ssfn(phi, sinphi, p.e)) - kHalfPi;
            const double sinX = sin(X), cosX = cos(X);
            const double A = p.akm1 / (p.cosph0 * (1. + p.sinph0 * sinX + p.cosph0 * cosX * coslam));
            py = A * (p.cosph0 * sinX - p.sinph0 * cosX * coslam);
            px = A * cosX;
        } else {
            if (p.mode == kSouth) { phi = -phi; coslam = -coslam; sinphi = -sinphi; }
            px = p.akm1 * tsfn(phi, sinphi, p.e);
            py = -px * coslam;
        }
        px *= sinlam;
    } else if (p.kind == kStere) {
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
        const double rho = (fabs(fabs(phi) - kHalfPi) < kEps10) ? 0.
                           : p.c * (p.es != 0 ? pow(tsfn(phi, sin(phi), p.e), p.n) : pow(tan(kFortPi + .5 * phi), -p.n));
        lam *= p.n;
        px = p.k0 * (rho * sin(lam));
        py = p.k0 * (p.rho0 - rho * cos(lam));
    } else if (p.kind == kMerc) {
        px = p.k0 * lam;
        py = p.es != 0 ? -p.k0 * log(tsfn(phi, sin(phi), p.e)) : p.k0 * log(tan(kFortPi + .5 * phi));
    } else if (p.kind == kTmerc) {  // PJ_tmerc.c (4.x): Gauss-Krueger series on the ellipsoid, closed form on the sphere
        constexpr double FC1 = 1., FC2 = .5, FC3 = .16666666666666666666, FC4 = .08333333333333333333, FC5 = .05,
                         FC6 = .03333333333333333333, FC7 = .02380952380952380952, FC8 = .01785714285714285714;
        const double sinphi = sin(phi), cosphi = cos(phi);
        if (p.es != 0) {
            if (lam < -kHalfPi || lam > kHalfPi) { px = NAN; py = NAN; }
            else {
                double t = fabs(cosphi) > 1e-10 ? sinphi / cosphi : 0.;
                t *= t;
                double al = cosphi * lam;
                const double als = al * al;
                al /= sqrt(1. - p.es * sinphi * sinphi);
                const double n = p.esp * cosphi * cosphi;
                px = p.k0 * al * (FC1 + FC3 * als * (1. - t + n + FC5 * als * (5. + t * (t - 18.) + n * (14. - 58. * t) +
                                  FC7 * als * (61. + t * (t * (179. - t) - 479.)))));
                py = p.k0 * (mlfn(phi, sinphi, cosphi, p.en) - p.ml0 + sinphi * al * lam * FC2 * (1. + FC4 * als * (5. - t + n * (9. + 4. * n) +
                             FC6 * als * (61. + t * (t - 58.) + n * (270. - 330. * t) + FC8 * als * (1385. + t * (t * (543. - t) - 3111.))))));
            }
        } else {
            double b = cosphi * sin(lam);
            if (fabs(fabs(b) - 1.) <= kEps10) { px = NAN; py = NAN; }
            else {
                px = p.ml0 * log((1. + b) / (1. - b));
                py = cosphi * cos(lam) / sqrt(1. - b * b);
                b = fabs(py);
                py = b >= 1. ? 0. : acos(py);
                if (phi < 0.) py = -py;
                py = p.esp * (py - p.phi0);
            }
        }
    } else if (p.kind == kSinu) {
        const double sn = sin(phi), cs = cos(phi);
        if (p.es != 0) { py = mlfn(phi, sn, cs, p.en); px = lam * cs / sqrt(1. - p.es * sn * sn); }
        else { px = lam * cs; py = phi; }
    } else if (p.kind == kCea) {
        px = p.k0 * lam;
        py = p.es != 0 ? .5 * qsfn(sin(phi), p.e, 1. - p.es) / p.k0 : sin(phi) / p.k0;
    } else if (p.kind == kOrtho) {  // PJ_ortho.c s_forward
        const double cosphi = cos(phi), sinphi = sin(phi);
        double coslam = cos(lam);
        bool bad;
        if (p.mode == kEquatorial) { bad = cosphi * coslam < -kEps10; py = sinphi; }
        else if (p.mode == kOblique) {
            bad = p.sinph0 * sinphi + p.cosph0 * cosphi * coslam < -kEps10;
            py = p.cosph0 * sinphi - p.sinph0 * cosphi * coslam;
        } else {
            if (p.mode == kNorth) coslam = -coslam;
            bad = fabs(phi - p.phi0) - kEps10 > kHalfPi;
            py = cosphi * coslam;
        }
        px = cosphi * sin(lam);
        if (bad) { px = NAN; py = NAN; }
    } else if (p.kind == kAeqd) {  // PJ_aeqd.c s_forward
        const double sinphi = sin(phi), cosphi = cos(phi);
        double coslam = cos(lam);
        if (p.mode == kEquatorial || p.mode == kOblique) {
            py = p.mode == kEquatorial ? cosphi * coslam : p.sinph0 * sinphi + p.cosph0 * cosphi * coslam;
            if (fabs(fabs(py) - 1.) < 1e-14) {
                if (py < 0.) { px = NAN; py = NAN; }
                else { px = 0.; py = 0.; }
            } else {
                py = acos(py);
                py /= sin(py);
                px = py * cosphi * sin(lam);
                py *= p.mode == kEquatorial ? sinphi : p.cosph0 * sinphi - p.sinph0 * cosphi * coslam;
            }
        } else {
            if (p.mode == kNorth) { phi = -phi; coslam = -coslam; }
            if (fabs(phi - kHalfPi) < kEps10) { px = NAN; py = NAN; }
            else {
                py = kHalfPi + phi;
                px = py * sin(lam);
                py *= coslam;
            }
        }
    } else if (p.kind == kNsper) {  // PJ_nsper.c s_forward (no tilt)
        const double sinphi = sin(phi), cosphi = cos(phi), coslam = cos(lam);
        py = p.mode == kOblique ? p.sinph0 * sinphi + p.cosph0 * cosphi * coslam
             : (p.mode == kEquatorial ? cosphi * coslam : (p.mode == kSouth ? -sinphi : sinphi));
        if (py < p.rp) { px = NAN; py = NAN; }  // beyond the horizon
        else {
            py = p.pn1 / (p.pp - py);
            px = py * cosphi * sin(lam);
            if (p.mode == kOblique) py *= p.cosph0 * sinphi - p.sinph0 * cosphi * coslam;
            else if (p.mode == kEquatorial) py *= sinphi;
            else py *= cosphi * (p.mode == kNorth ? -coslam : coslam);
        }
    } else if (p.kind == kOmerc) {  // PJ_omerc.c e_forward
        double u, v;
        bool ok = true;
        if (fabs(fabs(phi) - kHalfPi) > kEps10) {
            const double Q = p.oE / pow(tsfn(phi, sin(phi), p.e), p.oB);
            const double temp = 1. / Q, S = .5 * (Q - temp), T = .5 * (Q + temp);
            const double V = sin(p.oB * lam), U = (S * p.singam - V * p.cosgam) / T;
            if (fabs(fabs(U) - 1.0) < kEps10) ok = false;
            v = 0.5 * p.ArB * log((1. - U) / (1. + U));
            u = p.ArB * atan2((S * p.cosgam + V * p.singam), cos(p.oB * lam));
        } else {
            v = phi > 0 ? p.v_pole_n : p.v_pole_s;
            u = p.ArB * phi;
        }
        if (!ok) { px = NAN; py = NAN; }
        else if (p.mode) { px = u; py = v; }
        else {
            u -= p.u_0;
            px = v * p.cosrot + u * p.sinrot;
            py = u * p.cosrot - v * p.sinrot;
        }
    } else if (p.kind == kGeos) {  // PJ_geos.c e_forward (the spherical form is this one with radius_p = 1)
        phi = atan(p.radius_p2 * tan(phi));
        const double r = p.radius_p / hypot(p.radius_p * cos(phi), sin(phi));
        const double Vx = r * cos(lam) * cos(phi), Vy = r * sin(lam) * cos(phi), Vz = r * sin(phi);
        if (((p.radius_g - Vx) * Vx - Vy * Vy - Vz * Vz * p.radius_p_inv2) < 0.) { px = NAN; py = NAN; }  // behind the limb
        else {
            const double tmp = p.radius_g - Vx;
            if (p.mode) { px = p.radius_g_1 * atan(Vy / hypot(Vz, tmp)); py = p.radius_g_1 * atan(Vz / tmp); }
            else { px = p.radius_g_1 * atan(Vy / tmp); py = p.radius_g_1 * atan(Vz / hypot(Vy, tmp)); }
        }
    } else if (p.kind == kAea) {  // PJ_aea.c e_forward
        double rho = p.c - (p.es != 0 ? p.n * qsfn(sin(phi), p.e, 1. - p.es) : p.n2 * sin(phi));
        if (rho < 0.) { px = NAN; py = NAN; }
        else {
            rho = p.dd * sqrt(rho);
            lam *= p.n;
            px = rho * sin(lam);
            py = p.rho0 - rho * cos(lam);
        }
    } else if (p.kind == kLaea) {  // PJ_laea.c e_forward / s_forward
        double coslam = cos(lam);
        const double sinlam = sin(lam), sinphi = sin(phi);
        if (p.es != 0) {
            double q = qsfn(sinphi, p.e, 1. - p.es), b;
            if (p.mode == kOblique || p.mode == kEquatorial) {
                const double sinb = q / p.qp, cosb = sqrt(1. - sinb * sinb);
                if (p.mode == kOblique) {
                    b = 1. + p.sinb1 * sinb + p.cosb1 * cosb * coslam;
                    if (fabs(b) < kEps10) { px = NAN; py = NAN; }
                    else {
                        b = sqrt(2. / b);
                        py = p.ymf * b * (p.cosb1 * sinb - p.sinb1 * cosb * coslam);
                        px = p.xmf * b * cosb * sinlam;
                    }
                } else {
                    b = 1. + cosb * coslam;
                    if (fabs(b) < kEps10) { px = NAN; py = NAN; }
                    else {
                        b = sqrt(2. / b);
                        py = b * sinb * p.ymf;
                        px = p.xmf * b * cosb * sinlam;
                    }
                }
            } else {
                if (p.mode == kNorth) { b = kHalfPi + phi; q = p.qp - q; }
                else { b = phi - kHalfPi; q = p.qp + q; }
                if (fabs(b) < kEps10) { px = NAN; py = NAN; }
                else if (q >= 0.) {
                    b = sqrt(q);
                    px = b * sinlam;
                    py = coslam * (p.mode == kSouth ? b : -b);
                } else { px = 0.; py = 0.; }
            }
        } else {
            const double cosphi = cos(phi);
            if (p.mode == kEquatorial || p.mode == kOblique) {
                py = p.mode == kEquatorial ? 1. + cosphi * coslam : 1. + p.sinb1 * sinphi + p.cosb1 * cosphi * coslam;
                if (py <= kEps10) { px = NAN; py = NAN; }
                else {
                    py = sqrt(2. / py);
                    px = py * cosphi * sinlam;
                    py *= p.mode == kEquatorial ? sinphi : p.cosb1 * sinphi - p.sinb1 * cosphi * coslam;
                }
            } else {
                if (p.mode == kNorth) coslam = -coslam;
                if (fabs(phi + p.phi0) < kEps10) { px = NAN; py = NAN; }
                else {
                    py = kFortPi - phi * .5;
                    py = 2. * (p.mode == kSouth ? cos(py) : sin(py));
                    px = py * sinlam;
                    py *= coslam;
                }
            }
        }
    } else if (p.kind == kEtmerc) {  // PJ_etmerc.c e_forward
        double Cn = gatg(p.cbg, phi), Ce = lam;                     // geodetic -> Gaussian latitude
        const double sinCn = sin(Cn), cosCn = cos(Cn), sinCe = sin(Ce), cosCe = cos(Ce);
        Cn = atan2(sinCn, cosCe * cosCn);                           // -> complementary spherical
        Ce = atan2(sinCe * cosCn, hypot(sinCn, cosCn * cosCe));
        Ce = asinh(tan(Ce));
        double dCn, dCe;
        clenS(p.gtu, 2 * Cn, 2 * Ce, dCn, dCe);                     // -> normalised ellipsoidal N, E
        Cn += dCn;
        Ce += dCe;
        if (fabs(Ce) <= 2.623395162778) { py = p.Qn * Cn + p.Zb; px = p.Qn * Ce; }
        else { px = NAN; py = NAN; }
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
    x = p.frMeter * (p.a * px + p.x0);  // pj_fwd.c
    y = p.frMeter * (p.a * py + p.y0);
}

// projected -> geographic (rad)
__device__ void proj_inverse(const ProjParams& p, double x, double y, double& lon, double& lat)
{
    if (p.kind == kLatLong) { lon = x; lat = y; return; }
    double xs, ys;
    if (p.kind == kObTran) { xs = x - p.x0; ys = y - p.y0; }
    else { xs = (x * p.toMeter - p.x0) / p.a; ys = (y * p.toMeter - p.y0) / p.a; }  // pj_inv.c
    double lam = 0, phi = 0;
    if (p.kind == kStere && p.es != 0) {  // PJ_stere.c e_inverse
        const double rho = hypot(xs, ys);
        double tp, phi_l, halfpi, halfe;
        if (p.mode == kOblique) {
            tp = 2. * atan2(rho * p.cosph0, p.akm1);
            const double cosphi = cos(tp), sinphi = sin(tp);
            phi_l = rho == 0. ? asin(cosphi * p.sinph0) : asin(cosphi * p.sinph0 + (ys * sinphi * p.cosph0 / rho));
            tp = tan(.5 * (kHalfPi + phi_l));
            xs *= sinphi;
            ys = rho * p.cosph0 * cosphi - ys * p.sinph0 * sinphi;
            halfpi = kHalfPi;
            halfe = .5 * p.e;
        } else {
            if (p.mode == kNorth) ys = -ys;
            phi_l = kHalfPi - 2. * atan(tp = -rho / p.akm1);
            halfpi = -kHalfPi;
            halfe = -.5 * p.e;
        }
        phi = NAN;  // no convergence in 8 rounds: pj_transform reports an error
        lam = NAN;
        for (int i = 8; i--; phi_l = phi) {
            const double sinphi = p.e * sin(phi_l);
            phi = 2. * atan(tp * pow((1. + sinphi) / (1. - sinphi), halfe)) - halfpi;
            if (fabs(phi_l - phi) < 1e-10) {
                if (p.mode == kSouth) phi = -phi;
                lam = (xs == 0. && ys == 0.) ? 0. : atan2(xs, ys);
                break;
            }
            if (i == 0) phi = NAN;
        }
    } else if (p.kind == kStere) {
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
            phi = p.es != 0 ? phi2(pow(rho / p.c, 1. / p.n), p.e) : 2. * atan(pow(p.c / rho, 1. / p.n)) - kHalfPi;
            lam = atan2(xs, ys) / p.n;
        } else {
            lam = 0.;
            phi = p.n > 0. ? kHalfPi : -kHalfPi;
        }
    } else if (p.kind == kMerc) {
        lam = xs / p.k0;
        phi = p.es != 0 ? phi2(exp(-ys / p.k0), p.e) : kHalfPi - 2. * atan(exp(-ys / p.k0));
    } else if (p.kind == kTmerc) {
        constexpr double FC1 = 1., FC2 = .5, FC3 = .16666666666666666666, FC4 = .08333333333333333333, FC5 = .05,
                         FC6 = .03333333333333333333, FC7 = .02380952380952380952, FC8 = .01785714285714285714;
        if (p.es != 0) {
            phi = inv_mlfn(p.ml0 + ys / p.k0, p.es, p.en);
            if (fabs(phi) >= kHalfPi) {
                phi = ys < 0. ? -kHalfPi : kHalfPi;
                lam = 0.;
            } else {
                const double sinphi = sin(phi), cosphi = cos(phi);
                double t = fabs(cosphi) > 1e-10 ? sinphi / cosphi : 0.;
                const double n = p.esp * cosphi * cosphi;
                double con = 1. - p.es * sinphi * sinphi;
                const double d = xs * sqrt(con) / p.k0;
                con *= t;
                t *= t;
                const double ds = d * d;
                phi -= (con * ds / (1. - p.es)) * FC2 * (1. - ds * FC4 * (5. + t * (3. - 9. * n) + n * (1. - 4 * n) -
                        ds * FC6 * (61. + t * (90. - 252. * n + 45. * t) + 46. * n - ds * FC8 * (1385. + t * (3633. + t * (4095. + 1574. * t))))));
                lam = d * (FC1 - ds * FC3 * (1. + 2. * t + n - ds * FC5 * (5. + t * (28. + 24. * t + 8. * n) + 6. * n -
                           ds * FC7 * (61. + t * (662. + t * (1320. + 720. * t)))))) / cosphi;
            }
        } else {
            double h = exp(xs / p.esp);
            const double g = .5 * (h - 1. / h);
            h = cos(p.phi0 + ys / p.esp);
            phi = asin(sqrt((1. - h * h) / (1. + g * g)));
            if (ys < 0. && -phi + p.phi0 < 0.) phi = -phi;  // the hemisphere test of PROJ 4.9 (4.8 and older: y < 0 alone, wrong for lat_0 != 0)
            lam = (g != 0. || h != 0.) ? atan2(g, h) : 0.;
        }
    } else if (p.kind == kSinu) {
        if (p.es != 0) {
            phi = inv_mlfn(ys, p.es, p.en);
            const double ab = fabs(phi);
            if (ab < kHalfPi) {
                const double sn = sin(phi);
                lam = xs * sqrt(1. - p.es * sn * sn) / cos(phi);
            } else if (ab - kEps10 < kHalfPi) lam = 0.;
            else { lam = NAN; phi = NAN; }
        } else {
            phi = ys;
            lam = xs / cos(ys);
        }
    } else if (p.kind == kCea) {
        lam = xs / p.k0;
        if (p.es != 0) {
            double q = 2. * ys * p.k0 / p.qp;
            q = q > 1 ? 1 : (q < -1 ? -1 : q);
            const double beta = asin(q), t = beta + beta;
            phi = beta + p.apa[0] * sin(t) + p.apa[1] * sin(t + t) + p.apa[2] * sin(t + t + t);
        } else {
            ys *= p.k0;
            const double t = fabs(ys);
            if (t - kEps10 <= 1.) phi = t >= 1. ? (ys < 0. ? -kHalfPi : kHalfPi) : asin(ys);
            else { phi = NAN; lam = NAN; }
        }
    } else if (p.kind == kOrtho) {  // PJ_ortho.c s_inverse
        const double rh = hypot(xs, ys);
        double sinc = rh;
        if (sinc - 1. > kEps10) { phi = NAN; lam = NAN; }
        else {
            if (sinc > 1.) sinc = 1.;
            const double cosc = sqrt(1. - sinc * sinc);
            if (fabs(rh) <= kEps10) { phi = p.phi0; lam = 0.; }
            else {
                if (p.mode == kNorth) { ys = -ys; phi = acos(sinc); }
                else if (p.mode == kSouth) phi = -acos(sinc);
                else {
                    if (p.mode == kEquatorial) { phi = ys * sinc / rh; xs *= sinc; ys = cosc * rh; }
                    else {
                        phi = cosc * p.sinph0 + ys * sinc * p.cosph0 / rh;
                        ys = (cosc - p.sinph0 * phi) * rh;
                        xs *= sinc * p.cosph0;
                    }
                    phi = fabs(phi) >= 1. ? (phi < 0. ? -kHalfPi : kHalfPi) : asin(phi);
                }
                lam = (ys == 0. && (p.mode == kOblique || p.mode == kEquatorial)) ? (xs == 0. ? 0. : (xs < 0. ? -kHalfPi : kHalfPi)) : atan2(xs, ys);
            }
        }
    } else if (p.kind == kAeqd) {  // PJ_aeqd.c s_inverse
        double c_rh = hypot(xs, ys);
        if (c_rh - kEps10 > kPi) { phi = NAN; lam = NAN; }
        else if (c_rh < kEps10) { phi = p.phi0; lam = 0.; }
        else {
            if (c_rh > kPi) c_rh = kPi;
            if (p.mode == kOblique || p.mode == kEquatorial) {
                const double sinc = sin(c_rh), cosc = cos(c_rh);
                double arg;
                if (p.mode == kEquatorial) { arg = ys * sinc / c_rh; xs *= sinc; ys = cosc * c_rh; }
                else {
                    arg = cosc * p.sinph0 + ys * sinc * p.cosph0 / c_rh;
                    arg = arg > 1 ? 1 : (arg < -1 ? -1 : arg);
                    ys = (cosc - p.sinph0 * arg) * c_rh;  // sin(asin(arg))
                    xs *= sinc * p.cosph0;
                }
                arg = arg > 1 ? 1 : (arg < -1 ? -1 : arg);
                phi = asin(arg);
                lam = ys == 0. ? 0. : atan2(xs, ys);
            } else if (p.mode == kNorth) { phi = kHalfPi - c_rh; lam = atan2(xs, -ys); }
            else { phi = c_rh - kHalfPi; lam = atan2(xs, ys); }
        }
    } else if (p.kind == kNsper) {  // PJ_nsper.c s_inverse
        const double rh = hypot(xs, ys);
        double sinz = 1. - rh * rh * p.pfact;
        if (sinz < 0.) { phi = NAN; lam = NAN; }
        else if (fabs(rh) <= kEps10) { lam = 0.; phi = p.phi0; }
        else {
            sinz = (p.pp - sqrt(sinz)) / (p.pn1 / rh + rh / p.pn1);
            const double cosz = sqrt(1. - sinz * sinz);
            if (p.mode == kOblique) {
                phi = asin(cosz * p.sinph0 + ys * sinz * p.cosph0 / rh);
                ys = (cosz - p.sinph0 * sin(phi)) * rh;
                xs *= sinz * p.cosph0;
            } else if (p.mode == kEquatorial) {
                phi = asin(ys * sinz / rh);
                ys = cosz * rh;
                xs *= sinz;
            } else if (p.mode == kNorth) { phi = asin(cosz); ys = -ys; }
            else phi = -asin(cosz);
            lam = atan2(xs, ys);
        }
    } else if (p.kind == kOmerc) {  // PJ_omerc.c e_inverse
        double u, v;
        if (p.mode) { v = ys; u = xs; }
        else {
            v = xs * p.cosrot - ys * p.sinrot;
            u = ys * p.cosrot + xs * p.sinrot + p.u_0;
        }
        const double Qp = exp(-p.BrA * v), Sp = .5 * (Qp - 1. / Qp), Tp = .5 * (Qp + 1. / Qp);
        const double Vp = sin(p.BrA * u), Up = (Vp * p.cosgam + Sp * p.singam) / Tp;
        if (fabs(fabs(Up) - 1.) < kEps10) {
            lam = 0.;
            phi = Up < 0. ? -kHalfPi : kHalfPi;
        } else {
            phi = p.oE / sqrt((1. + Up) / (1. - Up));
            phi = phi2(pow(phi, 1. / p.oB), p.e);
            lam = -p.rB * atan2((Sp * p.cosgam - Vp * p.singam), cos(p.BrA * u));
        }
    } else if (p.kind == kGeos) {  // PJ_geos.c e_inverse
        double Vx = -1.0, Vy, Vz;
        if (p.mode) { Vz = tan(ys / p.radius_g_1); Vy = tan(xs / p.radius_g_1) * hypot(1.0, Vz); }
        else { Vy = tan(xs / p.radius_g_1); Vz = tan(ys / p.radius_g_1) * hypot(1.0, Vy); }
        double a = Vz / p.radius_p;
        a = Vy * Vy + a * a + Vx * Vx;
        const double b = 2 * p.radius_g * Vx;
        const double det = (b * b) - 4 * a * p.C;
        if (det < 0.) { lam = NAN; phi = NAN; }  // off the disc
        else {
            const double k = (-b - sqrt(det)) / (2. * a);
            Vx = p.radius_g + k * Vx;
            Vy *= k;
            Vz *= k;
            lam = atan2(Vy, Vx);
            phi = atan(Vz * cos(lam) / Vx);
            phi = atan(p.radius_p_inv2 * tan(phi));
        }
    } else if (p.kind == kAea) {  // PJ_aea.c e_inverse
        ys = p.rho0 - ys;
        double rho = hypot(xs, ys);
        if (rho != 0.) {
            if (p.n < 0.) { rho = -rho; xs = -xs; ys = -ys; }
            phi = rho / p.dd;
            if (p.es != 0) {
                phi = (p.c - phi * phi) / p.n;
                if (fabs(p.ec - fabs(phi)) > 1e-7) {  // pj_phi1_: Newton on the authalic q
                    const double qs = phi, one_es = 1. - p.es;
                    double Phi = asin(.5 * qs), dphi;
                    int i = 15;
                    do {
                        const double sinpi = sin(Phi), cospi = cos(Phi), con = p.e * sinpi, com = 1. - con * con;
                        dphi = .5 * com * com / cospi * (qs / one_es - sinpi / com + .5 / p.e * log((1. - con) / (1. + con)));
                        Phi += dphi;
                    } while (fabs(dphi) > 1e-10 && --i);
                    phi = i ? Phi : NAN;
                } else phi = phi < 0. ? -kHalfPi : kHalfPi;
            } else {
                phi = (p.c - phi * phi) / p.n2;
                phi = fabs(phi) <= 1. ? asin(phi) : (phi < 0. ? -kHalfPi : kHalfPi);
            }
            lam = atan2(xs, ys) / p.n;
        } else {
            lam = 0.;
            phi = p.n > 0. ? kHalfPi : -kHalfPi;
        }
    } else if (p.kind == kLaea) {  // PJ_laea.c e_inverse / s_inverse
        if (p.es != 0) {
            double ab = 0;
            bool centre = false;
            if (p.mode == kEquatorial || p.mode == kOblique) {
                xs /= p.dd;
                ys *= p.dd;
                const double rho = hypot(xs, ys);
                if (rho < kEps10) centre = true;
                else {
                    double sCe = 2. * asin(.5 * rho / p.rq);
                    const double cCe = cos(sCe);
                    sCe = sin(sCe);
                    xs *= sCe;
                    if (p.mode == kOblique) {
                        ab = cCe * p.sinb1 + ys * sCe * p.cosb1 / rho;
                        ys = rho * p.cosb1 * cCe - ys * p.sinb1 * sCe;
                    } else {
                        ab = ys * sCe / rho;
                        ys = rho * cCe;
                    }
                }
            } else {
                if (p.mode == kNorth) ys = -ys;
                const double q = xs * xs + ys * ys;
                if (q == 0.) centre = true;
                else {
                    ab = 1. - q / p.qp;
                    if (p.mode == kSouth) ab = -ab;
                }
            }
            if (centre) { lam = 0.; phi = p.phi0; }
            else {
                lam = atan2(xs, ys);
                const double beta = asin(ab), t = beta + beta;  // pj_authlat
                phi = beta + p.apa[0] * sin(t) + p.apa[1] * sin(t + t) + p.apa[2] * sin(t + t + t);
            }
        } else {
            const double rh = hypot(xs, ys);
            phi = rh * .5;
            if (phi > 1.) { phi = NAN; lam = NAN; }
            else {
                phi = 2. * asin(phi);
                const double sinz = sin(phi), cosz = cos(phi);
                if (p.mode == kEquatorial) {
                    phi = fabs(rh) <= kEps10 ? 0. : asin(ys * sinz / rh);
                    xs *= sinz;
                    ys = cosz * rh;
                } else if (p.mode == kOblique) {
                    phi = fabs(rh) <= kEps10 ? p.phi0 : asin(cosz * p.sinb1 + ys * sinz * p.cosb1 / rh);
                    xs *= sinz * p.cosb1;
                    ys = (cosz - sin(phi) * p.sinb1) * rh;
                } else if (p.mode == kNorth) {
                    ys = -ys;
                    phi = kHalfPi - phi;
                } else {
                    phi -= kHalfPi;
                }
                lam = (ys == 0. && (p.mode == kEquatorial || p.mode == kOblique)) ? 0. : atan2(xs, ys);
            }
        }
    } else if (p.kind == kEtmerc) {  // PJ_etmerc.c e_inverse
        double Cn = (ys - p.Zb) / p.Qn, Ce = xs / p.Qn;
        if (fabs(Ce) <= 2.623395162778) {
            double dCn, dCe;
            clenS(p.utg, 2 * Cn, 2 * Ce, dCn, dCe);
            Cn += dCn;
            Ce += dCe;
            Ce = atan(sinh(Ce));
            const double sinCn = sin(Cn), cosCn = cos(Cn), sinCe = sin(Ce), cosCe = cos(Ce);
            Ce = atan2(sinCe, cosCe * cosCn);
            Cn = atan2(sinCn * cosCe, hypot(sinCe, cosCe * cosCn));
            phi = gatg(p.cgb, Cn);
            lam = Ce;
        } else { phi = NAN; lam = NAN; }
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

// pj_datum_transform applies when both sides name a datum and pj_compare_datums finds them different
struct ProjPair { ProjParams src, dst; };
ProjPair parse_pair(const char* projIn, const char* projOut)
{
    ProjPair pp{parse_proj4(projIn), parse_proj4(projOut)};
    const ProjParams &a = pp.src, &b = pp.dst;
    if (a.datumType != 0 && b.datumType != 0) {
        bool same = a.datumType == b.datumType && a.a == b.a && std::fabs(a.es - b.es) <= 0.000000000050;
        const int nPar = a.datumType == 1 ? 3 : (a.datumType == 2 ? 7 : 0);
        for (int i = 0; i < nPar; ++i) same = same && a.towgs84[i] == b.towgs84[i];
        const bool params = a.datumType == 1 || a.datumType == 2 || b.datumType == 1 || b.datumType == 2;
        pp.src.doShift = pp.dst.doShift = !same && (a.es != b.es || a.a != b.a || params);  // either side may be the source of a call
    }
    return pp;
}

// geodetic (height 0) -> geocentric -> WGS84 -> the other datum -> geodetic (pj_datum_transform, pj_geocentric_to_wgs84 /
// _from_wgs84, geocent.c's iteration: at most 30 rounds, 1e-12)
__device__ void datum_shift(const ProjParams& src, const ProjParams& dst, double& lon, double& lat)
{
    const double sinlat = sin(lat), coslat = cos(lat);
    const double Rn = src.a / sqrt(1.0 - src.es * sinlat * sinlat);
    double X = Rn * coslat * cos(lon), Y = Rn * coslat * sin(lon), Z = (Rn * (1 - src.es)) * sinlat;
    const double* v = src.towgs84;
    if (src.datumType == 1) { X += v[0]; Y += v[1]; Z += v[2]; }
    else if (src.datumType == 2) {
        const double xo = v[6] * (X - v[5] * Y + v[4] * Z) + v[0], yo = v[6] * (v[5] * X + Y - v[3] * Z) + v[1],
                     zo = v[6] * (-v[4] * X + v[3] * Y + Z) + v[2];
        X = xo; Y = yo; Z = zo;
    }
    v = dst.towgs84;
    if (dst.datumType == 1) { X -= v[0]; Y -= v[1]; Z -= v[2]; }
    else if (dst.datumType == 2) {
        const double xt = (X - v[0]) / v[6], yt = (Y - v[1]) / v[6], zt = (Z - v[2]) / v[6];
        X = xt + v[5] * yt - v[4] * zt;
        Y = -v[5] * xt + yt + v[3] * zt;
        Z = v[4] * xt - v[3] * yt + zt;
    }
    // pj_Convert_Geocentric_To_Geodetic
    const double a = dst.a, es = dst.es, P = sqrt(X * X + Y * Y), RR = sqrt(X * X + Y * Y + Z * Z);
    if (P / a < 1.E-12) {
        lon = 0.;
        if (RR / a < 1.E-12) { lat = kHalfPi; return; }
    } else {
        lon = atan2(Y, X);
    }
    const double CT = Z / RR, ST = P / RR;
    double RX = 1.0 / sqrt(1.0 - es * (2.0 - es) * ST * ST);
    double CPHI0 = ST * (1.0 - es) * RX, SPHI0 = CT * RX, CPHI, SPHI, SDPHI;
    int iter = 0;
    do {
        ++iter;
        const double RN = a / sqrt(1.0 - es * SPHI0 * SPHI0);
        const double Height = P * CPHI0 + Z * SPHI0 - RN * (1.0 - es * SPHI0 * SPHI0);
        const double RK = es * RN / (RN + Height);
        RX = 1.0 / sqrt(1.0 - RK * (2.0 - RK) * ST * ST);
        CPHI = ST * (1.0 - RK) * RX;
        SPHI = CT * RX;
        SDPHI = SPHI * CPHI0 - CPHI * SPHI0;
        CPHI0 = CPHI;
        SPHI0 = SPHI;
    } while (SDPHI * SDPHI > 1.E-24 && iter < 30);
    lat = atan(SPHI / fabs(CPHI));
}

// pj_transform(src, dst) on one point
__device__ __forceinline__ void transform_point(const ProjParams& src, const ProjParams& dst, double& x, double& y)
{
    double lon, lat;
    proj_inverse(src, x, y, lon, lat);
    lon += src.fromGreenwich;  // pj_transform.c: longitudes meet relative to Greenwich
    if (src.doShift) datum_shift(src, dst, lon, lat);
    lon -= dst.fromGreenwich;
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
    const ProjPair pair = parse_pair(projIn, projOut);
    const ProjParams &src = pair.src, &dst = pair.dst;
    if (n == 0) return;
    project_values_kernel<<<point_blocks(n), kBlock, 0, stream>>>(src, dst, d_x, d_y, n);
    FA_HIP(hipGetLastError());
}

void launch_project_axes(const char* projIn, const char* projOut, const double* h_xAxis, const double* h_yAxis, size_t ix, size_t iy,
                         double* d_outX, double* d_outY, hipStream_t stream)
{
    const ProjPair pair = parse_pair(projIn, projOut);
    const ProjParams &src = pair.src, &dst = pair.dst;
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
    const ProjPair pair = parse_pair(projIn, projOut);
    const ProjParams &in = pair.src, &out = pair.dst;
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
    const ProjPair pair = parse_pair(projIn, projOut);
    const ProjParams &in = pair.src, &out = pair.dst;
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
    const ProjPair pair = parse_pair(projIn, projOut);
    const ProjParams &in = pair.src, &out = pair.dst;
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
