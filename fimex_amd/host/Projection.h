// Map projections on spheres from their published closed forms (Snyder, USGS PP 1395) -- the HOST copy.
// Plan building itself runs on the device (fimex_amd/csrc/projection.hip behind fimex_amd_project_axes_* etc., which
// CDMInterpolator.cc calls); this copy serves host_cli's --project / --matrix modes, the GPU-less cross-check of the same
// formulas in tests/test_host_projection.py.
//
// The reference obtains these from the third-party library PROJ.4 through pj_init_plus / pj_transform
// (call sites src/interpolation.c:355,396,644,700,773,1185,1233).  PROJ.4 is not part of the reference tree and
// is not available here; this is an independent implementation of the same projections with PROJ.4's
// conventions at the pj_transform boundary: geographic coordinates in radians, projected coordinates
// x = a * x' + x_0, longitudes relative to lon_0 wrapped to [-pi, pi].  Ellipsoids are not implemented
// (every projection string in the reference's tests for this path is spherical except the UTM case).
#pragma once

#include <map>
#include <string>
#include <vector>

namespace FimexAmd {

class Projection {
public:
    // "+proj=stere +lat_0=90 +lon_0=0 +lat_ts=60 +R=6.371e6" ...; throws CDMException for unsupported strings
    explicit Projection(const std::string& proj4);

    bool isLatLong() const { return kind_ == Kind::LatLong; }
    // projected coordinates are degrees-like (radians at this boundary): latlong and rotated lat/lon
    bool isDegree() const { return kind_ == Kind::LatLong || kind_ == Kind::ObTran; }

    void forward(double lon, double lat, double& x, double& y) const;  // geographic rad -> projected
    void inverse(double x, double y, double& lon, double& lat) const;  // projected -> geographic rad

    const std::string& proj4() const { return proj4_; }

private:
    enum class Kind { LatLong, Stere, Lcc, Merc, ObTran };
    enum class StereMode { North, South, Oblique, Equatorial };
    std::string proj4_;
    std::map<std::string, std::string> par_;
    Kind kind_ = Kind::LatLong;
    double a_ = 1, lam0_ = 0, phi0_ = 0, x0_ = 0, y0_ = 0, k0_ = 1;
    // stere
    StereMode mode_ = StereMode::North;
    double akm1_ = 0, sinph0_ = 0, cosph0_ = 0;
    // lcc
    double n_ = 0, c_ = 0, rho0_ = 0;
    // ob_tran
    double lamp_ = 0, sphip_ = 0, cphip_ = 0;
    bool oblique_ = false;

    bool has(const std::string& k) const { return par_.count(k) != 0; }
    double num(const std::string& k, double dflt) const;
    double rad(const std::string& k, double dflt) const;
};

// pj_transform(src, dst, ...) in place on n points (no datum shift)
void transform(const Projection& src, const Projection& dst, double* x, double* y, size_t n);

// mifi_project_axes (src/interpolation.c:1199-1244): the [iy][ix] mesh of two axes, transformed
void projectAxes(const Projection& in, const Projection& out, const std::vector<double>& xAxis, const std::vector<double>& yAxis,
                 std::vector<double>& outX, std::vector<double>& outY);

// mifi_get_vector_reproject_matrix (src/interpolation.c:719-788 with :441-521 and :330-438):
// matrix[4*i] = (cos, sin, -sin, phi) of the local rotation from the input to the output projection at every
// point of the output mesh.  Axes in projection units (radians for degree axes).
void vectorReprojectMatrix(const Projection& in, const Projection& out, const std::vector<double>& outXAxis,
                           const std::vector<double>& outYAxis, std::vector<double>& matrix);

}  // namespace FimexAmd
