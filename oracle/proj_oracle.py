"""numpy restatement of the cartographic projections the reference obtains from PROJ.4.

TEST INFRASTRUCTURE ONLY (see oracle/fimex_oracle.h).

The reference calls the third-party library PROJ.4 through its legacy API
(proj_api.h: pj_init_plus / pj_transform; call sites src/interpolation.c:355,396,
644,700,773,1185,1233).  The library is not vendored in /root/reference and its
version is not pinned there (README.md:16 "proj-4 >= 4.4.9"; NEWS:57 mentions
4.8.0), and it is not installed in this image.  What follows restates the
published spherical formulas PROJ.4 implements (Snyder, "Map Projections - A
Working Manual", USGS PP 1395: stereographic eq. 21-2..21-4, 20-14..20-18;
Lambert conformal conic eq. 15-1..15-5; oblique transformation eq. 5-7..5-10b)
with PROJ.4's conventions: longitude/latitude in radians at the pj_transform
boundary, x = a*x' + x_0, lam = lon - lon_0 wrapped to [-pi, pi].
Only spheres are covered (+R, or +a with +e=0 / +ellps=sphere): every projection
string in the reference's tests for this path is spherical except the UTM case
of test/testInterpolator.cc:430-432, which is out of scope here.

Pins (tests/test_oracle_kats.py): tests/golden/coordTest.nc stores 2-D
longitude/latitude for its 11x11 polar-stereographic grid (121 points);
tests/golden/outData.txt holds the 180x90 nearest-neighbour output of the EMEP
polar-stereographic -> lat/lon chain of test/testInterpolation.cc:280-347.
Beyond those fixtures: parity unpinned.
"""
import math

import numpy as np

HALFPI = math.pi / 2
FORTPI = math.pi / 4
_SPI = 3.14159265359  # PROJ.4 adjlon.c threshold
_EPS10 = 1e-10


def parse(projstr):
    """'+proj=stere +lat_0=90 ...' -> dict (flags map to True)."""
    out = {}
    for tok in projstr.split():
        tok = tok.lstrip("+")
        if not tok:
            continue
        if "=" in tok:
            k, v = tok.split("=", 1)
            out[k] = v
        else:
            out[tok] = True
    return out


def _rad(p, key, default=0.0):
    return math.radians(float(p[key])) if key in p else default


def _radius(p):
    if "R" in p:
        return float(p["R"])
    if "a" in p:
        e = float(p.get("e", 0.0)) if "e" in p else 0.0
        if p.get("ellps", "sphere") == "sphere" and e == 0.0 and "b" not in p and "rf" not in p and "f" not in p:
            return float(p["a"])
        raise NotImplementedError("ellipsoid (only spheres are restated): %r" % (p,))
    if p.get("ellps") == "sphere":
        return 6370997.0
    if p["proj"] in ("latlong", "longlat", "latlon", "lonlat"):
        return 1.0  # radius is irrelevant for geographic coordinates without datum shift
    raise NotImplementedError("ellipsoid (only spheres are restated): %r" % (p,))


def adjlon(lon):
    lon = np.asarray(lon, dtype=np.float64)
    wrapped = lon + math.pi
    wrapped = wrapped - 2 * math.pi * np.floor(wrapped / (2 * math.pi)) - math.pi
    return np.where(np.abs(lon) <= _SPI, lon, wrapped)


def is_latlong(p):
    return p["proj"] in ("latlong", "longlat", "latlon", "lonlat")


class _Proj:
    def __init__(self, projstr):
        self.p = parse(projstr)
        self.name = self.p["proj"]
        self.a = _radius(self.p)
        self.lam0 = _rad(self.p, "lon_0")
        self.phi0 = _rad(self.p, "lat_0")
        self.x0 = float(self.p.get("x_0", 0.0))
        self.y0 = float(self.p.get("y_0", 0.0))
        self.k0 = float(self.p.get("k_0", self.p.get("k", 1.0)))
        self.latlong = is_latlong(self.p)
        getattr(self, "_setup_" + self._kind())()

    def _kind(self):
        if self.latlong:
            return "latlong"
        if self.name in ("stere", "lcc", "ob_tran", "merc"):
            return self.name
        raise NotImplementedError("projection %s" % self.name)

    # ---- geographic
    def _setup_latlong(self):
        pass

    # ---- stereographic, sphere
    def _setup_stere(self):
        p = self.p
        phits = _rad(p, "lat_ts", HALFPI) if "lat_ts" in p else HALFPI
        t = abs(self.phi0)
        if abs(t - HALFPI) < _EPS10:
            self.mode = "S" if self.phi0 < 0 else "N"
        else:
            self.mode = "O" if t > _EPS10 else "E"
        phits = abs(phits)
        if self.mode in ("N", "S"):
            if abs(phits - HALFPI) >= _EPS10:
                self.akm1 = math.cos(phits) / math.tan(FORTPI - .5 * phits)
            else:
                self.akm1 = 2. * self.k0
        else:
            self.sinph0, self.cosph0 = math.sin(self.phi0), math.cos(self.phi0)
            self.akm1 = 2. * self.k0

    def _fwd_stere(self, lam, phi):
        sinlam, coslam = np.sin(lam), np.cos(lam)
        if self.mode in ("N", "S"):
            if self.mode == "N":
                coslam, phi = -coslam, -phi
            y = self.akm1 * np.tan(FORTPI + .5 * phi)
            return sinlam * y, coslam * y
        sinphi, cosphi = np.sin(phi), np.cos(phi)
        if self.mode == "E":
            k = self.akm1 / (1. + cosphi * coslam)
            return k * cosphi * sinlam, k * sinphi
        k = self.akm1 / (1. + self.sinph0 * sinphi + self.cosph0 * cosphi * coslam)
        return k * cosphi * sinlam, k * (self.cosph0 * sinphi - self.sinph0 * cosphi * coslam)

    def _inv_stere(self, x, y):
        rh = np.hypot(x, y)
        c = 2. * np.arctan(rh / self.akm1)
        sinc, cosc = np.sin(c), np.cos(c)
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.mode == "N":
                yy = -y
                phi = np.where(np.abs(rh) <= _EPS10, self.phi0, np.arcsin(cosc))
                lam = np.where((x == 0.) & (yy == 0.), 0., np.arctan2(x, yy))
            elif self.mode == "S":
                phi = np.where(np.abs(rh) <= _EPS10, self.phi0, np.arcsin(-cosc))
                lam = np.where((x == 0.) & (y == 0.), 0., np.arctan2(x, y))
            elif self.mode == "E":
                phi = np.where(np.abs(rh) <= _EPS10, 0., np.arcsin(y * sinc / rh))
                lam = np.where((cosc != 0.) | (x != 0.), np.arctan2(x * sinc, cosc * rh), 0.)
            else:
                phi = np.where(np.abs(rh) <= _EPS10, self.phi0,
                               np.arcsin(cosc * self.sinph0 + y * sinc * self.cosph0 / rh))
                cc = cosc - self.sinph0 * np.sin(phi)
                lam = np.where((cc != 0.) | (x != 0.), np.arctan2(x * sinc * self.cosph0, cc * rh), 0.)
        return lam, phi

    # ---- Lambert conformal conic, sphere
    def _setup_lcc(self):
        p = self.p
        phi1 = _rad(p, "lat_1")
        phi2 = _rad(p, "lat_2", phi1) if "lat_2" in p else phi1
        if "lat_0" not in p:
            self.phi0 = phi1
        self.n = sinphi = math.sin(phi1)
        cosphi = math.cos(phi1)
        if abs(phi1 - phi2) >= _EPS10:
            self.n = math.log(cosphi / math.cos(phi2)) / math.log(
                math.tan(FORTPI + .5 * phi2) / math.tan(FORTPI + .5 * phi1))
        self.c = cosphi * math.pow(math.tan(FORTPI + .5 * phi1), self.n) / self.n
        self.rho0 = 0. if abs(abs(self.phi0) - HALFPI) < _EPS10 else \
            self.c * math.pow(math.tan(FORTPI + .5 * self.phi0), -self.n)

    def _fwd_lcc(self, lam, phi):
        with np.errstate(invalid="ignore", divide="ignore"):
            rho = np.where(np.abs(np.abs(phi) - HALFPI) < _EPS10, 0.,
                           self.c * np.power(np.tan(FORTPI + .5 * phi), -self.n))
        lam = lam * self.n
        return self.k0 * (rho * np.sin(lam)), self.k0 * (self.rho0 - rho * np.cos(lam))

    def _inv_lcc(self, x, y):
        x = x / self.k0
        y = self.rho0 - y / self.k0
        rho = np.hypot(x, y)
        if self.n < 0.:
            rho, x, y = -rho, -x, -y
        with np.errstate(invalid="ignore", divide="ignore"):
            phi = np.where(rho != 0., 2. * np.arctan(np.power(self.c / rho, 1. / self.n)) - HALFPI,
                           HALFPI if self.n > 0. else -HALFPI)
            lam = np.where(rho != 0., np.arctan2(x, y) / self.n, 0.)
        return lam, phi

    # ---- Mercator, sphere
    def _setup_merc(self):
        if "lat_ts" in self.p:
            self.k0 = math.cos(abs(_rad(self.p, "lat_ts")))

    def _fwd_merc(self, lam, phi):
        return self.k0 * lam, self.k0 * np.log(np.tan(FORTPI + .5 * phi))

    def _inv_merc(self, x, y):
        return x / self.k0, HALFPI - 2. * np.arctan(np.exp(-y / self.k0))

    # ---- general oblique transformation around a geographic "projection" (rotated pole)
    def _setup_ob_tran(self):
        p = self.p
        if p.get("o_proj") not in ("longlat", "latlong", "latlon", "lonlat"):
            raise NotImplementedError("ob_tran only with +o_proj=longlat")
        if "o_lat_p" not in p:
            raise NotImplementedError("ob_tran only with +o_lat_p / +o_lon_p")
        self.lamp = _rad(p, "o_lon_p")
        phip = _rad(p, "o_lat_p")
        self.oblique = abs(phip - HALFPI) > _EPS10
        self.sphip, self.cphip = math.sin(phip), math.cos(phip)

    def _fwd_ob_tran(self, lam, phi):
        coslam, sinphi, cosphi = np.cos(lam), np.sin(phi), np.cos(phi)
        if self.oblique:
            lamr = adjlon(np.arctan2(cosphi * np.sin(lam), self.sphip * cosphi * coslam + self.cphip * sinphi) + self.lamp)
            phir = np.arcsin(np.clip(self.sphip * sinphi - self.cphip * cosphi * coslam, -1., 1.))
        else:  # transverse aspect is never produced by o_lat_p; plain shift of the pole longitude
            lamr, phir = adjlon(lam + self.lamp), phi
        return lamr, phir  # the linked longlat "projection" leaves radians untouched

    def _inv_ob_tran(self, x, y):
        lamr, phir = x, y
        if self.oblique:
            lamr = lamr - self.lamp
            coslam, sinphi, cosphi = np.cos(lamr), np.sin(phir), np.cos(phir)
            phi = np.arcsin(np.clip(self.sphip * sinphi + self.cphip * cosphi * coslam, -1., 1.))
            lam = np.arctan2(cosphi * np.sin(lamr), self.sphip * cosphi * coslam - self.cphip * sinphi)
        else:
            lam, phi = lamr - self.lamp, phir
        return lam, phi

    # ---- pj_fwd / pj_inv envelopes
    def forward(self, lon, lat):
        """geographic radians -> projected units (radians for latlong and ob_tran+longlat)."""
        lon = np.asarray(lon, dtype=np.float64)
        lat = np.asarray(lat, dtype=np.float64)
        if self.latlong:
            return lon.copy(), lat.copy()
        lam = adjlon(lon - self.lam0)
        x, y = getattr(self, "_fwd_" + self.name)(lam, lat)
        if self.name == "ob_tran":
            return x + self.x0, y + self.y0
        return self.a * x + self.x0, self.a * y + self.y0

    def inverse(self, x, y):
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        if self.latlong:
            return x.copy(), y.copy()
        if self.name == "ob_tran":
            xs, ys = x - self.x0, y - self.y0
        else:
            xs, ys = (x - self.x0) / self.a, (y - self.y0) / self.a
        lam, phi = getattr(self, "_inv_" + self.name)(xs, ys)
        return adjlon(lam + self.lam0), phi


def transform(src, dst, x, y):
    """pj_transform(src, dst, ...): coordinates of src -> coordinates of dst (no datum shift)."""
    ps, pd = _Proj(src), _Proj(dst)
    lon, lat = ps.inverse(x, y)
    return pd.forward(lon, lat)


def project_axes(proj_in, proj_out, x_axis, y_axis):
    """mifi_project_axes (src/interpolation.c:1199-1244): the (y,x) mesh of two axes, transformed."""
    xx, yy = np.meshgrid(np.asarray(x_axis, dtype=np.float64), np.asarray(y_axis, dtype=np.float64))
    return transform(proj_in, proj_out, xx.ravel(), yy.ravel())
