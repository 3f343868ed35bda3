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
Ellipsoids (+ellps, +datum=WGS84/NAD83, +a with +b/+rf/+f/+e/+es) are covered for
merc, lcc, polar and oblique stere, laea, aea, geos, omerc, sinu, cea, tmerc, etmerc and utm (the UTM zone 33 / WGS84 string
of test/testInterpolator.cc:422) with the series PROJ.4 4.x uses (Snyder eq. 7-7,
7-9, 15-7..15-11, 21-33..21-40, 8-9..8-25, 3-21, 3-26); geodetic longitude and
latitude pass unchanged between the two sides unless both name a datum (+datum, +towgs84)
and the two differ: then the three- or seven-parameter shift of pj_datum_transform is applied
through geocentric coordinates at height 0 (grid shifts are refused).

Pins (tests/test_oracle_kats.py): tests/golden/coordTest.nc stores 2-D
longitude/latitude for its 11x11 polar-stereographic grid (121 points);
tests/golden/outData.txt holds the 180x90 nearest-neighbour output of the EMEP
polar-stereographic -> lat/lon chain of test/testInterpolation.cc:280-347.
The ellipsoidal forms are pinned only by the worked numerical examples of
Snyder's appendix A (tests/test_oracle_kats.py), not by a reference fixture.
Beyond those: parity unpinned.
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


# pj_ellps.c: name -> (a, "b" | "rf", value)
ELLIPSOIDS = {
    "sphere": (6370997.0, "b", 6370997.0),
    "WGS84": (6378137.0, "rf", 298.257223563),
    "GRS80": (6378137.0, "rf", 298.257222101),
    "WGS72": (6378135.0, "rf", 298.26),
    "GRS67": (6378160.0, "rf", 298.2471674270),
    "bessel": (6377397.155, "rf", 299.1528128),
    "intl": (6378388.0, "rf", 297.),
    "clrk66": (6378206.4, "b", 6356583.8),
    "clrk80": (6378249.145, "rf", 293.4663),
    "krass": (6378245.0, "rf", 298.3),
    "airy": (6377563.396, "b", 6356256.910),
    "evrstSS": (6377298.556, "rf", 300.8017),
}
DATUMS = {"WGS84": "WGS84", "NAD83": "GRS80", "GGRS87": "GRS80", "potsdam": "bessel"}  # pj_datums.c entries without a grid
DATUM_SHIFTS = {"WGS84": "0,0,0", "NAD83": "0,0,0", "GGRS87": "-199.87,74.79,246.62", "potsdam": "598.1,73.7,418.2,0.202,0.045,-2.455,6.7"}
_SEC_TO_RAD = 4.84813681109535993589914102357e-6
_UNSUPPORTED = ("geoc", "over", "vto_meter", "nadgrids", "geoidgrids",
                "R_A", "R_V", "R_a", "R_g", "R_h", "R_lat_a", "R_lat_g")

# pj_units.c: +units=<id> -> to_meter (text as PROJ.4 stores it: a number or a quotient)
UNITS = {"km": "1000.", "m": "1.", "dm": "1/10", "cm": "1/100", "mm": "1/1000", "kmi": "1852.0", "in": "0.0254", "ft": "0.3048",
         "yd": "0.9144", "mi": "1609.344", "fath": "1.8288", "ch": "20.1168", "link": "0.201168", "us-in": "1./39.37",
         "us-ft": "0.304800609601219", "us-yd": "0.914401828803658", "us-ch": "20.11684023368047", "us-mi": "1609.347218694437",
         "ind-yd": "0.91439523", "ind-ft": "0.30479841", "ind-ch": "20.11669506"}
# pj_datums.c pj_prime_meridians, degrees east of Greenwich
PRIME_MERIDIANS = {"greenwich": 0., "lisbon": -(9 + 7 / 60. + 54.862 / 3600.), "paris": 2 + 20 / 60. + 14.025 / 3600.,
                   "bogota": -(74 + 4 / 60. + 51.3 / 3600.), "madrid": -(3 + 41 / 60. + 16.58 / 3600.), "rome": 12 + 27 / 60. + 8.4 / 3600.,
                   "bern": 7 + 26 / 60. + 22.5 / 3600., "jakarta": 106 + 48 / 60. + 27.79 / 3600., "ferro": -(17 + 40 / 60.),
                   "brussels": 4 + 22 / 60. + 4.71 / 3600., "stockholm": 18 + 3 / 60. + 29.8 / 3600., "athens": 23 + 42 / 60. + 58.815 / 3600.,
                   "oslo": 10 + 43 / 60. + 22.5 / 3600.}


def to_meter_of(p):
    """pj_init: +to_meter=<number>[/<number>] wins over +units=<id>; 1 without either."""
    if "to_meter" in p:
        text = p["to_meter"]
    elif "units" in p:
        if p["units"] not in UNITS:
            raise NotImplementedError("+units=%s" % p["units"])
        text = UNITS[p["units"]]
    else:
        return 1.0
    num, _, den = text.partition("/")
    v = float(num)
    if den:
        v /= float(den)
    if not v > 0:
        raise ValueError("invalid +to_meter: %r" % (p,))
    return v


def from_greenwich_of(p):
    """pj_init: +pm=<name of pj_prime_meridians | decimal degrees east>, in radians."""
    if "pm" not in p:
        return 0.0
    return math.radians(PRIME_MERIDIANS[p["pm"]] if p["pm"] in PRIME_MERIDIANS else float(p["pm"]))


def _ellipsoid(p):
    """pj_ell_set: (a, es).  An explicit +a wins over the one +ellps implies; the shape comes from the first of
    +es, +e, +rf, +f, +b that is present (the ellipsoid's own b / rf counts as given last)."""
    for k in _UNSUPPORTED:
        if k in p:
            raise NotImplementedError("+%s is not restated: %r" % (k, p))
    if p.get("axis", "enu") != "enu":
        raise NotImplementedError("+axis other than enu: %r" % (p,))
    if "R" in p:
        return float(p["R"]), 0.0
    q = dict(p)
    if "datum" in q:
        if q["datum"] not in DATUMS:
            raise NotImplementedError("datum %s" % q["datum"])
        q.setdefault("ellps", DATUMS[q["datum"]])
    if "ellps" in q:
        if q["ellps"] not in ELLIPSOIDS:
            raise NotImplementedError("ellipsoid %s" % q["ellps"])
        a, shape, value = ELLIPSOIDS[q["ellps"]]
        q.setdefault("a", a)
        if not any(k in q for k in ("es", "e", "rf", "f", "b")):
            q[shape] = value
    if "a" not in q:
        if is_latlong(p):
            return 1.0, 0.0  # PROJ.4 would default to WGS84; without a datum shift geographic coordinates ignore it
        raise NotImplementedError("no ellipsoid in %r" % (p,))
    a = float(q["a"])
    if "es" in q:
        es = float(q["es"])
    elif "e" in q:
        es = float(q["e"]) ** 2
    elif "rf" in q:
        es = 1. / float(q["rf"])
        es = es * (2. - es)
    elif "f" in q:
        es = float(q["f"])
        es = es * (2. - es)
    elif "b" in q:
        b = float(q["b"])
        es = 1. - (b * b) / (a * a)
    else:
        es = 0.0
    return a, es


def datum_of(p, a, es):
    """pj_datum_set + the WGS84 test of pj_init: (type, seven parameters); type 0 unknown, 1 three-parameter, 2 seven-parameter
    (position vector: rotations in radians, scale as a factor), 3 WGS84."""
    text = p.get("towgs84")
    if text is None and "datum" in p:
        text = DATUM_SHIFTS[p["datum"]]
    if text is None:
        return 0, [0.] * 7
    v = ([float(t) for t in text.split(",")] + [0.] * 7)[:7]
    if any(v[3:]):
        return 2, v[:3] + [r * _SEC_TO_RAD for r in v[3:6]] + [v[6] / 1e6 + 1.]
    if not any(v[:3]) and a == 6378137.0 and abs(es - 0.006694379990) < 0.000000000050:
        return 3, v
    return 1, v


def geodetic_to_geocentric(lon, lat, h, a, es):
    n = a / np.sqrt(1. - es * np.sin(lat) ** 2)
    return (n + h) * np.cos(lat) * np.cos(lon), (n + h) * np.cos(lat) * np.sin(lon), (n * (1. - es) + h) * np.sin(lat)


def geocentric_to_geodetic(x, y, z, a, es):
    """Fixed point of tan(lat) = (Z + es N sin(lat)) / P (Bowring's relation, iterated to convergence) -- not the loop of
    PROJ.4's geocent.c, which the device code follows; the two meet below 1e-12 rad."""
    p = np.hypot(x, y)
    lat = np.arctan2(z, p * (1. - es))
    for _ in range(12):
        n = a / np.sqrt(1. - es * np.sin(lat) ** 2)
        lat = np.arctan2(z + es * n * np.sin(lat), p)
    n = a / np.sqrt(1. - es * np.sin(lat) ** 2)
    h = np.where(np.abs(np.cos(lat)) > 1e-8, p / np.cos(lat) - n, np.abs(z) - n * (1. - es))
    return np.arctan2(y, x), lat, h


def to_wgs84(kind, v, x, y, z):
    if kind == 1:
        return x + v[0], y + v[1], z + v[2]
    if kind == 2:   # pj_geocentric_to_wgs84, position vector rotation
        return (v[6] * (x - v[5] * y + v[4] * z) + v[0], v[6] * (v[5] * x + y - v[3] * z) + v[1], v[6] * (-v[4] * x + v[3] * y + z) + v[2])
    return x, y, z


def from_wgs84(kind, v, x, y, z):
    if kind == 1:
        return x - v[0], y - v[1], z - v[2]
    if kind == 2:   # pj_geocentric_from_wgs84
        xt, yt, zt = (x - v[0]) / v[6], (y - v[1]) / v[6], (z - v[2]) / v[6]
        return xt + v[5] * yt - v[4] * zt, -v[5] * xt + yt + v[3] * zt, v[4] * xt - v[3] * yt + zt
    return x, y, z


def datum_transform(ps, pd, lon, lat):
    """pj_datum_transform on geodetic coordinates at height 0 (the reference passes z = 0 and ignores what comes back)."""
    ks, vs = datum_of(ps.p, ps.a, ps.es)
    kd, vd = datum_of(pd.p, pd.a, pd.es)
    if ks == 0 or kd == 0:
        return lon, lat
    same = ks == kd and ps.a == pd.a and abs(ps.es - pd.es) <= 5e-11 and (ks == 3 or vs[:3 if ks == 1 else 7] == vd[:3 if ks == 1 else 7])
    if same:
        return lon, lat
    if ps.es == pd.es and ps.a == pd.a and ks not in (1, 2) and kd not in (1, 2):
        return lon, lat
    x, y, z = geodetic_to_geocentric(lon, lat, 0., ps.a, ps.es)
    x, y, z = to_wgs84(ks, vs, x, y, z)
    x, y, z = from_wgs84(kd, vd, x, y, z)
    lo, la, _ = geocentric_to_geodetic(x, y, z, pd.a, pd.es)
    return lo, la


def tsfn(phi, sinphi, e):
    """pj_tsfn, Snyder eq. 7-10 / 15-9."""
    con = e * sinphi
    return np.tan(.5 * (HALFPI - phi)) / np.power((1. - con) / (1. + con), .5 * e)


def msfn(sinphi, cosphi, es):
    """pj_msfn, Snyder eq. 14-15."""
    return cosphi / np.sqrt(1. - es * sinphi * sinphi)


def phi2(ts, e):
    """pj_phi2: latitude from the isometric-latitude function, Snyder eq. 7-9, at most 15 rounds, 1e-10."""
    ts = np.asarray(ts, dtype=np.float64)
    phi = HALFPI - 2. * np.arctan(ts)
    live = np.ones(phi.shape, dtype=bool)
    for _ in range(15):
        con = e * np.sin(phi)
        dphi = HALFPI - 2. * np.arctan(ts * np.power((1. - con) / (1. + con), .5 * e)) - phi
        phi = np.where(live, phi + dphi, phi)
        live = live & (np.abs(dphi) > 1e-10)
        if not live.any():
            break
    return phi


def enfn(es):
    """pj_enfn: coefficients of the meridional distance, Snyder eq. 3-21 regrouped."""
    c00, c02, c04, c06, c08 = 1., .25, .046875, .01953125, .01068115234375
    c22, c44, c46, c48 = .75, .46875, .01302083333333333333, .00712076822916666666
    c66, c68, c88 = .36458333333333333333, .00569661458333333333, .3076171875
    en = [0.] * 5
    en[0] = c00 - es * (c02 + es * (c04 + es * (c06 + es * c08)))
    en[1] = es * (c22 - es * (c04 + es * (c06 + es * c08)))
    t = es * es
    en[2] = t * (c44 - es * (c46 + es * c48))
    t *= es
    en[3] = t * (c66 - es * c68)
    en[4] = t * es * c88
    return en


def mlfn(phi, sphi, cphi, en):
    cphi = cphi * sphi
    sphi = sphi * sphi
    return en[0] * phi - cphi * (en[1] + sphi * (en[2] + sphi * (en[3] + sphi * en[4])))


def inv_mlfn(arg, es, en):
    """pj_inv_mlfn: Newton on the meridional distance, at most 10 rounds, 1e-11."""
    arg = np.asarray(arg, dtype=np.float64)
    k = 1. / (1. - es)
    phi = arg.copy()
    live = np.ones(phi.shape, dtype=bool)
    for _ in range(10):
        s = np.sin(phi)
        t = 1. - es * s * s
        t = (mlfn(phi, s, np.cos(phi), en) - arg) * (t * np.sqrt(t)) * k
        phi = np.where(live, phi - t, phi)
        live = live & (np.abs(t) >= 1e-11)
        if not live.any():
            break
    return phi


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
        self.a, self.es = _ellipsoid(self.p)
        self.e = math.sqrt(self.es)
        self.lam0 = _rad(self.p, "lon_0")
        self.phi0 = _rad(self.p, "lat_0")
        self.x0 = float(self.p.get("x_0", 0.0))
        self.y0 = float(self.p.get("y_0", 0.0))
        self.k0 = float(self.p.get("k_0", self.p.get("k", 1.0)))
        self.to_meter = to_meter_of(self.p)
        self.fr_meter = 1.0 / self.to_meter
        self.from_greenwich = from_greenwich_of(self.p)
        self.latlong = is_latlong(self.p)
        if self.name == "ob_tran" and self.to_meter != 1.0:
            raise NotImplementedError("+units with ob_tran")
        getattr(self, "_setup_" + self._kind())()

    def _kind(self):
        if self.latlong:
            return "latlong"
        if self.name in ("stere", "lcc", "ob_tran", "merc", "tmerc", "etmerc", "utm", "laea", "aea", "geos", "omerc", "sinu", "cea", "ortho", "aeqd", "nsper"):
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
        if self.es != 0.:
            e = self.e
            if self.mode == "E":
                raise NotImplementedError("equatorial stereographic on an ellipsoid (PROJ.4 releases differ there)")
            if self.mode in ("N", "S"):
                if abs(phits - HALFPI) < _EPS10:
                    self.akm1 = 2. * self.k0 / math.sqrt(math.pow(1 + e, 1 + e) * math.pow(1 - e, 1 - e))
                else:
                    t = math.sin(phits)
                    self.akm1 = math.cos(phits) / float(tsfn(phits, t, e))
                    t *= e
                    self.akm1 /= math.sqrt(1. - t * t)
            else:
                t = math.sin(self.phi0)
                chi = 2. * math.atan(self._ssfn(self.phi0, t)) - HALFPI
                t *= e
                self.akm1 = 2. * self.k0 * math.cos(self.phi0) / math.sqrt(1. - t * t)
                self.sinX1, self.cosX1 = math.sin(chi), math.cos(chi)
            return
        if self.mode in ("N", "S"):
            if abs(phits - HALFPI) >= _EPS10:
                self.akm1 = math.cos(phits) / math.tan(FORTPI - .5 * phits)
            else:
                self.akm1 = 2. * self.k0
        else:
            self.sinph0, self.cosph0 = math.sin(self.phi0), math.cos(self.phi0)
            self.akm1 = 2. * self.k0

    def _ssfn(self, phit, sinphi):
        """tan of half the conformal colatitude's complement, Snyder eq. 3-1"""
        con = self.e * sinphi
        return np.tan(.5 * (HALFPI + phit)) * np.power((1. - con) / (1. + con), .5 * self.e)

    def _fwd_stere_ell(self, lam, phi):
        sinlam, coslam, sinphi = np.sin(lam), np.cos(lam), np.sin(phi)
        if self.mode == "O":
            chi = 2. * np.arctan(self._ssfn(phi, sinphi)) - HALFPI
            sinX, cosX = np.sin(chi), np.cos(chi)
            A = self.akm1 / (self.cosX1 * (1. + self.sinX1 * sinX + self.cosX1 * cosX * coslam))
            return A * cosX * sinlam, A * (self.cosX1 * sinX - self.sinX1 * cosX * coslam)
        if self.mode == "S":
            phi, coslam, sinphi = -phi, -coslam, -sinphi
        x = self.akm1 * tsfn(phi, sinphi, self.e)
        return x * sinlam, -x * coslam

    def _inv_stere_ell(self, x, y):
        rho = np.hypot(x, y)
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.mode == "O":
                tp = 2. * np.arctan2(rho * self.cosX1, self.akm1)
                cosphi, sinphi = np.cos(tp), np.sin(tp)
                phi_l = np.where(rho == 0., np.arcsin(cosphi * self.sinX1),
                                 np.arcsin(cosphi * self.sinX1 + (y * sinphi * self.cosX1 / rho)))
                tp = np.tan(.5 * (HALFPI + phi_l))
                xx = x * sinphi
                yy = rho * self.cosX1 * cosphi - y * self.sinX1 * sinphi
                halfpi, halfe = HALFPI, .5 * self.e
            else:
                xx, yy = x, (-y if self.mode == "N" else y)
                tp = -rho / self.akm1
                phi_l = HALFPI - 2. * np.arctan(tp)
                halfpi, halfe = -HALFPI, -.5 * self.e
            phi = phi_l
            live = np.ones(np.shape(phi_l), dtype=bool)
            for _ in range(8):
                sinphi = self.e * np.sin(phi_l)
                nxt = 2. * np.arctan(tp * np.power((1. + sinphi) / (1. - sinphi), halfe)) - halfpi
                phi = np.where(live, nxt, phi)
                live = live & ~(np.abs(phi_l - nxt) < 1e-10)
                phi_l = np.where(live, nxt, phi_l)
                if not live.any():
                    break
            phi = np.where(live, np.nan, phi)  # no convergence: pj_transform reports an error (HUGE_VAL)
            if self.mode == "S":
                phi = -phi
            lam = np.where((xx == 0.) & (yy == 0.), 0., np.arctan2(xx, yy))
        return lam, phi

    def _fwd_stere(self, lam, phi):
        if self.es != 0.:
            return self._fwd_stere_ell(lam, phi)
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
        if self.es != 0.:
            return self._inv_stere_ell(x, y)
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
        if self.es != 0.:
            e = self.e
            m1 = float(msfn(sinphi, cosphi, self.es))
            ml1 = float(tsfn(phi1, sinphi, e))
            if abs(phi1 - phi2) >= _EPS10:
                self.n = math.log(m1 / float(msfn(math.sin(phi2), math.cos(phi2), self.es)))
                self.n /= math.log(ml1 / float(tsfn(phi2, math.sin(phi2), e)))
            self.c = self.rho0 = m1 * math.pow(ml1, -self.n) / self.n
            self.rho0 *= 0. if abs(abs(self.phi0) - HALFPI) < _EPS10 else \
                math.pow(float(tsfn(self.phi0, math.sin(self.phi0), e)), self.n)
            return
        if abs(phi1 - phi2) >= _EPS10:
            self.n = math.log(cosphi / math.cos(phi2)) / math.log(
                math.tan(FORTPI + .5 * phi2) / math.tan(FORTPI + .5 * phi1))
        self.c = cosphi * math.pow(math.tan(FORTPI + .5 * phi1), self.n) / self.n
        self.rho0 = 0. if abs(abs(self.phi0) - HALFPI) < _EPS10 else \
            self.c * math.pow(math.tan(FORTPI + .5 * self.phi0), -self.n)

    def _fwd_lcc(self, lam, phi):
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.es != 0.:
                scale = np.power(tsfn(phi, np.sin(phi), self.e), self.n)
            else:
                scale = np.power(np.tan(FORTPI + .5 * phi), -self.n)
            rho = np.where(np.abs(np.abs(phi) - HALFPI) < _EPS10, 0., self.c * scale)
        lam = lam * self.n
        return self.k0 * (rho * np.sin(lam)), self.k0 * (self.rho0 - rho * np.cos(lam))

    def _inv_lcc(self, x, y):
        x = x / self.k0
        y = self.rho0 - y / self.k0
        rho = np.hypot(x, y)
        if self.n < 0.:
            rho, x, y = -rho, -x, -y
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.es != 0.:
                onto = phi2(np.power(np.where(rho != 0., rho, 1.) / self.c, 1. / self.n), self.e)
            else:
                onto = 2. * np.arctan(np.power(self.c / rho, 1. / self.n)) - HALFPI
            phi = np.where(rho != 0., onto, HALFPI if self.n > 0. else -HALFPI)
            lam = np.where(rho != 0., np.arctan2(x, y) / self.n, 0.)
        return lam, phi

    # ---- Mercator, sphere
    def _setup_merc(self):
        if "lat_ts" in self.p:
            phits = abs(_rad(self.p, "lat_ts"))
            self.k0 = float(msfn(math.sin(phits), math.cos(phits), self.es)) if self.es != 0. else math.cos(phits)

    def _fwd_merc(self, lam, phi):
        if self.es != 0.:
            return self.k0 * lam, -self.k0 * np.log(tsfn(phi, np.sin(phi), self.e))
        return self.k0 * lam, self.k0 * np.log(np.tan(FORTPI + .5 * phi))

    def _inv_merc(self, x, y):
        if self.es != 0.:
            return x / self.k0, phi2(np.exp(-y / self.k0), self.e)
        return x / self.k0, HALFPI - 2. * np.arctan(np.exp(-y / self.k0))

    # ---- transverse Mercator (Gauss-Krueger series of PROJ.4 4.x, Snyder eq. 8-9..8-25) and UTM
    def _setup_utm(self):
        p = self.p
        if self.es == 0.:
            raise ValueError("utm needs an ellipsoid (PROJ.4 error -34)")
        self.y0 = 10000000. if "south" in p else 0.
        self.x0 = 500000.
        if "zone" in p:
            zone = int(p["zone"])
            if not 1 <= zone <= 60:
                raise ValueError("invalid UTM zone")
            zone -= 1
        else:
            zone = int(math.floor((float(adjlon(self.lam0)) + math.pi) * 30. / math.pi))
            zone = min(max(zone, 0), 59)
        self.lam0 = (zone + .5) * math.pi / 30. - math.pi
        self.k0 = 0.9996
        self.phi0 = 0.
        self._setup_etmerc()   # PROJ.4 4.9.3 (the release debian_bionic/control builds against): utm is etmerc; before: tmerc

    def _setup_tmerc(self):
        if self.es != 0.:
            self.en = enfn(self.es)
            self.ml0 = float(mlfn(self.phi0, math.sin(self.phi0), math.cos(self.phi0), self.en))
            self.esp = self.es / (1. - self.es)
        else:
            self.esp = self.k0
            self.ml0 = .5 * self.esp

    def _fwd_tmerc(self, lam, phi):
        sinphi, cosphi = np.sin(phi), np.cos(phi)
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.es == 0.:
                b = cosphi * np.sin(lam)
                x = self.ml0 * np.log((1. + b) / (1. - b))
                y = cosphi * np.cos(lam) / np.sqrt(1. - b * b)
                y = np.where(np.abs(y) >= 1., 0., np.arccos(np.clip(y, -1., 1.)))
                y = np.where(phi < 0., -y, y)
                x = np.where(np.abs(np.abs(b) - 1.) <= _EPS10, np.nan, x)
                return x, self.esp * (y - self.phi0)
            t = np.where(np.abs(cosphi) > 1e-10, sinphi / cosphi, 0.)
            t = t * t
            al = cosphi * lam
            als = al * al
            al = al / np.sqrt(1. - self.es * sinphi * sinphi)
            n = self.esp * cosphi * cosphi
            x = self.k0 * al * (1. + als / 6. * (1. - t + n + als / 20. * (
                5. + t * (t - 18.) + n * (14. - 58. * t) + als / 42. * (61. + t * (t * (179. - t) - 479.)))))
            y = self.k0 * (mlfn(phi, sinphi, cosphi, self.en) - self.ml0 + sinphi * al * lam * .5 * (
                1. + als / 12. * (5. - t + n * (9. + 4. * n) + als / 30. * (
                    61. + t * (t - 58.) + n * (270. - 330. * t) + als / 56. * (1385. + t * (t * (543. - t) - 3111.))))))
            bad = (lam < -HALFPI) | (lam > HALFPI)  # PROJ.4 >= 4.8 refuses the far side
            return np.where(bad, np.nan, x), np.where(bad, np.nan, y)

    def _inv_tmerc(self, x, y):
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.es == 0.:
                h = np.exp(x / self.esp)
                g = .5 * (h - 1. / h)
                h = np.cos(self.phi0 + y / self.esp)
                phi = np.arcsin(np.sqrt((1. - h * h) / (1. + g * g)))
                phi = np.where((y < 0.) & (-phi + self.phi0 < 0.), -phi, phi)  # the hemisphere test of 4.9 (4.8 and older: y < 0 only)
                lam = np.where((g != 0.) | (h != 0.), np.arctan2(g, h), 0.)
                return lam, phi
            phi = inv_mlfn(self.ml0 + y / self.k0, self.es, self.en)
            sinphi, cosphi = np.sin(phi), np.cos(phi)
            t = np.where(np.abs(cosphi) > 1e-10, sinphi / cosphi, 0.)
            n = self.esp * cosphi * cosphi
            con = 1. - self.es * sinphi * sinphi
            d = x * np.sqrt(con) / self.k0
            con = con * t
            t = t * t
            ds = d * d
            phi2_ = phi - (con * ds / (1. - self.es)) * .5 * (1. - ds / 12. * (
                5. + t * (3. - 9. * n) + n * (1. - 4. * n) - ds / 30. * (
                    61. + t * (90. - 252. * n + 45. * t) + 46. * n - ds / 56. * (
                        1385. + t * (3633. + t * (4095. + 1574. * t))))))
            lam = d * (1. - ds / 6. * (1. + 2. * t + n - ds / 20. * (
                5. + t * (28. + 24. * t + 8. * n) + 6. * n - ds / 42. * (
                    61. + t * (662. + t * (1320. + 720. * t)))))) / cosphi
            polar = np.abs(phi) >= HALFPI
            return np.where(polar, 0., lam), np.where(polar, np.where(y < 0., -HALFPI, HALFPI), phi2_)

    # ---- extended transverse Mercator (Krueger series to n^6: Engsager & Poder 2007; Karney 2011 eq. 35, 36), stated
    # here through the closed conformal latitude and plain complex sums instead of PROJ.4's trig series and Clenshaw loops
    _ALPHA = [  # Karney eq. 35 = PROJ.4 gtu
        [1 / 2, -2 / 3, 5 / 16, 41 / 180, -127 / 288, 7891 / 37800],
        [13 / 48, -3 / 5, 557 / 1440, 281 / 630, -1983433 / 1935360],
        [61 / 240, -103 / 140, 15061 / 26880, 167603 / 181440],
        [49561 / 161280, -179 / 168, 6601661 / 7257600],
        [34729 / 80640, -3418889 / 1995840],
        [212378941 / 319334400]]
    _BETA = [   # Karney eq. 36 = minus PROJ.4 utg
        [1 / 2, -2 / 3, 37 / 96, -1 / 360, -81 / 512, 96199 / 604800],
        [1 / 48, 1 / 15, -437 / 1440, 46 / 105, -1118711 / 3870720],
        [17 / 480, -37 / 840, -209 / 4480, 5569 / 90720],
        [4397 / 161280, -11 / 504, -830251 / 7257600],
        [4583 / 161280, -108847 / 3991680],
        [20648693 / 638668800]]

    def _setup_etmerc(self):
        if self.es == 0.:
            raise ValueError("etmerc needs an ellipsoid (PROJ.4 error -34)")
        f = self.es / (1. + math.sqrt(1. - self.es))
        n = f / (2. - f)
        self.Qn = self.k0 / (1. + n) * (1. + n * n * (1 / 4. + n * n * (1 / 64. + n * n / 256.)))
        self.alpha = [sum(c * n ** (j + 1 + i) for i, c in enumerate(row)) for j, row in enumerate(self._ALPHA)]
        self.beta = [sum(c * n ** (j + 1 + i) for i, c in enumerate(row)) for j, row in enumerate(self._BETA)]
        z = float(self._conformal(np.array([self.phi0]))[0])
        self.Zb = -self.Qn * (z + sum(a * math.sin(2 * (j + 1) * z) for j, a in enumerate(self.alpha)))

    def _conformal(self, phi):
        return np.arctan(np.sinh(np.arcsinh(np.tan(phi)) - self.e * np.arctanh(self.e * np.sin(phi))))

    def _fwd_etmerc(self, lam, phi):
        chi = self._conformal(phi)
        xi = np.arctan2(np.sin(chi), np.cos(lam) * np.cos(chi))
        eta = np.arcsinh(np.sin(lam) * np.cos(chi) / np.hypot(np.sin(chi), np.cos(chi) * np.cos(lam)))
        z = xi + 1j * eta
        w = z + sum(a * np.sin(2 * (j + 1) * z) for j, a in enumerate(self.alpha))
        bad = np.abs(w.imag) > 2.623395162778
        return np.where(bad, np.nan, self.Qn * w.imag), np.where(bad, np.nan, self.Qn * w.real + self.Zb)

    def _inv_etmerc(self, x, y):
        w = (y - self.Zb) / self.Qn + 1j * (x / self.Qn)
        bad = np.abs(w.imag) > 2.623395162778
        z = w - sum(b * np.sin(2 * (j + 1) * w) for j, b in enumerate(self.beta))
        xi, eta = z.real, z.imag
        lam = np.arctan2(np.sinh(eta), np.cos(xi))
        chi = np.arcsin(np.clip(np.sin(xi) / np.cosh(eta), -1., 1.))
        phi = chi.copy()
        for _ in range(12):  # geodetic from conformal latitude: fixed point of the closed form (contracts by ~e^2 per round)
            phi = np.arctan(np.sinh(np.arcsinh(np.tan(chi)) + self.e * np.arctanh(self.e * np.sin(phi))))
        return np.where(bad, np.nan, lam), np.where(bad, np.nan, phi)

    def _fwd_utm(self, lam, phi):
        return self._fwd_etmerc(lam, phi)

    def _inv_utm(self, x, y):
        return self._inv_etmerc(x, y)

    # ---- Lambert azimuthal equal-area (Snyder eq. 24-2..24-26, 3-11..3-18; authalic latitude series 3-18 to e^6)
    def _qsfn(self, sinphi):
        if self.e >= 1e-7:
            con = self.e * sinphi
            return (1. - self.es) * (sinphi / (1. - con * con) - (.5 / self.e) * np.log((1. - con) / (1. + con)))
        return sinphi + sinphi

    def _setup_laea(self):
        t = abs(self.phi0)
        if abs(t - HALFPI) < _EPS10:
            self.mode = "S" if self.phi0 < 0 else "N"
        else:
            self.mode = "E" if t < _EPS10 else "O"
        if self.es != 0.:
            es = self.es
            self.qp = float(self._qsfn(1.))
            # pj_authset: Snyder's series 3-18 with PROJ.4's constants (its last one, .0164150..., is not Snyder's 761/45360)
            self.apa = [es * .33333333333333333333 + es * es * .17222222222222222222 + es ** 3 * .10257936507936507936,
                        es * es * .06388888888888888888 + es ** 3 * .06640211640211640211, es ** 3 * .01641501294219154443]
            if self.mode in ("N", "S"):
                self.dd = 1.
            elif self.mode == "E":
                self.rq = math.sqrt(.5 * self.qp)
                self.dd = 1. / self.rq
                self.xmf, self.ymf = 1., .5 * self.qp
            else:
                self.rq = math.sqrt(.5 * self.qp)
                sinphi = math.sin(self.phi0)
                self.sinb1 = float(self._qsfn(sinphi)) / self.qp
                self.cosb1 = math.sqrt(1. - self.sinb1 * self.sinb1)
                self.dd = math.cos(self.phi0) / (math.sqrt(1. - es * sinphi * sinphi) * self.rq * self.cosb1)
                self.xmf = self.rq * self.dd
                self.ymf = self.rq / self.dd
        elif self.mode == "O":
            self.sinb1, self.cosb1 = math.sin(self.phi0), math.cos(self.phi0)

    def _fwd_laea(self, lam, phi):
        coslam, sinlam, sinphi = np.cos(lam), np.sin(lam), np.sin(phi)
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.es != 0.:
                q = self._qsfn(sinphi)
                if self.mode in ("O", "E"):
                    sinb = q / self.qp
                    cosb = np.sqrt(1. - sinb * sinb)
                    if self.mode == "O":
                        b = 1. + self.sinb1 * sinb + self.cosb1 * cosb * coslam
                        bb = np.sqrt(2. / b)
                        y = self.ymf * bb * (self.cosb1 * sinb - self.sinb1 * cosb * coslam)
                    else:
                        b = 1. + cosb * coslam
                        bb = np.sqrt(2. / b)
                        y = bb * sinb * self.ymf
                    x = self.xmf * bb * cosb * sinlam
                    bad = np.abs(b) < _EPS10
                else:
                    if self.mode == "N":
                        b, q = HALFPI + phi, self.qp - q
                    else:
                        b, q = phi - HALFPI, self.qp + q
                    bb = np.sqrt(np.where(q >= 0., q, 0.))
                    x = np.where(q >= 0., bb * sinlam, 0.)
                    y = np.where(q >= 0., coslam * (bb if self.mode == "S" else -bb), 0.)
                    bad = np.abs(b) < _EPS10
                return np.where(bad, np.nan, x), np.where(bad, np.nan, y)
            cosphi = np.cos(phi)
            if self.mode in ("E", "O"):
                y = 1. + cosphi * coslam if self.mode == "E" else 1. + self.sinb1 * sinphi + self.cosb1 * cosphi * coslam
                bad = y <= _EPS10
                k = np.sqrt(2. / y)
                x = k * cosphi * sinlam
                y = k * (sinphi if self.mode == "E" else self.cosb1 * sinphi - self.sinb1 * cosphi * coslam)
            else:
                if self.mode == "N":
                    coslam = -coslam
                bad = np.abs(phi + self.phi0) < _EPS10
                y = FORTPI - phi * .5
                y = 2. * (np.cos(y) if self.mode == "S" else np.sin(y))
                x = y * sinlam
                y = y * coslam
            return np.where(bad, np.nan, x), np.where(bad, np.nan, y)

    def _inv_laea(self, x, y):
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.es != 0.:
                if self.mode in ("E", "O"):
                    x = x / self.dd
                    y = y * self.dd
                    rho = np.hypot(x, y)
                    sCe = 2. * np.arcsin(.5 * rho / self.rq)
                    cCe, sCe = np.cos(sCe), np.sin(sCe)
                    x = x * sCe
                    if self.mode == "O":
                        ab = cCe * self.sinb1 + y * sCe * self.cosb1 / rho
                        y = rho * self.cosb1 * cCe - y * self.sinb1 * sCe
                    else:
                        ab = y * sCe / rho
                        y = rho * cCe
                    centre = rho < _EPS10
                else:
                    if self.mode == "N":
                        y = -y
                    q = x * x + y * y
                    ab = 1. - q / self.qp
                    if self.mode == "S":
                        ab = -ab
                    centre = q == 0.
                beta = np.arcsin(ab)
                t = beta + beta
                phi = beta + self.apa[0] * np.sin(t) + self.apa[1] * np.sin(t + t) + self.apa[2] * np.sin(t + t + t)
                lam = np.arctan2(x, y)
                return np.where(centre, 0., lam), np.where(centre, self.phi0, phi)
            rh = np.hypot(x, y)
            phi = rh * .5
            bad = phi > 1.
            phi = 2. * np.arcsin(np.clip(phi, -1., 1.))
            sinz, cosz = np.sin(phi), np.cos(phi)
            if self.mode == "E":
                phi = np.where(np.abs(rh) <= _EPS10, 0., np.arcsin(y * sinz / rh))
                x = x * sinz
                y = cosz * rh
            elif self.mode == "O":
                phi = np.where(np.abs(rh) <= _EPS10, self.phi0, np.arcsin(cosz * self.sinb1 + y * sinz * self.cosb1 / rh))
                x = x * sinz * self.cosb1
                y = (cosz - np.sin(phi) * self.sinb1) * rh
            elif self.mode == "N":
                y = -y
                phi = HALFPI - phi
            else:
                phi = phi - HALFPI
            lam = np.where((y == 0.) & (self.mode in ("E", "O")), 0., np.arctan2(x, y))
            return np.where(bad, np.nan, lam), np.where(bad, np.nan, phi)

    # ---- Albers equal-area conic (Snyder eq. 14-1..14-21; the inverse on the ellipsoid iterates eq. 3-16)
    def _setup_aea(self):
        p = self.p
        phi1 = _rad(p, "lat_1")
        phi2 = _rad(p, "lat_2")  # PROJ.4 reads +lat_2 with default 0, not lat_1
        if abs(phi1 + phi2) < _EPS10:
            raise ValueError("aea: lat_1 = -lat_2")
        self.n = sinphi = math.sin(phi1)
        cosphi = math.cos(phi1)
        secant = abs(phi1 - phi2) >= _EPS10
        if self.es != 0.:
            m1 = float(msfn(sinphi, cosphi, self.es))
            ml1 = float(self._qsfn(sinphi))
            if secant:
                s2, c2 = math.sin(phi2), math.cos(phi2)
                m2, ml2 = float(msfn(s2, c2, self.es)), float(self._qsfn(s2))
                self.n = (m1 * m1 - m2 * m2) / (ml2 - ml1)
            self.ec = 1. - .5 * (1. - self.es) * math.log((1. - self.e) / (1. + self.e)) / self.e
            self.c = m1 * m1 + self.n * ml1
            self.dd = 1. / self.n
            self.rho0 = self.dd * math.sqrt(self.c - self.n * float(self._qsfn(math.sin(self.phi0))))
        else:
            if secant:
                self.n = .5 * (self.n + math.sin(phi2))
            self.n2 = self.n + self.n
            self.c = cosphi * cosphi + self.n2 * sinphi
            self.dd = 1. / self.n
            self.rho0 = self.dd * math.sqrt(self.c - self.n2 * math.sin(self.phi0))

    def _fwd_aea(self, lam, phi):
        with np.errstate(invalid="ignore"):
            rho = self.c - (self.n * self._qsfn(np.sin(phi)) if self.es != 0. else self.n2 * np.sin(phi))
            bad = rho < 0.
            rho = self.dd * np.sqrt(np.where(bad, 0., rho))
            lam = lam * self.n
            return np.where(bad, np.nan, rho * np.sin(lam)), np.where(bad, np.nan, self.rho0 - rho * np.cos(lam))

    def _inv_aea(self, x, y):
        y = self.rho0 - y
        rho = np.hypot(x, y)
        if self.n < 0.:
            rho, x, y = -rho, -x, -y
        with np.errstate(invalid="ignore", divide="ignore"):
            q = rho / self.dd
            if self.es != 0.:
                q = (self.c - q * q) / self.n
                phi = np.arcsin(np.clip(.5 * q, -1., 1.))   # pj_phi1_: Newton on the authalic q, at most 15 rounds, 1e-10
                live = np.ones(np.shape(phi), dtype=bool)
                for _ in range(15):
                    sinpi, cospi = np.sin(phi), np.cos(phi)
                    con = self.e * sinpi
                    com = 1. - con * con
                    dphi = .5 * com * com / cospi * (q / (1. - self.es) - sinpi / com + .5 / self.e * np.log((1. - con) / (1. + con)))
                    phi = np.where(live, phi + dphi, phi)
                    live = live & (np.abs(dphi) > 1e-10)
                    if not live.any():
                        break
                phi = np.where(live, np.nan, phi)
                phi = np.where(np.abs(self.ec - np.abs(q)) > 1e-7, phi, np.where(q < 0., -HALFPI, HALFPI))
            else:
                q = (self.c - q * q) / self.n2
                phi = np.where(np.abs(q) <= 1., np.arcsin(np.clip(q, -1., 1.)), np.where(q < 0., -HALFPI, HALFPI))
            lam = np.arctan2(x, y) / self.n
            pole = rho == 0.
            return np.where(pole, 0., lam), np.where(pole, HALFPI if self.n > 0. else -HALFPI, phi)

    # ---- geostationary satellite view (PJ_geos.c; CGMS 03 "LRIT/HRIT global specification" 4.4.3.2)
    def _setup_geos(self):
        p = self.p
        self.h = float(p.get("h", 0.))
        if self.h <= 0.:
            raise ValueError("geos needs +h > 0 (PROJ.4 error -30)")
        if self.phi0 != 0.:
            raise ValueError("geos: lat_0 must be 0 (PROJ.4 error -46)")
        sweep = p.get("sweep")
        if sweep not in (None, "x", "y"):
            raise ValueError("geos: +sweep must be x or y")
        self.flip_axis = sweep == "x"
        self.radius_g_1 = self.h / self.a
        self.radius_g = 1. + self.radius_g_1
        self.C = self.radius_g * self.radius_g - 1.
        if self.es != 0.:
            self.radius_p, self.radius_p2, self.radius_p_inv2 = math.sqrt(1. - self.es), 1. - self.es, 1. / (1. - self.es)
        else:
            self.radius_p = self.radius_p2 = self.radius_p_inv2 = 1.

    def _fwd_geos(self, lam, phi):
        with np.errstate(invalid="ignore", divide="ignore"):
            phi = np.arctan(self.radius_p2 * np.tan(phi))
            r = self.radius_p / np.hypot(self.radius_p * np.cos(phi), np.sin(phi))
            vx, vy, vz = r * np.cos(lam) * np.cos(phi), r * np.sin(lam) * np.cos(phi), r * np.sin(phi)
            hidden = ((self.radius_g - vx) * vx - vy * vy - vz * vz * self.radius_p_inv2) < 0.
            tmp = self.radius_g - vx
            if self.flip_axis:
                x, y = self.radius_g_1 * np.arctan(vy / np.hypot(vz, tmp)), self.radius_g_1 * np.arctan(vz / tmp)
            else:
                x, y = self.radius_g_1 * np.arctan(vy / tmp), self.radius_g_1 * np.arctan(vz / np.hypot(vy, tmp))
            return np.where(hidden, np.nan, x), np.where(hidden, np.nan, y)

    def _inv_geos(self, x, y):
        with np.errstate(invalid="ignore", divide="ignore"):
            vx = -1.
            if self.flip_axis:
                vz = np.tan(y / self.radius_g_1)
                vy = np.tan(x / self.radius_g_1) * np.hypot(1., vz)
            else:
                vy = np.tan(x / self.radius_g_1)
                vz = np.tan(y / self.radius_g_1) * np.hypot(1., vy)
            a = vz / self.radius_p
            a = vy * vy + a * a + vx * vx
            b = 2. * self.radius_g * vx
            det = b * b - 4. * a * self.C
            k = (-b - np.sqrt(det)) / (2. * a)
            vx = self.radius_g + k * vx
            vy, vz = vy * k, vz * k
            lam = np.arctan2(vy, vx)
            phi = np.arctan(vz * np.cos(lam) / vx)
            phi = np.arctan(self.radius_p_inv2 * np.tan(phi))
            return np.where(det < 0., np.nan, lam), np.where(det < 0., np.nan, phi)

    # ---- Hotine oblique Mercator, central point and azimuth (PJ_omerc.c; Snyder eq. 9-11..9-48, alternate B)
    def _setup_omerc(self):
        p = self.p
        if "alpha" not in p and "gamma" not in p:
            raise NotImplementedError("omerc by two points (+lat_1 +lon_1 +lat_2 +lon_2)")
        self.no_rot = "no_rot" in p
        no_off = "no_off" in p or "no_uoff" in p
        alp, gam = "alpha" in p, "gamma" in p
        alpha_c, gamma = _rad(p, "alpha"), _rad(p, "gamma")
        lamc = _rad(p, "lonc")
        es, e = self.es, self.e
        com = math.sqrt(1. - es)
        if abs(self.phi0) > _EPS10:
            sinph0, cosph0 = math.sin(self.phi0), math.cos(self.phi0)
            con = 1. - es * sinph0 * sinph0
            B = cosph0 * cosph0
            B = math.sqrt(1. + es * B * B / (1. - es))
            A = B * self.k0 * com / con
            D = B * com / (cosph0 * math.sqrt(con))
            F = D * D - 1.
            F = 0. if F <= 0. else math.copysign(math.sqrt(F), self.phi0)
            F += D
            E = F * math.pow(float(tsfn(self.phi0, sinph0, e)), B)
        else:
            B, A, E, D, F = 1. / com, self.k0, 1., 1., 1.
        if alp:
            gamma0 = math.asin(math.sin(alpha_c) / D)
            if not gam:
                gamma = alpha_c
        else:
            gamma0 = gamma
            alpha_c = math.asin(D * math.sin(gamma0))
        con = abs(alpha_c)
        if con <= 1e-7 or abs(con - math.pi) <= 1e-7 or abs(abs(self.phi0) - HALFPI) <= 1e-7:
            raise ValueError("omerc: alpha is 0 or 180 degrees, or lat_0 a pole (PROJ.4 error -32)")
        self.lam0 = lamc - math.asin(.5 * (F - 1. / F) * math.tan(gamma0)) / B   # +lon_0 plays no role
        self.B, self.A, self.E = B, A, E
        self.singam, self.cosgam = math.sin(gamma0), math.cos(gamma0)
        self.sinrot, self.cosrot = math.sin(gamma), math.cos(gamma)
        self.rB = 1. / B
        self.ArB = A * self.rB
        self.BrA = 1. / self.ArB
        if no_off:
            self.u_0 = 0.
        else:
            self.u_0 = math.copysign(abs(self.ArB * math.atan2(math.sqrt(D * D - 1.), math.cos(alpha_c))), 1. if self.phi0 >= 0. else -1.)
        self.v_pole_n = self.ArB * math.log(math.tan(FORTPI - .5 * gamma0))
        self.v_pole_s = self.ArB * math.log(math.tan(FORTPI + .5 * gamma0))

    def _fwd_omerc(self, lam, phi):
        with np.errstate(invalid="ignore", divide="ignore"):
            Q = self.E / np.power(tsfn(phi, np.sin(phi), self.e), self.B)
            S, T = .5 * (Q - 1. / Q), .5 * (Q + 1. / Q)
            V = np.sin(self.B * lam)
            U = (S * self.singam - V * self.cosgam) / T
            v = .5 * self.ArB * np.log((1. - U) / (1. + U))
            u = self.ArB * np.arctan2(S * self.cosgam + V * self.singam, np.cos(self.B * lam))
            bad = np.abs(np.abs(U) - 1.) < _EPS10
            pole = np.abs(np.abs(phi) - HALFPI) <= _EPS10
            v = np.where(pole, np.where(phi > 0, self.v_pole_n, self.v_pole_s), v)
            u = np.where(pole, self.ArB * phi, u)
            if self.no_rot:
                x, y = u, v
            else:
                u = u - self.u_0
                x, y = v * self.cosrot + u * self.sinrot, u * self.cosrot - v * self.sinrot
            return np.where(bad & ~pole, np.nan, x), np.where(bad & ~pole, np.nan, y)

    def _inv_omerc(self, x, y):
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.no_rot:
                v, u = y, x
            else:
                v = x * self.cosrot - y * self.sinrot
                u = y * self.cosrot + x * self.sinrot + self.u_0
            Qp = np.exp(-self.BrA * v)
            Sp, Tp = .5 * (Qp - 1. / Qp), .5 * (Qp + 1. / Qp)
            Vp = np.sin(self.BrA * u)
            Up = (Vp * self.cosgam + Sp * self.singam) / Tp
            pole = np.abs(np.abs(Up) - 1.) < _EPS10
            t = self.E / np.sqrt((1. + Up) / (1. - Up))
            phi = phi2(np.power(np.where(pole, 1., t), 1. / self.B), self.e)
            lam = -self.rB * np.arctan2(Sp * self.cosgam - Vp * self.singam, np.cos(self.BrA * u))
            return np.where(pole, 0., lam), np.where(pole, np.where(Up < 0., -HALFPI, HALFPI), phi)

    # ---- the remaining projections of the reference's src/coordSys: sinusoidal, cylindrical equal-area, orthographic,
    # azimuthal equidistant, vertical near-side perspective (Snyder ch. 30, 10, 20, 25, 23; the last three on the sphere only,
    # as in PROJ.4 4.x apart from aeqd's geodesic form)
    def _aspect(self):
        t = abs(self.phi0)
        if abs(t - HALFPI) < _EPS10:
            return "S" if self.phi0 < 0 else "N"
        return "E" if t < _EPS10 else "O"

    def _setup_sinu(self):
        if self.es != 0.:
            self.en = enfn(self.es)

    def _fwd_sinu(self, lam, phi):
        s, c = np.sin(phi), np.cos(phi)
        if self.es != 0.:
            return lam * c / np.sqrt(1. - self.es * s * s), mlfn(phi, s, c, self.en)
        return lam * c, phi

    def _inv_sinu(self, x, y):
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.es != 0.:
                phi = inv_mlfn(y, self.es, self.en)
                s = np.sin(phi)
                lam = x * np.sqrt(1. - self.es * s * s) / np.cos(phi)
                a = np.abs(phi)
                return np.where(a < HALFPI, lam, np.where(a - _EPS10 < HALFPI, 0., np.nan)), np.where(a - _EPS10 < HALFPI, phi, np.nan)
            return x / np.cos(y), y

    def _setup_cea(self):
        t = _rad(self.p, "lat_ts")
        self.k0 = math.cos(t)
        if self.es != 0.:
            st = math.sin(t)
            self.k0 /= math.sqrt(1. - self.es * st * st)
            es = self.es
            self.apa = [es * .33333333333333333333 + es * es * .17222222222222222222 + es ** 3 * .10257936507936507936,
                        es * es * .06388888888888888888 + es ** 3 * .06640211640211640211, es ** 3 * .01641501294219154443]
            self.qp = float(self._qsfn(1.))

    def _fwd_cea(self, lam, phi):
        if self.es != 0.:
            return self.k0 * lam, .5 * self._qsfn(np.sin(phi)) / self.k0
        return self.k0 * lam, np.sin(phi) / self.k0

    def _inv_cea(self, x, y):
        with np.errstate(invalid="ignore"):
            if self.es != 0.:
                beta = np.arcsin(np.clip(2. * y * self.k0 / self.qp, -1., 1.))
                t = beta + beta
                return x / self.k0, beta + self.apa[0] * np.sin(t) + self.apa[1] * np.sin(t + t) + self.apa[2] * np.sin(t + t + t)
            y = y * self.k0
            t = np.abs(y)
            phi = np.where(t >= 1., np.where(y < 0., -HALFPI, HALFPI), np.arcsin(np.clip(y, -1., 1.)))
            return np.where(t - _EPS10 <= 1., x / self.k0, np.nan), np.where(t - _EPS10 <= 1., phi, np.nan)

    def _sphere_only(self):
        if self.es != 0.:
            raise NotImplementedError("%s on an ellipsoid" % self.name)
        self.mode = self._aspect()
        self.sinph0, self.cosph0 = math.sin(self.phi0), math.cos(self.phi0)

    def _setup_ortho(self):
        self._sphere_only()

    def _fwd_ortho(self, lam, phi):
        cosphi, coslam, sinphi = np.cos(phi), np.cos(lam), np.sin(phi)
        if self.mode == "E":
            bad, y = cosphi * coslam < -_EPS10, sinphi
        elif self.mode == "O":
            bad = self.sinph0 * sinphi + self.cosph0 * cosphi * coslam < -_EPS10
            y = self.cosph0 * sinphi - self.sinph0 * cosphi * coslam
        else:
            if self.mode == "N":
                coslam = -coslam
            bad, y = np.abs(phi - self.phi0) - _EPS10 > HALFPI, cosphi * coslam
        return np.where(bad, np.nan, cosphi * np.sin(lam)), np.where(bad, np.nan, y)

    def _inv_ortho(self, x, y):
        with np.errstate(invalid="ignore", divide="ignore"):
            rh = np.hypot(x, y)
            bad = rh - 1. > _EPS10
            sinc = np.minimum(rh, 1.)
            cosc = np.sqrt(1. - sinc * sinc)
            if self.mode == "N":
                y, phi = -y, np.arccos(sinc)
            elif self.mode == "S":
                phi = -np.arccos(sinc)
            else:
                if self.mode == "E":
                    phi = y * sinc / rh
                    x, y = x * sinc, cosc * rh
                else:
                    phi = cosc * self.sinph0 + y * sinc * self.cosph0 / rh
                    y = (cosc - self.sinph0 * phi) * rh
                    x = x * sinc * self.cosph0
                phi = np.where(np.abs(phi) >= 1., np.where(phi < 0., -HALFPI, HALFPI), np.arcsin(np.clip(phi, -1., 1.)))
            if self.mode in ("E", "O"):
                lam = np.where(y == 0., np.where(x == 0., 0., np.where(x < 0., -HALFPI, HALFPI)), np.arctan2(x, y))
            else:
                lam = np.arctan2(x, y)
            centre = np.abs(rh) <= _EPS10
            return np.where(bad, np.nan, np.where(centre, 0., lam)), np.where(bad, np.nan, np.where(centre, self.phi0, phi))

    def _setup_aeqd(self):
        self._sphere_only()

    def _fwd_aeqd(self, lam, phi):
        sinphi, cosphi, coslam = np.sin(phi), np.cos(phi), np.cos(lam)
        with np.errstate(invalid="ignore", divide="ignore"):
            if self.mode in ("E", "O"):
                y = cosphi * coslam if self.mode == "E" else self.sinph0 * sinphi + self.cosph0 * cosphi * coslam
                edge = np.abs(np.abs(y) - 1.) < 1e-14
                k = np.arccos(np.clip(y, -1., 1.))
                k = k / np.sin(k)
                x = k * cosphi * np.sin(lam)
                yy = k * (sinphi if self.mode == "E" else self.cosph0 * sinphi - self.sinph0 * cosphi * coslam)
                return np.where(edge, np.where(y < 0., np.nan, 0.), x), np.where(edge, np.where(y < 0., np.nan, 0.), yy)
            if self.mode == "N":
                phi, coslam = -phi, -coslam
            bad = np.abs(phi - HALFPI) < _EPS10
            r = HALFPI + phi
            return np.where(bad, np.nan, r * np.sin(lam)), np.where(bad, np.nan, r * coslam)

    def _inv_aeqd(self, x, y):
        with np.errstate(invalid="ignore", divide="ignore"):
            c_rh = np.hypot(x, y)
            bad = c_rh - _EPS10 > math.pi
            c_rh = np.minimum(c_rh, math.pi)
            centre = c_rh < _EPS10
            if self.mode in ("E", "O"):
                sinc, cosc = np.sin(c_rh), np.cos(c_rh)
                if self.mode == "E":
                    phi = np.arcsin(np.clip(y * sinc / c_rh, -1., 1.))
                    x, y = x * sinc, cosc * c_rh
                else:
                    phi = np.arcsin(np.clip(cosc * self.sinph0 + y * sinc * self.cosph0 / c_rh, -1., 1.))
                    y = (cosc - self.sinph0 * np.sin(phi)) * c_rh
                    x = x * sinc * self.cosph0
                lam = np.where(y == 0., 0., np.arctan2(x, y))
            elif self.mode == "N":
                phi, lam = HALFPI - c_rh, np.arctan2(x, -y)
            else:
                phi, lam = c_rh - HALFPI, np.arctan2(x, y)
            return np.where(bad, np.nan, np.where(centre, 0., lam)), np.where(bad, np.nan, np.where(centre, self.phi0, phi))

    def _setup_nsper(self):
        self._sphere_only()
        height = float(self.p.get("h", 0.))
        if height <= 0.:
            raise ValueError("nsper needs +h > 0 (PROJ.4 error -30)")
        self.pn1 = height / self.a
        self.pp = 1. + self.pn1
        self.rp = 1. / self.pp
        self.hh = 1. / self.pn1
        self.pfact = (self.pp + 1.) * self.hh

    def _fwd_nsper(self, lam, phi):
        sinphi, cosphi, coslam = np.sin(phi), np.cos(phi), np.cos(lam)
        y = {"O": self.sinph0 * sinphi + self.cosph0 * cosphi * coslam, "E": cosphi * coslam, "S": -sinphi, "N": sinphi}[self.mode]
        bad = y < self.rp
        with np.errstate(invalid="ignore", divide="ignore"):
            k = self.pn1 / (self.pp - y)
            x = k * cosphi * np.sin(lam)
            if self.mode == "O":
                yy = k * (self.cosph0 * sinphi - self.sinph0 * cosphi * coslam)
            elif self.mode == "E":
                yy = k * sinphi
            else:
                yy = k * cosphi * (-coslam if self.mode == "N" else coslam)
        return np.where(bad, np.nan, x), np.where(bad, np.nan, yy)

    def _inv_nsper(self, x, y):
        with np.errstate(invalid="ignore", divide="ignore"):
            rh = np.hypot(x, y)
            sinz = 1. - rh * rh * self.pfact
            bad = sinz < 0.
            sinz = (self.pp - np.sqrt(sinz)) / (self.pn1 / rh + rh / self.pn1)
            cosz = np.sqrt(1. - sinz * sinz)
            if self.mode == "O":
                phi = np.arcsin(np.clip(cosz * self.sinph0 + y * sinz * self.cosph0 / rh, -1., 1.))
                y = (cosz - self.sinph0 * np.sin(phi)) * rh
                x = x * sinz * self.cosph0
            elif self.mode == "E":
                phi = np.arcsin(np.clip(y * sinz / rh, -1., 1.))
                y, x = cosz * rh, x * sinz
            elif self.mode == "N":
                phi, y = np.arcsin(cosz), -y
            else:
                phi = -np.arcsin(cosz)
            lam = np.arctan2(x, y)
            centre = np.abs(rh) <= _EPS10
            return np.where(bad, np.nan, np.where(centre, 0., lam)), np.where(bad, np.nan, np.where(centre, self.phi0, phi))

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
        return self.fr_meter * (self.a * x + self.x0), self.fr_meter * (self.a * y + self.y0)  # pj_fwd.c

    def inverse(self, x, y):
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        if self.latlong:
            return x.copy(), y.copy()
        if self.name == "ob_tran":
            xs, ys = x - self.x0, y - self.y0
        else:
            xs, ys = (x * self.to_meter - self.x0) / self.a, (y * self.to_meter - self.y0) / self.a  # pj_inv.c
        lam, phi = getattr(self, "_inv_" + self.name)(xs, ys)
        return adjlon(lam + self.lam0), phi


def transform(src, dst, x, y):
    """pj_transform(src, dst, ...): coordinates of src -> coordinates of dst (no datum shift)."""
    ps, pd = _Proj(src), _Proj(dst)
    lon, lat = ps.inverse(x, y)
    lon = lon + ps.from_greenwich  # pj_transform.c: longitudes meet relative to Greenwich
    lon, lat = datum_transform(ps, pd, lon, lat)
    lon = lon - pd.from_greenwich
    return pd.forward(lon, lat)


def project_axes(proj_in, proj_out, x_axis, y_axis):
    """mifi_project_axes (src/interpolation.c:1199-1244): the (y,x) mesh of two axes, transformed."""
    xx, yy = np.meshgrid(np.asarray(x_axis, dtype=np.float64), np.asarray(y_axis, dtype=np.float64))
    return transform(proj_in, proj_out, xx.ravel(), yy.ravel())
