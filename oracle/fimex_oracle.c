/*
 * fimex_oracle.c -- CPU restatement of the Fimex regridding hot path (see fimex_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/, smoke() and bench.py's
 * cpu_baseline leg.  Never linked into the product.
 *
 * Build: gcc -std=gnu99 -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 *
 * Parity pins: tests/test_oracle_kats.py replays every known-answer test the
 * reference holds for these functions (test/testInterpolation.cc:49-155,
 * 280-393 with test/inData.txt + test/outData.txt, 396-654).
 *
 * Deliberate, documented divergences from the reference (all are undefined
 * behaviour or an out-of-bounds read there):
 *  D1 bilinear nearest/nearest corner: the reference accepts lround(y) == iy
 *     (src/interpolation.c:936, "y0 <= iy") and then reads one row past the
 *     slice; here that cell is undefined (NaN).
 *  D2 coordinates that are non-finite or beyond +-2^30 are "outside" (NaN
 *     result); the reference casts them to int (UB).
 *  D3 offsets are size_t, the reference's int mifi_3d_array_position
 *     (include/fimex/interpolation.h:423-426) overflows beyond 2^31 cells.
 *  D4 forward_undef_median with a NaN in the bucket returns NaN; the reference
 *     runs std::nth_element with a comparator that is not a strict weak order
 *     (result implementation-defined).
 *  D5 fill2d / creepfill on slices with nx < 2 or ny < 2 return ORC_ERROR when
 *     there is something to fill (the reference reads out of bounds).
 *  D6 float -> integer conversion of values beyond the range of long: lround is
 *     unspecified there; LONG_MIN (glibc, x86-64) is kept, then truncated to int
 *     as MetNoFimex::round does (include/fimex/Utils.h:72-75).
 */
#include "fimex_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.1415926535897932384626433832795 /* mifi_constants.h:42 */
#define ORC_RAD_TO_DEG 57.29577951308232         /* proj_api.h RAD_TO_DEG */
#define ORC_COORD_LIMIT 1073741824.0             /* D2 */

static float orc_nanf(void) { return nanf(""); }

static void fill_undefined(float* out, size_t iz)
{
    for (size_t z = 0; z < iz; ++z) out[z] = orc_nanf();
}

static int coord_usable(double x, double y)
{
    return isfinite(x) && isfinite(y) && fabs(x) < ORC_COORD_LIMIT && fabs(y) < ORC_COORD_LIMIT;
}

/* ------------------------------------------------------------------ nearest */
/* src/interpolation.c:862-879 */
int orc_get_values_f(const float* in, float* out, double x, double y, size_t ix, size_t iy, size_t iz)
{
    if (!coord_usable(x, y)) { fill_undefined(out, iz); return ORC_OK; }
    long rx = lround(x); /* half away from zero, :864 */
    long ry = lround(y); /* :865 */
    if (rx >= 0 && rx < (long)ix && ry >= 0 && ry < (long)iy) { /* :867-868 */
        const size_t layer = ix * iy;
        size_t pos = (size_t)ry * ix + (size_t)rx;
        for (size_t z = 0; z < iz; ++z, pos += layer) out[z] = in[pos]; /* :869-871 */
    } else {
        fill_undefined(out, iz); /* :874-876 */
    }
    return ORC_OK;
}

/* ----------------------------------------------------------------- bilinear */
/* src/interpolation.c:881-957 */
int orc_get_values_bilinear_f(const float* in, float* out, double x, double y, size_t ix, size_t iy, size_t iz)
{
    if (!coord_usable(x, y)) { fill_undefined(out, iz); return ORC_OK; }
    const size_t layer = ix * iy;
    long x0 = (long)floor(x);          /* :883 */
    long y0 = (long)floor(y);          /* :886 */
    float xfrac = (float)(x - (double)x0); /* :885 double difference rounded to float */
    float yfrac = (float)(y - (double)y0); /* :888 */
    int xlin = (0 <= x0) && (x0 + 1 < (long)ix); /* :889 */
    int ylin = (0 <= y0) && (y0 + 1 < (long)iy); /* :890 / :925 */

    if (xlin && ylin) {
        /* :892-902 */
        size_t pos = (size_t)y0 * ix + (size_t)x0;
        for (size_t z = 0; z < iz; ++z, pos += layer) {
            float s00 = in[pos], s01 = in[pos + 1], s10 = in[pos + ix], s11 = in[pos + ix + 1];
            out[z] = (1.f - yfrac) * ((1.f - xfrac) * s00 + xfrac * s01)
                   + yfrac * ((1.f - xfrac) * s10 + xfrac * s11); /* :899-900 */
        }
    } else if (xlin) {
        long ry = lround(y); /* :904 */
        if (0 <= ry && ry < (long)iy) {
            /* linear in x, nearest in y :907-913 */
            size_t pos = (size_t)ry * ix + (size_t)x0;
            for (size_t z = 0; z < iz; ++z, pos += layer) {
                float s00 = in[pos], s01 = in[pos + 1];
                out[z] = (1.f - xfrac) * s00 + xfrac * s01; /* :911 */
            }
        } else {
            fill_undefined(out, iz); /* :916-918 */
        }
    } else {
        long rx = lround(x); /* :922 */
        if (0 <= rx && rx < (long)ix) {
            if (ylin) {
                /* nearest in x, linear in y :927-933 */
                size_t pos = (size_t)y0 * ix + (size_t)rx;
                for (size_t z = 0; z < iz; ++z, pos += layer) {
                    float s00 = in[pos], s10 = in[pos + ix];
                    out[z] = (1 - yfrac) * s00 + (yfrac * s10); /* :931 */
                }
            } else {
                long ry = lround(y); /* :935 */
                if (0 <= ry && ry < (long)iy) { /* :936 has "<= iy": divergence D1 */
                    size_t pos = (size_t)ry * ix + (size_t)rx;
                    for (size_t z = 0; z < iz; ++z, pos += layer) out[z] = in[pos]; /* :939-942 */
                } else {
                    fill_undefined(out, iz);
                }
            }
        } else {
            fill_undefined(out, iz); /* :950-952 */
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ bicubic */
/* src/interpolation.c:959-1028 (Keys kernel, a = -0.5) */
int orc_get_values_bicubic_f(const float* in, float* out, double x, double y, size_t ix, size_t iy, size_t iz)
{
    if (!coord_usable(x, y)) { fill_undefined(out, iz); return ORC_OK; }
    /* :962-968: convolution matrix times one half (exact) */
    static const double M[4][4] = {{ 0.0,  1.0,  0.0,  0.0},
                                   {-0.5,  0.0,  0.5,  0.0},
                                   { 1.0, -2.5,  2.0, -0.5},
                                   {-0.5,  1.5, -1.5,  0.5}};
    long x0 = (long)floor(x);           /* :970 */
    long y0 = (long)floor(y);           /* :972 */
    double xfrac = x - (double)x0;      /* :971 */
    double yfrac = y - (double)y0;      /* :973 */
    if (!((1 <= x0) && (x0 + 2 < (long)ix) && (1 <= y0) && (y0 + 2 < (long)iy))) { /* :975-976 */
        fill_undefined(out, iz); /* :1022-1026 */
        return ORC_OK;
    }
    double X[4], Y[4], XM[4], MY[4];
    X[0] = 1; X[1] = xfrac; X[2] = xfrac * xfrac; X[3] = X[2] * xfrac; /* :981-984 */
    Y[0] = 1; Y[1] = yfrac; Y[2] = yfrac * yfrac; Y[3] = Y[2] * yfrac; /* :991-994 */
    for (int i = 0; i < 4; ++i) {
        XM[i] = 0;
        for (int j = 0; j < 4; ++j) XM[i] += X[j] * M[j][i]; /* :985-990 */
        MY[i] = 0;
        for (int j = 0; j < 4; ++j) MY[i] += Y[j] * M[j][i]; /* :995-1000 */
    }
    const size_t layer = ix * iy;
    const size_t base = (size_t)(y0 - 1) * ix + (size_t)(x0 - 1);
    for (size_t z = 0; z < iz; ++z) {
        const float* s = in + z * layer + base;
        float acc = 0; /* :1005 accumulates into the float output */
        for (int i = 0; i < 4; ++i) {      /* i: y offset of the stencil row */
            double xmf = 0;                /* :1013 */
            for (int j = 0; j < 4; ++j)    /* j: x offset, F[j][i] = in(x0+j-1, y0+i-1) :1008,1015 */
                xmf += XM[j] * (double)s[(size_t)i * ix + (size_t)j];
            acc = (float)((double)acc + xmf * MY[i]); /* :1019 float += double */
        }
        out[z] = acc;
    }
    return ORC_OK;
}

/* -------------------------------------------------------- backward apply loop */
typedef int (*orc_point_fn)(const float*, float*, double, double, size_t, size_t, size_t);

/* src/CachedInterpolation.cc:93-147 */
int orc_interpolate_values(int funcType, const double* px, const double* py,
                           const float* in, size_t inX, size_t inY, size_t inZ,
                           size_t outX, size_t outY, float* out, int nthreads)
{
    orc_point_fn fn;
    switch (funcType) { /* :107-115 */
        case ORC_BILINEAR: fn = orc_get_values_bilinear_f; break;
        case ORC_BICUBIC: fn = orc_get_values_bicubic_f; break;
        case ORC_NEAREST: case ORC_COORD_NN: case ORC_COORD_NN_KD: fn = orc_get_values_f; break;
        default: return ORC_ERROR;
    }
    const size_t outLayer = outX * outY;
    if (nthreads < 1) nthreads = 1;
    int failed = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads) default(shared)
#endif
    {
        float* zValues = (float*)malloc((inZ ? inZ : 1) * sizeof(float)); /* :129 per-thread scratch */
#ifdef _OPENMP
#pragma omp for
#endif
        for (long long xy = 0; xy < (long long)outLayer; ++xy) { /* :133 */
            if (zValues == NULL) { failed = 1; continue; }
            fn(in, zValues, px[xy], py[xy], inX, inY, inZ);
            float* o = out + xy;
            for (size_t z = 0; z < inZ; ++z, o += outLayer) *o = zValues[z]; /* :136-139 */
        }
        free(zValues);
    }
    return failed ? ORC_ERROR : ORC_OK;
}

/* src/CachedInterpolation.cc:149-157 */
static long long clamp_ll(long long low, double dvalue, long long high)
{
    long long value = (long long)dvalue;
    if (value < low) return low;
    if (value < high) return value;
    return high;
}

/* src/CachedInterpolation.cc:159-200 */
int orc_create_reduced_domain(double* px, double* py, size_t n, size_t inX, size_t inY,
                              size_t* xMin, size_t* yMin, size_t* newInX, size_t* newInY)
{
    if (n == 0) return 0;
    double minx = px[0], maxx = px[0], miny = py[0], maxy = py[0];
    for (size_t i = 1; i < n; ++i) { /* std::min_element / max_element with operator< :165-168 */
        if (px[i] < minx) minx = px[i];
        if (maxx < px[i]) maxx = px[i];
        if (py[i] < miny) miny = py[i];
        if (maxy < py[i]) maxy = py[i];
    }
    const long long EXTEND = 2; /* :171 */
    long long x0 = clamp_ll(0, floor(minx) - EXTEND, (long long)inX - 1);
    long long y0 = clamp_ll(0, floor(miny) - EXTEND, (long long)inY - 1);
    long long x1 = clamp_ll(0, ceil(maxx) + EXTEND, (long long)inX - 1);
    long long y1 = clamp_ll(0, ceil(maxy) + EXTEND, (long long)inY - 1);
    if ((x1 - x0) < 1 || (y1 - y0) < 1) return 0; /* :178-179 */
    for (size_t i = 0; i < n; ++i) { px[i] -= x0; py[i] -= y0; } /* :182-185 */
    *xMin = (size_t)x0; *yMin = (size_t)y0;
    *newInX = (size_t)(x1 - x0 + 1); *newInY = (size_t)(y1 - y0 + 1); /* :198-199 */
    return 1;
}

/* ------------------------------------------------------------------ forward */
/* src/Utils.cc:42-58; round() is half away from zero */
int orc_round_and_clamp(double d, int mini, int maxi, int invalid)
{
    if (!isfinite(d) || fabs(d) >= ORC_COORD_LIMIT) return invalid; /* D2 */
    int r = (int)round(d);
    return (r >= mini && r <= maxi) ? r : invalid;
}

static int cmp_float_asc(const void* a, const void* b)
{
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* the five aggregators of src/CachedForwardInterpolation.cc:38-59 on one bucket (push order) */
static float aggregate_bucket(int kind, float* v, size_t n)
{
    switch (kind) {
        case 0: { float s = 0.f; for (size_t i = 0; i < n; ++i) s = s + v[i]; return s; }              /* :38-45 */
        case 1: { float s = 0.f; for (size_t i = 0; i < n; ++i) s = s + v[i]; return s / (float)n; }   /* :46-48 */
        case 2: {                                                                                       /* :49-53 */
            for (size_t i = 0; i < n; ++i) if (isnan(v[i])) return orc_nanf(); /* D4 */
            qsort(v, n, sizeof(float), cmp_float_asc);
            return v[n / 2];
        }
        case 3: { size_t m = 0; for (size_t i = 1; i < n; ++i) if (v[m] < v[i]) m = i; return v[m]; }  /* :54-56 */
        default: { size_t m = 0; for (size_t i = 1; i < n; ++i) if (v[i] < v[m]) m = i; return v[m]; } /* :57-59 */
    }
}

/* src/CachedForwardInterpolation.cc:62-131 */
int orc_forward_interpolate_values(int funcType, const double* px, const double* py,
                                   const float* in, size_t inX, size_t inY, size_t inZ,
                                   size_t outX, size_t outY, float* out)
{
    int kind, undefAggr;
    switch (funcType) { /* :76-89 */
        case ORC_FWD_SUM: kind = 0; undefAggr = 0; break;
        case ORC_FWD_MEAN: kind = 1; undefAggr = 0; break;
        case ORC_FWD_MEDIAN: kind = 2; undefAggr = 0; break;
        case ORC_FWD_MAX: kind = 3; undefAggr = 0; break;
        case ORC_FWD_MIN: kind = 4; undefAggr = 0; break;
        case ORC_FWD_UNDEF_SUM: kind = 0; undefAggr = 1; break;
        case ORC_FWD_UNDEF_MEAN: kind = 1; undefAggr = 1; break;
        case ORC_FWD_UNDEF_MEDIAN: kind = 2; undefAggr = 1; break;
        case ORC_FWD_UNDEF_MAX: kind = 3; undefAggr = 1; break;
        case ORC_FWD_UNDEF_MIN: kind = 4; undefAggr = 1; break;
        default: return ORC_ERROR;
    }
    const size_t nIn = inX * inY, nOut = outX * outY;
    int* tx = (int*)malloc((nIn ? nIn : 1) * sizeof(int));
    int* ty = (int*)malloc((nIn ? nIn : 1) * sizeof(int));
    size_t* start = (size_t*)calloc(nOut + 1, sizeof(size_t));
    size_t* fillp = (size_t*)malloc((nOut ? nOut : 1) * sizeof(size_t));
    float* vals = (float*)malloc((nIn ? nIn : 1) * sizeof(float));
    if (!tx || !ty || !start || !fillp || !vals) { free(tx); free(ty); free(start); free(fillp); free(vals); return ORC_ERROR; }
    for (size_t i = 0; i < nIn; ++i) { /* ctor :72-73 */
        tx[i] = orc_round_and_clamp(px[i], 0, (int)outX - 1, -1);
        ty[i] = orc_round_and_clamp(py[i], 0, (int)outY - 1, -1);
    }
    for (size_t z = 0; z < inZ; ++z) {
        const float* src = in + z * nIn;
        /* the reference push_back()s in source scan order (:101-110); a count/fill pass
         * over the same scan order yields the same per-bucket sequences */
        memset(start, 0, (nOut + 1) * sizeof(size_t));
        for (size_t i = 0; i < nIn; ++i) {
            float val = src[i];
            if ((undefAggr || !isnan(val)) && tx[i] >= 0 && ty[i] >= 0)
                start[(size_t)ty[i] * outX + (size_t)tx[i] + 1]++;
        }
        for (size_t t = 0; t < nOut; ++t) { start[t + 1] += start[t]; fillp[t] = start[t]; }
        for (size_t i = 0; i < nIn; ++i) {
            float val = src[i];
            if ((undefAggr || !isnan(val)) && tx[i] >= 0 && ty[i] >= 0)
                vals[fillp[(size_t)ty[i] * outX + (size_t)tx[i]]++] = val;
        }
        float* o = out + z * nOut;
        for (size_t t = 0; t < nOut; ++t) { /* :119-129 */
            size_t n = start[t + 1] - start[t];
            o[t] = (n == 0) ? orc_nanf() : aggregate_bucket(kind, vals + start[t], n);
        }
    }
    free(tx); free(ty); free(start); free(fillp); free(vals);
    return ORC_OK;
}

/* ---------------------------------------------------------- vector rotation */
/* src/interpolation.c:790-812; matrix = (cos, sin, -sin, phi) per cell, :429-432 */
int orc_vector_reproject_values_by_matrix_f(const double* matrix, float* u, float* v, size_t ox, size_t oy, size_t oz)
{
    const size_t layer = ox * oy;
    for (size_t z = 0; z < oz; ++z) {
        float* uz = u + z * layer;
        float* vz = v + z * layer;
        for (size_t i = 0; i < layer; ++i) {
            const double c = matrix[4 * i], s = matrix[4 * i + 1];
            double un = (double)uz[i] * c - (double)vz[i] * s; /* :804 */
            double vn = (double)uz[i] * s + (double)vz[i] * c; /* :805 */
            uz[i] = (float)un;
            vz[i] = (float)vn;
        }
    }
    return ORC_OK;
}

/* src/interpolation.c:814-835 */
int orc_vector_reproject_direction_by_matrix_f(const double* matrix, float* angles, size_t ox, size_t oy, size_t oz)
{
    const size_t layer = ox * oy;
    for (size_t z = 0; z < oz; ++z) {
        float* a = angles + z * layer;
        for (size_t i = 0; i < layer; ++i) {
            double an = (double)a[i] - ORC_RAD_TO_DEG * matrix[4 * i + 3]; /* :827 */
            if (an < 0) an += 360;   /* :829 */
            if (an > 360) an -= 360; /* :830 */
            a[i] = (float)an;
        }
    }
    return ORC_OK;
}

/* src/interpolation.c:311-329 */
static double bearing(double lat0, double lon0, double lat1, double lon1)
{
    double dlon = lon0 - lon1;
    return atan2(sin(dlon) * cos(lat1), cos(lat0) * sin(lat1) - sin(lat0) * cos(lat1) * cos(dlon));
}

/* src/interpolation.c:330-438 with the two pj_transform calls replaced by inputs */
int orc_vector_matrix_from_deltas(const double* out_x, const double* out_y,
                                  const double* xdx_x, const double* xdx_y,
                                  const double* ydy_x, const double* ydy_y,
                                  double deltaX, double deltaY, int outIsLatLon,
                                  size_t n, double* matrix)
{
    const double signX = deltaX > 0 ? 1. : -1.; /* :363 */
    const double signY = deltaY > 0 ? 1. : -1.; /* :405 */
    for (size_t i = 0; i < n; ++i) {
        double phiy; /* angle of the x-displaced point, kept in matrix[0] by the reference :379 */
        if (outIsLatLon) {
            phiy = bearing(out_y[i], out_x[i], xdx_y[i], xdx_x[i]); /* :367 */
        } else {
            phiy = atan2(xdx_y[i] - out_y[i], xdx_x[i] - out_x[i]); /* :372-373 */
            if (signX < 0) phiy += ORC_PI;                          /* :374-376 */
        }
        double phi0;
        if (outIsLatLon) {
            phi0 = bearing(out_y[i], out_x[i], ydy_y[i], ydy_x[i]); /* :409 */
            if (signY < 0) phi0 += ORC_PI;
        } else {
            double phix = -1 * atan2(ydy_x[i] - out_x[i], ydy_y[i] - out_y[i]); /* :414-415 */
            if (signY < 0) phix += ORC_PI;
            phi0 = .5 * (phix + phiy); /* :424 */
        }
        double c = cos(phi0), s = sin(phi0);
        matrix[4 * i + 0] = c;
        matrix[4 * i + 1] = s;
        matrix[4 * i + 2] = -1 * s;
        matrix[4 * i + 3] = phi0; /* :429-432 */
    }
    return ORC_OK;
}

/* -------------------------------------------------------------------- fills */
/* src/interpolation.c:1246-1376 */
int orc_fill2d_f(size_t nx, size_t ny, float* field, float relaxCrit, float corrEff, size_t maxLoop, size_t* nChanged)
{
    const size_t total = nx * ny;
    if (total == 0) return ORC_OK;
    double sum = 0;
    size_t nUndef = 0;
    for (size_t i = 0; i < total; ++i) { /* :1256-1264 */
        if (isnan(field[i])) nUndef++;
        else sum += field[i];
    }
    *nChanged = nUndef;
    const size_t nDef = total - nUndef;
    if (nDef == 0 || nUndef == 0) return ORC_OK; /* :1266-1268 */
    if (nx < 2 || ny < 2) return ORC_ERROR;      /* D5 */

    float* w = (float*)malloc(total * sizeof(float));
    float* e = (float*)malloc(total * sizeof(float));
    if (!w || !e) { free(w); free(e); return ORC_ERROR; }
    memset(e, 0, total * sizeof(float));

    const double average = sum / nDef; /* :1281 */
    double dev = 0;
    for (size_t i = 0; i < total; ++i) { /* :1288-1299 */
        if (isnan(field[i])) {
            w[i] = 1.f;
            field[i] = (float)average;
        } else {
            dev += fabs(field[i] - average);
            w[i] = 0.f;
        }
    }
    dev /= nDef;                          /* :1300 */
    const double crit = relaxCrit * dev;  /* :1302 */
    const size_t nxm1 = nx - 1, nym1 = ny - 1;
    for (size_t y = 1; y < nym1; ++y)     /* :1311-1315 */
        for (size_t x = 1; x < nxm1; ++x) w[y * nx + x] *= corrEff;

    for (size_t n = 0; n < maxLoop; ++n) { /* :1324 */
        for (size_t y = 1; y < nym1; ++y) {
            for (size_t x = 1; x < nxm1; ++x) { /* in-place Gauss-Seidel sweep :1329-1336 */
                const size_t p = y * nx + x;
                e[p] = (float)((field[p + 1] + field[p - 1] + field[p + nx] + field[p - nx]) * 0.25 - field[p]); /* :1332 */
                field[p] += e[p] * w[p];                                                                        /* :1333 */
            }
        }
        if ((n < (maxLoop - 5)) && (n % 10 == 0)) { /* :1339-1360, size_t arithmetic as in the reference */
            const float crtest = (float)(crit * corrEff);
            int nbad = 0;
            for (size_t y = 1; y < nym1; ++y) {
                if (nbad) break; /* checked once per row, :1346 */
                for (size_t x = 1; x < nxm1; ++x) {
                    const size_t p = y * nx + x;
                    if (fabs(e[p] * w[p]) > crtest) nbad = 1; /* :1349 */
                }
            }
            if (!nbad) { free(e); free(w); return ORC_OK; }
        }
        for (size_t y = 1; y < nym1; ++y) { /* :1363-1366 */
            field[y * nx] += (field[y * nx + 1] - field[y * nx]) * w[y * nx];
            field[y * nx + nxm1] += (field[y * nx + nx - 2] - field[y * nx + nxm1]) * w[y * nx + nxm1];
        }
        for (size_t x = 0; x < nx; ++x) { /* :1367-1370 */
            field[x] += (field[nx + x] - field[x]) * w[x];
            field[nym1 * nx + x] += (field[(nym1 - 1) * nx + x] - field[nym1 * nx + x]) * w[nym1 * nx + x];
        }
    }
    free(e); free(w);
    return ORC_OK;
}

/* src/interpolation.c:1378-1493 */
static int creepfill_impl(size_t nx, size_t ny, float* field, float defaultVal, unsigned short repeat, char setWeight, size_t nUndef)
{
    const size_t total = nx * ny;
    if (total == 0) return ORC_OK;
    const size_t nDef = total - nUndef;
    if (nDef == 0 || nUndef == 0) return ORC_OK; /* :1384-1386 */
    if (nx < 2 || ny < 2) return ORC_ERROR;      /* D5 */
    char* w = (char*)malloc(total);
    unsigned short* r = (unsigned short*)malloc(total * sizeof(unsigned short));
    if (!w || !r) { free(w); free(r); return ORC_ERROR; }
    for (size_t i = 0; i < total; ++i) { /* :1408-1421 */
        if (isnan(field[i])) { w[i] = 0; r[i] = 0; field[i] = defaultVal; }
        else { w[i] = setWeight; r[i] = repeat; }
    }
    const size_t nxm1 = nx - 1, nym1 = ny - 1;
    size_t l = 0, changedInLoop = 1;
    while (changedInLoop > 0 && l < nDef) { /* :1430 */
        changedInLoop = 0;
        l++;
        for (size_t y = 1; y < nym1; ++y) {
            for (size_t x = 1; x < nxm1; ++x) { /* :1440-1461 */
                const size_t p = y * nx + x;
                if (r[p] < repeat) {
                    size_t wsum = (size_t)(w[p + 1] + w[p - 1] + w[p + nx] + w[p - nx]); /* :1445 */
                    if (wsum != 0) {
                        field[p] += w[p + 1] * field[p + 1] + w[p - 1] * field[p - 1]
                                  + w[p + nx] * field[p + nx] + w[p - nx] * field[p - nx]; /* :1451 */
                        field[p] /= (1 + wsum);                                            /* :1452 */
                        w[p] = 1;
                        r[p]++;
                        changedInLoop++;
                    }
                }
            }
        }
    }
    for (size_t k = 0; k < repeat; ++k) { /* :1464-1489 */
        for (size_t y = 1; y < nym1; ++y) {
            if (r[y * nx] < repeat) {
                field[y * nx] += field[y * nx + 1] * w[y * nx + 1];
                field[y * nx] /= (1 + w[y * nx + 1]);
                w[y * nx] = 1;
            }
            if (r[y * nx + nxm1] < repeat) {
                field[y * nx + nxm1] += field[y * nx + nx - 2] * w[y * nx + nx - 2];
                field[y * nx + nxm1] /= (1 + w[y * nx + nx - 2]);
                w[y * nx + nxm1] = 1;
            }
        }
        for (size_t x = 0; x < nx; ++x) {
            if (r[x] < repeat) {
                field[x] += field[nx + x] * w[nx + x];
                field[x] /= (1 + w[nx + x]);
                w[x] = 1;
            }
            if (r[nym1 * nx + x] < repeat) {
                field[nym1 * nx + x] += field[(nym1 - 1) * nx + x] * w[(nym1 - 1) * nx + x];
                field[nym1 * nx + x] /= (1 + w[(nym1 - 1) * nx + x]);
                w[nym1 * nx + x] = 1;
            }
        }
    }
    free(r); free(w);
    return ORC_OK;
}

/* src/interpolation.c:1495-1519 */
int orc_creepfill2d_f(size_t nx, size_t ny, float* field, unsigned short repeat, char setWeight, size_t* nChanged)
{
    const size_t total = nx * ny;
    if (total == 0) return ORC_OK;
    double sum = 0;
    size_t nUndef = 0;
    for (size_t i = 0; i < total; ++i) {
        if (isnan(field[i])) nUndef++;
        else sum += field[i];
    }
    *nChanged = nUndef;
    const size_t nDef = total - nUndef;
    if (nDef == 0) return ORC_OK;
    float average = (float)(sum / nDef); /* :1516 */
    return creepfill_impl(nx, ny, field, average, repeat, setWeight, nUndef);
}

/* src/interpolation.c:1521-1537 */
int orc_creepfillval2d_f(size_t nx, size_t ny, float* field, float defaultVal, unsigned short repeat, char setWeight, size_t* nChanged)
{
    const size_t total = nx * ny;
    if (total == 0) return ORC_OK;
    size_t nUndef = 0;
    for (size_t i = 0; i < total; ++i) if (isnan(field[i])) nUndef++;
    *nChanged = nUndef;
    return creepfill_impl(nx, ny, field, defaultVal, repeat, setWeight, nUndef);
}

/* ----------------------------------------------------------- axis positions */
/* three-way compare in axis order; dir = +1 ascending axis, -1 descending (:104-117) */
static int axis_compare(double key, double elem, int dir)
{
    int c = (key > elem) ? 1 : ((key == elem) ? 0 : -1);
    return dir * c;
}

/* src/interpolation.c:124-146: index if found, else -(insertion point) - 1 */
static int axis_search(double key, const double* axis, int num, int dir)
{
    int lo = 0, hi = num - 1, mid = 0, c = 0;
    while (lo <= hi) {
        mid = (lo + hi) / 2;
        c = axis_compare(key, axis[mid], dir);
        if (c > 0) lo = mid + 1;
        else if (c < 0) hi = mid - 1;
        else break;
    }
    if (c == 0) return mid;
    return (c > 0) ? -(mid + 1) - 1 : -mid - 1;
}

/* src/interpolation.c:148-217 */
int orc_points2position(double* points, size_t n, const double* axis, int num, int axis_type)
{
    const int dir = (axis[0] < axis[num - 1]) ? 1 : -1; /* :152-153 */
    int circular = 0;
    if (axis_type == ORC_LONGITUDE) {
        if (axis[0] < 0 || axis[num - 1] < 0) { /* axis is -180..180 :157-161 */
            for (size_t i = 0; i < n; ++i) if (points[i] > ORC_PI) points[i] -= 2 * ORC_PI;
        } else {                                 /* axis is 0..360 :163-166 */
            for (size_t i = 0; i < n; ++i) if (points[i] < 0) points[i] += 2 * ORC_PI;
        }
        double next = axis[num - 1] + (axis[1] - axis[0]) * 1.01; /* :168 */
        if (dir > 0) { next -= 2 * ORC_PI; if (next >= axis[0]) circular = 1; } /* :169-173 */
        else         { next += 2 * ORC_PI; if (next <= axis[0]) circular = 1; } /* :174-178 */
    }
    for (size_t i = 0; i < n; ++i) {
        if (!isfinite(points[i])) { points[i] = -999.; continue; } /* :183-186 */
        int pos = axis_search(points[i], axis, num, dir);
        if (pos >= 0) { points[i] = (double)pos; continue; }
        int np = -1 * (pos + 1);     /* :192 */
        if (np == num) np--;         /* extrapolate right :193-194 */
        else if (np == 0) np++;      /* extrapolate left :195-197 */
        double slope = axis[np] - axis[np - 1];       /* :199 */
        double offset = axis[np] - (slope * np);      /* :200 */
        double ap = (points[i] - offset) / slope;     /* :201 */
        if (circular && ap <= -0.5) ap += num;        /* :202-204 */
        if (circular && ap > (num - 0.5)) ap -= num;  /* :205-207 */
        points[i] = ap;
    }
    return ORC_OK;
}

/* --------------------------------------------------------- fill value <-> NaN */
/* src/interpolation.c:1775-1783 */
size_t orc_bad2nanf(float* p, float* end, float badVal)
{
    if (!isnan(badVal)) for (; p != end; ++p) if (*p == badVal) *p = orc_nanf();
    return 0;
}

/* src/interpolation.c:1785-1793 */
size_t orc_nanf2bad(float* p, float* end, float badVal)
{
    if (!isnan(badVal)) for (; p != end; ++p) if (isnan(*p)) *p = badVal;
    return 0;
}

/* ------------------------------------------------------- typed slice edges (n1) */
size_t orc_cdm_type_size(int type)
{
    switch (type) {
    case ORC_CDM_CHAR: case ORC_CDM_UCHAR: return 1;
    case ORC_CDM_SHORT: case ORC_CDM_USHORT: return 2;
    case ORC_CDM_INT: case ORC_CDM_UINT: case ORC_CDM_FLOAT: return 4;
    case ORC_CDM_DOUBLE: case ORC_CDM_INT64: case ORC_CDM_UINT64: return 8;
    default: return 0;
    }
}

/* src/CDMInterpolator.cc:115-119.  Data::asFloat() is ArrayTypeConverter<float, C> (src/DataImpl.h:99,132,384-389):
 * data_caster<float, C> = static_cast<float> per element (include/fimex/Utils.h:94-116, no rounding step since the
 * target is not an integer type); then mifi_bad2nanf(begin, end, badValue) with the double fill value converted to the
 * float parameter (src/interpolation.c:1775-1783). */
int orc_data2interpolation_array(const void* in, int type, size_t n, double badValue, float* out)
{
#define ORC_AS_FLOAT(T) { const T* p = (const T*)in; for (size_t i = 0; i < n; ++i) out[i] = (float)p[i]; } break
    switch (type) {
    case ORC_CDM_CHAR: ORC_AS_FLOAT(char);
    case ORC_CDM_SHORT: ORC_AS_FLOAT(short);
    case ORC_CDM_INT: ORC_AS_FLOAT(int);
    case ORC_CDM_FLOAT: ORC_AS_FLOAT(float);
    case ORC_CDM_DOUBLE: ORC_AS_FLOAT(double);
    case ORC_CDM_UCHAR: ORC_AS_FLOAT(unsigned char);
    case ORC_CDM_USHORT: ORC_AS_FLOAT(unsigned short);
    case ORC_CDM_UINT: ORC_AS_FLOAT(unsigned int);
    case ORC_CDM_INT64: ORC_AS_FLOAT(long long);
    case ORC_CDM_UINT64: ORC_AS_FLOAT(unsigned long long);
    default: return ORC_ERROR;
    }
#undef ORC_AS_FLOAT
    orc_bad2nanf(out, out + n, (float)badValue);
    return ORC_OK;
}

/* MetNoFimex::round(double), include/fimex/Utils.h:72-75: int round(double) { return ::lround(num); }.
 * D6: lround of a value outside the range of long is unspecified in C; glibc on x86-64 yields LONG_MIN, kept here. */
static int orc_mifi_round(double num)
{
    long r;
    if (!(fabs(num) < 9223372036854775808.0)) r = (long)(-9223372036854775807LL - 1);
    else r = lround(num);
    return (int)r; /* long -> int: wraps (gcc), like the reference's implicit conversion */
}

/* src/CDMInterpolator.cc:121-124 -> DataImpl<float>::convertDataType(MIFI_UNDEFINED_F, 1., 0., newType, badValue, 1., 0.)
 * (src/DataImpl.h:314-319, 329-348) -> ScaleValue<float, OUT> (include/fimex/Utils.h:444-464):
 *   (in == oldFill || isnan(in)) ? static_cast<OUT>(newFill) : data_caster<OUT, double>()(1.0 * in + 0.0)
 * with oldScale/newScale = 1, (oldOffset - newOffset)/newScale = 0; data_caster rounds through MetNoFimex::round (an int)
 * when OUT is an integer type (Utils.h:94-116).  Note 1.0 * in + 0.0 turns -0.0 into +0.0. */
int orc_interpolation_array2data(const float* in, size_t n, int newType, double badValue, void* out)
{
#define ORC_SCALE_INT(T) { T* p = (T*)out; const T fill = (T)badValue; \
        for (size_t i = 0; i < n; ++i) p[i] = isnan(in[i]) ? fill : (T)orc_mifi_round(1.0 * in[i] + 0.0); } break
#define ORC_SCALE_FLT(T) { T* p = (T*)out; const T fill = (T)badValue; \
        for (size_t i = 0; i < n; ++i) p[i] = isnan(in[i]) ? fill : (T)(1.0 * in[i] + 0.0); } break
    switch (newType) {
    case ORC_CDM_CHAR: ORC_SCALE_INT(char);
    case ORC_CDM_SHORT: ORC_SCALE_INT(short);
    case ORC_CDM_INT: ORC_SCALE_INT(int);
    case ORC_CDM_FLOAT: ORC_SCALE_FLT(float);
    case ORC_CDM_DOUBLE: ORC_SCALE_FLT(double);
    case ORC_CDM_UCHAR: ORC_SCALE_INT(unsigned char);
    case ORC_CDM_USHORT: ORC_SCALE_INT(unsigned short);
    case ORC_CDM_UINT: ORC_SCALE_INT(unsigned int);
    case ORC_CDM_INT64: ORC_SCALE_INT(long long);
    case ORC_CDM_UINT64: ORC_SCALE_INT(unsigned long long);
    default: return ORC_ERROR;
    }
#undef ORC_SCALE_INT
#undef ORC_SCALE_FLT
    return ORC_OK;
}

/* ------------------------------------------------------------- 1-D blends (n4) */
/* src/interpolation.c:1038-1049 */
static void orc_linear_simple(const float* A, const float* B, float* out, size_t n, float f)
{
    for (size_t i = 0; i < n; ++i) out[i] = A[i] + f * (B[i] - A[i]);
}

/* :1050-1063 */
static int orc_linear_f(const float* A, const float* B, float* out, size_t n, double a, double b, double x)
{
    const float f = (a == b) ? 0 : ((x - a) / (b - a));
    if (f == 0) memcpy(out, A, n * sizeof(float));
    else if (f == 1) memcpy(out, B, n * sizeof(float));
    else orc_linear_simple(A, B, out, n, f);
    return ORC_OK;
}

/* :1085-1104 */
static int orc_linear_conf_extrapol_f(float left, float right, const float* A, const float* B, float* out, size_t n, double a, double b, double x)
{
    const float f = (a == b) ? 0 : ((x - a) / (b - a));
    if (f == 0) memcpy(out, A, n * sizeof(float));
    else if (f == 1) memcpy(out, B, n * sizeof(float));
    else if ((f >= left) && (f <= right)) orc_linear_simple(A, B, out, n, f);
    else for (size_t i = 0; i < n; ++i) out[i] = orc_nanf();
    return ORC_OK;
}

/* :1134-1145 */
static int orc_log_f(const float* A, const float* B, float* out, size_t n, double a, double b, double x)
{
    if (a <= 0 || b <= 0 || x <= 0) return ORC_ERROR;
    return orc_linear_f(A, B, out, n, log(a), log(b), log(x));
}

int orc_get_values_1d_f(int kind, const float* A, const float* B, float* out, size_t n, double a, double b, double x)
{
    switch (kind) {
    case ORC_1D_NEAREST: memcpy(out, A, n * sizeof(float)); return ORC_OK;            /* :1030-1034 */
    case ORC_1D_LINEAR: return orc_linear_f(A, B, out, n, a, b, x);
    case ORC_1D_LINEAR_WEAK_EXTRAPOL: return orc_linear_conf_extrapol_f(-1.f, 2.f, A, B, out, n, a, b, x); /* :1106-1109 */
    case ORC_1D_LINEAR_NO_EXTRAPOL: return orc_linear_conf_extrapol_f(0.f, 1.f, A, B, out, n, a, b, x);    /* :1110-1113 */
    case ORC_1D_LINEAR_CONST_EXTRAPOL: {                                                /* :1115-1126 */
        const float f = (a == b) ? 0 : ((x - a) / (b - a));
        if (f >= 1) memcpy(out, B, n * sizeof(float));
        else if (f <= 0) memcpy(out, A, n * sizeof(float));
        else orc_linear_simple(A, B, out, n, f);
        return ORC_OK;
    }
    case ORC_1D_LOG: return orc_log_f(A, B, out, n, a, b, x);
    case ORC_1D_LOG_LOG:                                                                /* :1147-1156 */
        if (a <= 0 || b <= 0 || x <= 0) return ORC_ERROR;
        orc_log_f(A, B, out, n, log(a + M_E), log(b + M_E), log(x + M_E));              /* the inner status is dropped there too */
        return ORC_OK;
    default: return ORC_ERROR;
    }
}

/* :1065-1083 */
int orc_get_values_linear_d(const double* A, const double* B, double* out, size_t n, double a, double b, double x)
{
    const double f = (a == b) ? 0 : ((x - a) / (b - a));
    if (f == 0) memcpy(out, A, n * sizeof(double));
    else if (f == 1) memcpy(out, B, n * sizeof(double));
    else for (size_t i = 0; i < n; ++i) out[i] = A[i] + f * (B[i] - A[i]);
    return ORC_OK;
}

/* -------------------------------------- coordinate-based nearest neighbour plans (n3) */
/* getGridDistance, src/CDMInterpolator.cc:1069-1141 */
double orc_get_grid_distance(const double* lonVals, const double* latVals, size_t orgX, size_t orgY)
{
    const size_t n = orgX * orgY;
    size_t steps, stepSize;
    if (n > 1000) { steps = 53; stepSize = n / steps; } else { stepSize = 1; steps = n; }
    double minOfMax = 2;
    int any = 0;
    for (size_t ik = 0; ik < steps; ++ik) {
        const size_t samplePos = ik * stepSize;
        const double lon0 = lonVals[samplePos], lat0 = latVals[samplePos];
        if (isnan(lon0) || isnan(lat0)) continue;
        double min_cos_d = -2;
        for (size_t pos = 0; pos < n; ++pos) {
            if (pos == samplePos) continue;
            const double lon1 = lonVals[pos], lat1 = latVals[pos];
            if (isnan(lon1) || isnan(lat1)) continue;
            const double dlon = lon0 - lon1;
            const double cos_d = cos(lat0) * cos(lat1) * cos(dlon) + sin(lat0) * sin(lat1); /* :1103 */
            if (cos_d > min_cos_d) min_cos_d = cos_d;
        }
        if (min_cos_d < minOfMax) minOfMax = min_cos_d; /* min_element of the samples :1136 */
        any = 1;
    }
    if (!any) return nan("");
    double d = acos(minOfMax);
    d *= 1.414;
    if (d > ORC_PI) d = ORC_PI;
    return d;
}

typedef struct { double lat, lon, x, y; } orc_ll_point;
static int orc_ll_cmp(const void* a, const void* b)
{
    const double la = ((const orc_ll_point*)a)->lat, lb = ((const orc_ll_point*)b)->lat;
    return (la > lb) - (la < lb);
}

/* fastTranslatePointsToClosestInputCell, :1158-1217: latitude-sorted list, walk up and down from the query's latitude
 * while the latitude difference alone does not exceed the best distance so far.  (qsort instead of std::sort: the order
 * of cells with equal latitude is unspecified in both.) */
int orc_fast_translate_points(double* pointsX, double* pointsY, size_t nPoints, const double* lonVals, const double* latVals, size_t orgX, size_t orgY)
{
    const double max_grid_d = orc_get_grid_distance(lonVals, latVals, orgX, orgY);
    if (isnan(max_grid_d)) return ORC_ERROR;
    const double min_grid_cos_d = cos(max_grid_d);
    orc_ll_point* ll = (orc_ll_point*)malloc((orgX * orgY + 1) * sizeof(orc_ll_point));
    if (!ll) return ORC_ERROR;
    size_t m = 0;
    for (size_t ix = 0; ix < orgX; ++ix)
        for (size_t iy = 0; iy < orgY; ++iy) { /* :1169-1176, x-major as the reference */
            const size_t pos = ix + iy * orgX;
            if (!(isnan(lonVals[pos]) || isnan(latVals[pos]))) {
                ll[m].lat = latVals[pos]; ll[m].lon = lonVals[pos]; ll[m].x = (double)ix; ll[m].y = (double)iy; ++m;
            }
        }
    qsort(ll, m, sizeof(orc_ll_point), orc_ll_cmp);
#pragma omp parallel for schedule(dynamic, 256)
    for (long i = 0; i < (long)nPoints; ++i) {
        const double plat = pointsY[i], plon = pointsX[i];
        double px = -1., py = -1.;
        double min_cos_d = min_grid_cos_d, min_d = acos(min_cos_d);
        size_t lb = 0, hi = m; /* lower_bound on lat */
        while (lb < hi) { const size_t mid = (lb + hi) / 2; if (ll[mid].lat < plat) lb = mid + 1; else hi = mid; }
        for (size_t it = lb; it < m; ++it) { /* :1191-1208 */
            const double dlon = ll[it].lon - plon;
            if (fabs(ll[it].lat - plat) > min_d) break;
            const double cos_d = cos(ll[it].lat) * cos(plat) * cos(dlon) + sin(ll[it].lat) * sin(plat);
            if (cos_d > min_cos_d) { min_cos_d = cos_d; min_d = acos(min_cos_d); px = ll[it].x; py = ll[it].y; }
        }
        for (size_t it2 = lb; it2 > 0; ) { /* :1210-1228 */
            --it2;
            const double dlon = ll[it2].lon - plon;
            if (fabs(ll[it2].lat - plat) > min_d) break;
            const double cos_d = cos(ll[it2].lat) * cos(plat) * cos(dlon) + sin(ll[it2].lat) * sin(plat);
            if (cos_d > min_cos_d) { min_cos_d = cos_d; min_d = acos(min_cos_d); px = ll[it2].x; py = ll[it2].y; }
        }
        pointsY[i] = py;
        pointsX[i] = px;
    }
    free(ll);
    return ORC_OK;
}

/* flannTranslatePointsToClosestInputCell, :991-1067.  nanoflann (include/nanoflann/nanoflann.hpp, vendored in the
 * reference) returns the matches inside the squared radius sorted by distance; the first one is the cell with the smallest
 * kdtree_distance (:966-972) -- found here by exhaustive search, which defines the same result except among exactly
 * equidistant cells (tree order there, lowest index here). */
int orc_flann_translate_points(double maxDist, double* pointsX, double* pointsY, size_t nPoints, const double* lonVals, const double* latVals,
                               size_t orgX, size_t orgY)
{
    if (!(maxDist > 0)) return ORC_ERROR;
    maxDist /= 6371000.; /* MIFI_EARTH_RADIUS_M */
    const size_t n = orgX * orgY;
    double* pts = (double*)malloc(3 * (n + 1) * sizeof(double));
    if (!pts) return ORC_ERROR;
    for (size_t pos = 0; pos < n; ++pos) {
        if (!(isnan(latVals[pos]) || isnan(lonVals[pos]))) {
            const double sinLat = sin(latVals[pos]), cosLat = cos(latVals[pos]), sinLon = sin(lonVals[pos]), cosLon = cos(lonVals[pos]);
            pts[3 * pos] = cosLat * cosLon; pts[3 * pos + 1] = cosLat * sinLon; pts[3 * pos + 2] = sinLat;
        } else {
            pts[3 * pos] = pts[3 * pos + 1] = pts[3 * pos + 2] = nan("");
        }
    }
    const double search_radius = maxDist * maxDist;
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < (long)nPoints; ++i) {
        const double sinLat = sin(pointsY[i]), cosLat = cos(pointsY[i]), sinLon = sin(pointsX[i]), cosLon = cos(pointsX[i]);
        const double q0 = cosLat * cosLon, q1 = cosLat * sinLon, q2 = sinLat;
        double best = search_radius;
        size_t bestPos = n;
        for (size_t pos = 0; pos < n; ++pos) {
            const double d0 = q0 - pts[3 * pos], d1 = q1 - pts[3 * pos + 1], d2 = q2 - pts[3 * pos + 2];
            const double d = d0 * d0 + d1 * d1 + d2 * d2;
            if (d < best) { best = d; bestPos = pos; } /* addPoint: dist < radius; NaN never */
        }
        if (bestPos < n) { pointsX[i] = (double)(bestPos % orgX); pointsY[i] = (double)(bestPos / orgX); }
        else { pointsX[i] = -1000; pointsY[i] = -1000; }
    }
    free(pts);
    return ORC_OK;
}
