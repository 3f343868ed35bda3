"""Seeded synthetic inputs shared by the parity tests (oracle vs HIP path)."""
import numpy as np

SPECIAL = np.array([np.nan, np.inf, -np.inf, 1e300, -1e300, -999.0, 2.0 ** 31, -(2.0 ** 31) - 1, 2.0 ** 40])


def field(nz, ny, nx, seed, nan_frac=0.01, extremes=True):
    """[nz][ny][nx] float32: smooth + noise, some NaN, and (optionally) inf / -0.0 / denormals."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    base = 280 + 20 * np.sin(0.013 * x) * np.cos(0.021 * y)
    f = (base[None] + rng.normal(0, 1, (nz, ny, nx)) + 0.01 * np.arange(nz)[:, None, None]).astype(np.float32)
    n = f.size
    if nan_frac > 0 and n:
        f.reshape(-1)[rng.choice(n, max(1, int(n * nan_frac)), replace=False)] = np.nan
    if extremes and n > 64:
        idx = rng.choice(n, 12, replace=False)
        f.reshape(-1)[idx[0:2]] = np.inf
        f.reshape(-1)[idx[2:4]] = -np.inf
        f.reshape(-1)[idx[4:6]] = -0.0
        f.reshape(-1)[idx[6:8]] = np.float32(1e-42)   # denormal
        f.reshape(-1)[idx[8:10]] = np.float32(3e38)
        f.reshape(-1)[idx[10:12]] = np.float32(-3e38)
    return f


def backward_positions(inX, inY, outX, outY, seed, overshoot=2.0, special=True):
    """Fractional source positions per output cell: a rotated, slightly warped grid that overshoots the
    source on every side (undefined cells + every border branch), with lround/floor tie cases mixed in."""
    rng = np.random.default_rng(seed)
    j, i = np.meshgrid(np.arange(outY, dtype=np.float64), np.arange(outX, dtype=np.float64), indexing="ij")
    u = i / max(outX - 1, 1) - 0.5
    v = j / max(outY - 1, 1) - 0.5
    ang = 0.17
    ur = np.cos(ang) * u - np.sin(ang) * v
    vr = np.sin(ang) * u + np.cos(ang) * v
    px = (inX - 1) / 2.0 + ur * (inX - 1 + 2 * overshoot) * 1.05 + 0.8 * np.sin(7 * v)
    py = (inY - 1) / 2.0 + vr * (inY - 1 + 2 * overshoot) * 1.05 + 0.8 * np.cos(5 * u)
    px, py = px.ravel(), py.ravel()
    n = px.size
    if special and n >= 64:
        k = max(8, n // 50)
        idx = rng.choice(n, k, replace=False)
        # exact grid points, half-way ties, and the rims of the nearest-neighbour border zone
        xs = np.array([0.0, 0.5, 1.0, 1.5, -0.5, -0.5000001, -0.4999999, -1.0, inX - 1.0, inX - 1.5, inX - 0.5,
                       inX - 0.5000001, inX - 1 + 1e-9, inX - 2.0, inX * 1.0, 2.5, 2.0, 1.999999999])
        ys = np.array([0.0, 0.5, 1.0, 1.5, -0.5, -0.5000001, -0.4999999, -1.0, inY - 1.0, inY - 1.5, inY - 0.5,
                       inY - 0.5000001, inY - 1 + 1e-9, inY - 2.0, inY * 1.0, 2.5, 2.0, 1.999999999])
        px[idx] = rng.choice(xs, k)
        py[idx] = rng.choice(ys, k)
        idx2 = rng.choice(n, min(len(SPECIAL), n // 8), replace=False)
        px[idx2] = SPECIAL[: len(idx2)]
        idx3 = rng.choice(n, min(len(SPECIAL), n // 8), replace=False)
        py[idx3] = SPECIAL[: len(idx3)][::-1]
    return px, py


def coherent_positions(inX, inY, outX, outY, seed, wrap=False, outliers=0, angle=0.12):
    """Fractional source positions of a target grid that is rotated against the source and lies mostly inside it -- the
    shape of a real reprojection, which the LDS-staged kernels tile.  wrap: the source is periodic in x and the target
    straddles its seam (a global longitude axis and a target across the date line: positions jump by inX inside a row);
    outliers: that many isolated cells get positions elsewhere in the source."""
    rng = np.random.default_rng(seed)
    j, i = np.meshgrid(np.arange(outY, dtype=np.float64), np.arange(outX, dtype=np.float64), indexing="ij")
    sx, sy = (inX - 8.0) / (outX * 1.15), (inY - 8.0) / (outY * 1.15)
    u, v = i - outX / 2.0, j - outY / 2.0
    px = inX / 2.0 + sx * (np.cos(angle) * u - np.sin(angle) * v) + 0.3 * np.sin(0.05 * j)
    py = inY / 2.0 + sy * (np.sin(angle) * u + np.cos(angle) * v) + 0.3 * np.cos(0.04 * i)
    if wrap:
        px = np.mod(px + inX * 0.37, inX - 1.0)
    px, py = px.ravel(), py.ravel()
    if outliers:
        idx = rng.choice(px.size, outliers, replace=False)
        px[idx] = rng.uniform(0, inX - 1, outliers)
        py[idx] = rng.uniform(0, inY - 1, outliers)
    return px, py


def forward_positions(inX, inY, outX, outY, seed, density=1.0, special=True):
    """Fractional target positions per SOURCE cell.  density > 1: several source cells per target
    cell (long buckets); < 1: sparse buckets with many empty targets.  Part of the source falls outside."""
    rng = np.random.default_rng(seed)
    j, i = np.meshgrid(np.arange(inY, dtype=np.float64), np.arange(inX, dtype=np.float64), indexing="ij")
    u = i / max(inX - 1, 1) - 0.5
    v = j / max(inY - 1, 1) - 0.5
    ang = -0.23
    ur = np.cos(ang) * u - np.sin(ang) * v
    vr = np.sin(ang) * u + np.cos(ang) * v
    scale = 1.25 / np.sqrt(density)
    px = (outX - 1) / 2.0 + ur * (outX - 1) * scale * 1.1 + 0.5 * np.sin(9 * v)
    py = (outY - 1) / 2.0 + vr * (outY - 1) * scale * 1.1 + 0.5 * np.cos(4 * u)
    px, py = px.ravel(), py.ravel()
    n = px.size
    if special and n >= 64:
        k = max(8, n // 50)
        idx = rng.choice(n, k, replace=False)
        xs = np.array([0.0, 0.5, -0.5, -0.4999999, 1.5, 2.5, outX - 1.0, outX - 0.5, outX - 0.5000001, outX - 1.5, -1.0])
        ys = np.array([0.0, 0.5, -0.5, -0.4999999, 1.5, 2.5, outY - 1.0, outY - 0.5, outY - 0.5000001, outY - 1.5, -1.0])
        px[idx] = rng.choice(xs, k)
        py[idx] = rng.choice(ys, k)
        idx2 = rng.choice(n, min(len(SPECIAL), n // 8), replace=False)
        px[idx2] = SPECIAL[: len(idx2)]
        idx3 = rng.choice(n, min(len(SPECIAL), n // 8), replace=False)
        py[idx3] = SPECIAL[: len(idx3)][::-1]
    return px, py


def rotation_matrix(ox, oy, seed):
    """double[4*ox*oy] = (cos, sin, -sin, phi) of a smooth angle field, as interpolation.c:429-432 stores it."""
    rng = np.random.default_rng(seed)
    j, i = np.meshgrid(np.arange(oy), np.arange(ox), indexing="ij")
    phi = (0.9 * np.sin(0.05 * i) + 0.7 * np.cos(0.03 * j) + rng.normal(0, 0.01, (oy, ox))).ravel()
    phi[:: max(1, phi.size // 17)] *= 4  # some angles beyond +-pi
    m = np.empty((phi.size, 4))
    m[:, 0] = np.cos(phi)
    m[:, 1] = np.sin(phi)
    m[:, 2] = -np.sin(phi)
    m[:, 3] = phi
    return m.ravel()


def holes(nz, ny, nx, seed, frac=0.3, blobs=6):
    """field with NaN blobs (land/sea-mask like) for the fill tests."""
    rng = np.random.default_rng(seed)
    f = field(nz, ny, nx, seed, nan_frac=0.0, extremes=False)
    y, x = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    for z in range(nz):
        mask = np.zeros((ny, nx), bool)
        for _ in range(blobs):
            cy, cx = rng.uniform(0, ny), rng.uniform(0, nx)
            r = rng.uniform(0.05, 0.25) * min(nx, ny) * np.sqrt(frac / 0.3)
            mask |= (y - cy) ** 2 + (x - cx) ** 2 < r * r
        mask |= rng.random((ny, nx)) < 0.01
        f[z][mask] = np.nan
    return f


def same(a, b):
    """bit-exact on defined values, identical NaN positions (NaN payloads may differ between CPU and GPU)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))


def describe_mismatch(a, b, limit=5):
    a = np.asarray(a, dtype=np.float32).ravel()
    b = np.asarray(b, dtype=np.float32).ravel()
    na, nb = np.isnan(a), np.isnan(b)
    bad = (na != nb) | (~na & ~nb & (a.view(np.uint32) != b.view(np.uint32)))
    idx = np.nonzero(bad)[0]
    return "%d of %d differ; first: %s" % (idx.size, a.size, [(int(i), float(a[i]), float(b[i])) for i in idx[:limit]])
