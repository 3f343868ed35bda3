"""The geometry of the creep fill by rectangles (fimex_amd/csrc/creep_rects.hpp, used by run_creepfill in fill.hip) is host code: compiled
here on its own with g++ and checked on random masks.  What run_creepfill relies on: every undefined cell lies in exactly one rectangle;
a rectangle's outermost rows and columns hold no undefined cell unless they are the field's own border; rectangles are at least four
cells each way where the field allows."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WRAPPER = r"""
#include "creep_rects.hpp"
extern "C" int creep_rects_of(uint32_t nx, uint32_t ny, uint32_t words, const uint32_t* bits, uint32_t* out, int cap, int* worthIt)
{
    std::vector<fimex_amd::creep_rects::Rect> r;
    *worthIt = fimex_amd::creep_rects::slice_rects(bits, nx, ny, words, r) ? 1 : 0;
    int n = 0;
    for (const auto& q : r) {
        if (n < cap) { out[4 * n] = q.xa; out[4 * n + 1] = q.xb; out[4 * n + 2] = q.ya; out[4 * n + 3] = q.yb; }
        ++n;
    }
    return n;
}
"""


@pytest.fixture(scope="module")
def lib():
    d = tempfile.mkdtemp(prefix="creep_rects_")
    src = os.path.join(d, "wrap.cc")
    with open(src, "w") as f:
        f.write(WRAPPER)
    so = os.path.join(d, "libcreep_rects.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "fimex_amd", "csrc"), src, "-o", so], check=True)
    return ctypes.CDLL(so)


def rects_of(lib, mask):
    ny, nx = mask.shape
    words = (nx + 63) // 64 * 2
    bits = np.zeros((ny, words * 32), dtype=bool)
    bits[:, :nx] = mask
    packed = np.packbits(bits.reshape(ny, words, 32), axis=2, bitorder="little").view(np.uint32).reshape(ny, words).copy()
    out = np.zeros(4 * 4096, dtype=np.uint32)
    worth = ctypes.c_int(0)
    n = lib.creep_rects_of(nx, ny, words, packed.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                           4096, ctypes.byref(worth))
    return [tuple(int(v) for v in out[4 * k:4 * k + 4]) for k in range(min(n, 4096))], bool(worth.value)


def check(mask, rects):
    ny, nx = mask.shape
    cover = np.zeros(mask.shape, dtype=np.int32)
    for xa, xb, ya, yb in rects:
        assert 0 <= xa <= xb < nx and 0 <= ya <= yb < ny
        assert xb - xa + 1 >= min(4, nx) and yb - ya + 1 >= min(4, ny), (xa, xb, ya, yb)
        cover[ya:yb + 1, xa:xb + 1] += mask[ya:yb + 1, xa:xb + 1]
        # the ring: defined throughout, or the field's border
        if ya > 0:
            assert not mask[ya, xa:xb + 1].any(), ("top", xa, xb, ya, yb)
        if yb < ny - 1:
            assert not mask[yb, xa:xb + 1].any(), ("bottom", xa, xb, ya, yb)
        if xa > 0:
            assert not mask[ya:yb + 1, xa].any(), ("left", xa, xb, ya, yb)
        if xb < nx - 1:
            assert not mask[ya:yb + 1, xb].any(), ("right", xa, xb, ya, yb)
    assert np.array_equal(cover, mask.astype(np.int32)), "an undefined cell outside every rectangle, or inside two"


def random_mask(rng, nx, ny, regions, specks, borders):
    m = np.zeros((ny, nx), dtype=bool)
    for _ in range(regions):
        w, h = int(rng.integers(1, max(2, nx // 3))), int(rng.integers(1, max(2, ny // 3)))
        x, y = int(rng.integers(0, nx - w + 1)), int(rng.integers(0, ny - h + 1))
        if rng.random() < 0.3:   # a wedge instead of a block
            yy, xx = np.mgrid[0:h, 0:w]
            m[y:y + h, x:x + w] |= (yy * w + xx * h) < w * h // 2
        else:
            m[y:y + h, x:x + w] = True
    for _ in range(specks):
        m[int(rng.integers(0, ny)), int(rng.integers(0, nx))] = True
    for _ in range(borders):
        side = int(rng.integers(0, 4))
        if side == 0: m[0, int(rng.integers(0, nx))] = True
        elif side == 1: m[ny - 1, int(rng.integers(0, nx))] = True
        elif side == 2: m[int(rng.integers(0, ny)), 0] = True
        else: m[int(rng.integers(0, ny)), nx - 1] = True
    return m


@pytest.mark.parametrize("shape", [(64, 64), (97, 131), (300, 77), (65, 400), (513, 258)])
def test_rectangles_cover_every_undefined_cell_once_and_keep_a_defined_ring(lib, shape):
    nx, ny = shape
    rng = np.random.default_rng(nx * 1000 + ny)
    seen_worth = 0
    for trial in range(120):
        m = random_mask(rng, nx, ny, regions=int(rng.integers(0, 5)), specks=int(rng.integers(0, 6)), borders=int(rng.integers(0, 4)))
        rects, worth = rects_of(lib, m)
        if not m.any():
            assert rects == [] and not worth
            continue
        check(m, rects)
        seen_worth += worth
        if worth:
            assert len(rects) <= 64 and sum((xb - xa + 1) * (yb - ya + 1) for xa, xb, ya, yb in rects) * 2 <= nx * ny
    assert seen_worth > 10


def test_rectangles_of_the_usual_shapes(lib):
    """scattered holes: one rectangle, the whole field, not worth cutting; regions one defined row apart stay apart; a region in a
    corner keeps the field's border as its ring."""
    rng = np.random.default_rng(3)
    m = rng.random((200, 300)) < 0.3
    rects, worth = rects_of(lib, m)
    check(m, rects)
    assert rects == [(0, 299, 0, 199)] and not worth
    m = np.zeros((200, 300), dtype=bool)
    m[50:60, 40:80] = True
    m[61:70, 45:90] = True     # row 60 is defined throughout between them
    m[:30, :20] = True         # the upper left corner
    rects, worth = rects_of(lib, m)
    check(m, rects)
    assert worth and sorted(rects) == sorted([(0, 20, 0, 30), (39, 80, 49, 60), (44, 90, 60, 70)])
