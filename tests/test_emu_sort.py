"""The engine's radix sort (csrc/slk_sort.hip) against numpy's stable argsort, on the fiber emulator (the `-m gpu`
twin is tests/test_gpu_sort.py).  Every shape the training paths use: single tile, several tiles with look-back,
one segment per minibatch, the three key / payload widths, partial last tiles, skewed digits."""
import numpy as np
import pytest

from emu_backend import EmuBackend

KT = {0: (np.uint32, np.uint32), 1: (np.uint32, np.uint64), 2: (np.uint64, np.uint32)}


def check_sort(be, kind, n, bits, seg_len=0, seed=0, skew=False, cfg=None, clobber=False):
    rng = np.random.RandomState(seed)
    kt, vt = KT[kind]
    top = np.iinfo(kt).max
    keys = rng.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(kt)
    if kind == 2:
        keys = keys | (rng.randint(0, 2 ** 32, size=n, dtype=np.uint64) << np.uint64(32))
    if skew:  # most keys share their low digits: long runs, look-back chains of aggregates
        keys = np.where(rng.rand(n) < 0.8, kt(12345), keys).astype(kt)
    vals = rng.randint(0, 2 ** 31, size=n).astype(vt)
    if kind == 1:
        vals = vals | (np.arange(n, dtype=np.uint64) << np.uint64(33))
    k_in, v_in = be.alloc(keys), be.alloc(vals)
    k_out, v_out = be.alloc(np.zeros(n, kt)), be.alloc(np.zeros(n, vt))
    if cfg is not None:  # 1: the large sorts' tiles (512 threads x 16 keys) whatever the size, 0: the small sorts' (256 x 16)
        be.engine.set_option('sort_big_min', 1 if cfg else 1 << 62)
    be.engine.probe_sort(kind + (8 if clobber else 0), be.ptr(k_in), be.ptr(k_out), be.ptr(v_in), be.ptr(v_out), n, bits, seg_len=seg_len)
    mask = kt(top if bits >= 8 * keys.itemsize else (1 << bits) - 1)
    got_k, got_v = be.get(k_out), be.get(v_out)
    assert clobber or (np.array_equal(be.get(k_in), keys) and np.array_equal(be.get(v_in), vals))  # inputs intact
    seg = seg_len if seg_len else max(n, 1)
    for s0 in range(0, n, seg):
        s1 = min(n, s0 + seg)
        order = np.argsort(keys[s0:s1] & mask, kind='stable')
        assert np.array_equal(got_k[s0:s1], keys[s0:s1][order]), (kind, n, bits, seg_len, s0)
        assert np.array_equal(got_v[s0:s1], vals[s0:s1][order]), (kind, n, bits, seg_len, s0)


@pytest.fixture(scope='module')
def be():
    b = EmuBackend()
    yield b
    b.close()


@pytest.mark.parametrize('kind', [0, 1, 2])
@pytest.mark.parametrize('n,bits', [(1, 5), (63, 8), (64, 9), (1000, 20), (4096, 17), (4097, 12), (9000, 24), (20000, 32)])
def test_sort_matches_numpy(be, kind, n, bits):
    check_sort(be, kind, n, bits, seed=n + kind)


def test_sort_u64_high_bits(be):
    check_sort(be, 2, 6000, 47, seed=3)
    check_sort(be, 2, 300, 64, seed=4)


@pytest.mark.parametrize('kind', [0, 1])
def test_sort_segmented(be, kind):
    # segments sorted on their own; a short last segment; tiles that end inside a segment
    check_sort(be, kind, 3 * 5000 + 123, 13, seg_len=5000, seed=7)
    check_sort(be, kind, 2 * 4096, 10, seg_len=4096, seed=8)
    # a segmented request small enough for the one-tile form keeps its segments (it was sorted as one array)
    check_sort(be, kind, 4005, 9, seg_len=100, seed=9)
    check_sort(be, kind, 4096, 12, seg_len=4096, seed=10)


def test_sort_skewed_digits(be):
    check_sort(be, 0, 15000, 20, seed=9, skew=True)
    check_sort(be, 1, 9000, 16, seg_len=4500, seed=10, skew=True)


@pytest.mark.parametrize('n,bits', [(4096, 13), (4096, 8), (9000, 16), (9000, 17), (20000, 32), (5000, 24)])
def test_sort_clobbering_input(be, n, bits):
    # even and odd pass counts with the input as the second buffer pair (the shuffle's and the bloom lists' sorts)
    check_sort(be, 0, n, bits, seed=n, clobber=True)
    check_sort(be, 1, n, bits, seed=n + 1, clobber=True)


@pytest.mark.parametrize('cfg', [0, 1])
def test_sort_big_tile_shapes_at_small_sizes(be, cfg):
    """The large sorts' tile shapes (512 x 16) forced onto sizes the emulator can hold:
    several tiles with look-back, segments, a short last tile, skewed digits, both payload widths."""
    try:
        check_sort(be, 0, 20000, 20, seed=21, cfg=cfg)
        check_sort(be, 1, 3 * 9000 + 55, 13, seg_len=9000, seed=22, cfg=cfg)
        check_sort(be, 0, 16000, 24, seed=23, skew=True, cfg=cfg)
        check_sort(be, 2, 8000, 40, seed=24, cfg=cfg)
        check_sort(be, 1, 15000, 32, seed=25, cfg=cfg, clobber=True)
    finally:
        be.engine.set_option('sort_big_min', 1 << 20)
