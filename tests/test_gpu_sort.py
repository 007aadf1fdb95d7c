"""The engine's radix sort (csrc/slk_sort.hip) on the GPU against numpy's stable argsort: the same checks as
tests/test_emu_sort.py plus the sizes the training prep runs (chunks of 2^20-pair segments, both tile shapes), where
look-back chains cross XCDs under load."""
import numpy as np
import pytest

from test_emu_sort import check_sort

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    from hip_backend import HipBackend
    b = HipBackend()
    yield b
    b.close()


@pytest.mark.parametrize('kind', [0, 1, 2])
@pytest.mark.parametrize('n,bits', [(1, 5), (64, 9), (4096, 17), (4097, 12), (9000, 24), (200000, 32), (1 << 20, 21)])
def test_sort_matches_numpy(be, kind, n, bits):
    check_sort(be, kind, n, bits, seed=n + kind)


@pytest.mark.parametrize('cfg', [0, 1])
@pytest.mark.parametrize('kind,bits', [(0, 20), (1, 24)])
def test_sort_segmented_chunk(be, kind, bits, cfg):
    # the training prep's shape: 8 minibatches of 2^20 pairs (the last one short), sorted on the id bits only
    check_sort(be, kind, 7 * (1 << 20) + 300001, bits, seg_len=1 << 20, seed=11, cfg=cfg)
    be.engine.set_option('sort_big_min', 1 << 20)


@pytest.mark.parametrize('cfg', [0, 1])
def test_sort_large_skewed(be, cfg):
    # 80 % of the keys equal: every tile's look-back meets runs of aggregates
    check_sort(be, 0, 5 * (1 << 20) + 17, 23, seed=12, skew=True, cfg=cfg)
    check_sort(be, 1, 3 * (1 << 20), 27, seg_len=1 << 20, seed=13, skew=True, cfg=cfg)
    be.engine.set_option('sort_big_min', 1 << 20)


@pytest.mark.parametrize('n,bits', [(4096, 13), (9000, 16), (1 << 21, 22), (1 << 21, 32)])
def test_sort_clobbering_input(be, n, bits):
    check_sort(be, 0, n, bits, seed=n, clobber=True)
    check_sort(be, 1, n, bits, seed=n + 1, clobber=True)


def test_sort_repeated_under_load(be):
    # the same sort 20 times back to back (stale look-back words of an earlier run must never be read)
    for r in range(20):
        check_sort(be, 0, (1 << 20) + 123 * r, 20, seg_len=1 << 18, seed=100 + r)
