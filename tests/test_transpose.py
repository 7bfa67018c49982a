"""Stand-alone transposition: sventt_transpose /
sventt_transpose_inplace against numpy, with the self-checks the reference's own
benchmark uses (tests/bench-transpose.cpp:17-103: 2-D iota in, 0x55 fill, transpose
back and compare; padded leading dimensions; in-place square)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def eng():
    import sve_ntt_amd
    return sve_ntt_amd


def iota_2d(rows, cols, ld, start):
    """row r holds start + r*cols + (0..cols-1), padding words 0xaa.. (bench-transpose.cpp:44)"""
    buf = np.full(ld * (rows - 1) + cols, 0xAAAAAAAAAAAAAAAA, dtype=np.uint64)
    for r in range(rows):
        buf[r * ld:r * ld + cols] = np.arange(start + r * cols, start + (r + 1) * cols, dtype=np.uint64)
    return buf


def dev(a):
    return torch.from_numpy(a.view(np.int64)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("rows,cols,pad_src,pad_dst", [
    (64, 64, 0, 0), (256, 512, 0, 0), (512, 256, 32, 0), (2048, 8192, 0, 32), (4096, 4096, 32, 32),
    (1, 1, 0, 0), (1, 700, 3, 0), (700, 1, 0, 5), (65, 63, 1, 1), (1000, 333, 7, 9), (8, 8, 0, 0)])
def test_out_of_place_matches_numpy(eng, rows, cols, pad_src, pad_dst):
    ld_src, ld_dst = cols + pad_src, rows + pad_dst
    src = iota_2d(rows, cols, ld_src, 0x0123456789ABCDEF)
    dst = np.full(ld_dst * (cols - 1) + rows, 0x5555555555555555, dtype=np.uint64)
    d = dev(dst)
    eng.transpose(d, dev(src), rows, cols, ld_dst, ld_src)
    got = host(d)
    want = dst.copy()
    for c in range(cols):
        want[c * ld_dst:c * ld_dst + rows] = src[c:c + ld_src * (rows - 1) + 1:ld_src]
    assert np.array_equal(got, want)  # padding words of dst untouched
    # and back again, as the reference's benchmark verifies itself
    back = dev(np.full_like(src, 0xAAAAAAAAAAAAAAAA))
    eng.transpose(back, d, cols, rows, ld_src, ld_dst)
    assert np.array_equal(host(back), src)


@pytest.mark.parametrize("dim", [1, 8, 64, 100, 1024, 4096, 4097])
def test_in_place_square(eng, dim):
    a = np.arange(dim * dim, dtype=np.uint64) + np.uint64(0xFEDCBA9800000000)
    d = dev(a)
    eng.transpose_inplace(d, dim)
    assert np.array_equal(host(d).reshape(dim, dim), a.reshape(dim, dim).T)
    eng.transpose(d, d, dim, dim, dim, dim)  # dst == src routes to the in-place kernel
    assert np.array_equal(host(d), a)


def test_host_pointers_and_errors(eng):
    rows, cols = 96, 160
    src = iota_2d(rows, cols, cols + 4, 7)
    dst = np.full((rows + 2) * (cols - 1) + rows, 0x5555555555555555, dtype=np.uint64)
    eng.transpose(dst, src, rows, cols, rows + 2, cols + 4)  # numpy arrays: staged through the device
    for c in (0, 1, cols - 1):
        assert np.array_equal(dst[c * (rows + 2):c * (rows + 2) + rows], src[c::cols + 4][:rows])
    assert dst[rows] == 0x5555555555555555
    sq = np.arange(64 * 64, dtype=np.uint64)
    eng.transpose_inplace(sq, 64)
    assert np.array_equal(sq.reshape(64, 64).T.ravel(), np.arange(64 * 64, dtype=np.uint64))
    d = dev(np.zeros(64 * 32, dtype=np.uint64))
    with pytest.raises(ValueError):
        eng.transpose(d, d, 64, 32, 64, 32)  # in place needs a square matrix
    with pytest.raises(ValueError):
        eng.transpose(d, dev(np.zeros(64 * 32, dtype=np.uint64)), 64, 32, 63, 32)  # ld_dst < rows
    with pytest.raises(ValueError):
        eng.transpose(d[8:], d, 8, 8, 8, 8)  # overlap
    with pytest.raises(ValueError):
        eng.transpose(np.zeros(64, dtype=np.uint64), dev(np.zeros(64, dtype=np.uint64)), 8, 8, 8, 8)
