"""neuray_amd/imgproc.py: closed-form properties of the OpenCV routines it restates (cv2 is not available to compare with:
the module header says which of them are exact restatements and which may differ by one grey level)."""
import numpy as np

from neuray_amd import imgproc as I


def test_gaussian_kernel_and_blur():
    k = I.gaussian_kernel(7, 4 / 3)
    assert abs(k.sum() - 1) < 1e-15 and np.allclose(k, k[::-1]) and k[3] == k.max()
    const = np.full((9, 11, 3), 137, np.uint8)
    assert np.array_equal(I.gaussian_blur(const, 5, 1.1), const)                 # a normalised kernel keeps constants
    ramp = np.tile(np.arange(40, dtype=np.float32)[None, :, None], (6, 1, 1))
    out = I.gaussian_blur(ramp, 5, 1.0)
    assert np.allclose(out[:, 2:-2], ramp[:, 2:-2], atol=1e-5)                    # symmetric kernel: linear ramps pass unchanged
    # reflect-101 border: sample -1 is sample 1 (not 0)
    imp = np.zeros((1, 8, 1), np.float32); imp[0, 1] = 1.0
    k3 = I.gaussian_kernel(3, 0.8)
    assert np.isclose(I.gaussian_blur(imp, 3, 0.8)[0, 0, 0], 2 * k3[0], atol=1e-6)
    # the reference's choice of kernel for a 2x / 4x reduction (utils/base_utils.py:128-134)
    x = np.random.RandomState(0).randint(0, 256, (16, 16, 3)).astype(np.uint8)
    assert np.array_equal(I.downsample_gaussian_blur(x, 0.5), I.gaussian_blur(x, 3, 2 / 3))
    assert np.array_equal(I.downsample_gaussian_blur(x, 0.25), I.gaussian_blur(x, 7, 4 / 3))


def test_resize_nearest_area_linear():
    a = np.arange(8 * 12, dtype=np.float32).reshape(8, 12)
    assert np.array_equal(I.resize(a, (6, 4), I.INTER_NEAREST), a[::2, ::2])      # floor(x * 2)
    assert np.array_equal(I.resize(a, (12, 8), I.INTER_NEAREST), a)
    up = I.resize(a[:2, :3], (6, 4), I.INTER_NEAREST)
    assert np.array_equal(up, np.kron(a[:2, :3], np.ones((2, 2))))
    u8 = np.random.RandomState(1).randint(0, 256, (8, 12, 3)).astype(np.uint8)
    area = I.resize(u8, (3, 2), I.INTER_AREA)
    want = u8.reshape(2, 4, 3, 4, 3).astype(np.float64).mean((1, 3))
    assert np.array_equal(area, np.rint(want).astype(np.uint8))
    # ties: the 8-bit 2 x 2 case is OpenCV's integer fast path (a + b + c + d + 2) >> 2 - halves round UP; every other integer
    # factor goes through cvRound (half to even)
    assert I.resize(np.array([[1, 2], [1, 2]], np.uint8), (1, 1), I.INTER_AREA)[0, 0] == 2          # 1.5 -> 2
    assert I.resize(np.array([[0, 1], [0, 1]], np.uint8), (1, 1), I.INTER_AREA)[0, 0] == 1          # 0.5 -> 1 (2 x 2: up)
    q = np.zeros((4, 4), np.uint8)
    q[0, :] = 2                                                                                     # mean of the 4 x 4 block 0.5
    assert I.resize(q, (1, 1), I.INTER_AREA)[0, 0] == 0                                             # 0.5 -> 0 (4 x 4: even)
    const = np.full((10, 14, 3), 93, np.uint8)
    assert np.array_equal(I.resize(const, (7, 5), I.INTER_LINEAR), np.full((5, 7, 3), 93, np.uint8))
    ramp = np.tile(np.arange(0, 160, 4, dtype=np.float32)[None, :], (4, 1))          # 40 columns, slope 4
    half = I.resize(ramp, (20, 2), I.INTER_LINEAR)
    assert np.allclose(half[0], (np.arange(20) + 0.5) * 2 * 4 - 0.5 * 4)            # value at source coordinate (x + 0.5) * 2 - 0.5
    assert np.array_equal(I.resize(ramp, (40, 4), I.INTER_LINEAR), ramp)


def test_resize_img_shape_and_range():
    x = np.random.RandomState(2).randint(0, 256, (800 // 8, 800 // 8, 3)).astype(np.uint8)
    y = I.resize_img(x, 0.5)
    assert y.shape == (50, 50, 3) and y.dtype == np.uint8
    assert abs(float(y.mean()) - float(x.mean())) < 2.0


def test_decompose_projection_matrix_recovers_the_camera():
    rng = np.random.RandomState(3)
    for _ in range(5):
        K = np.array([[2890.0 + 10 * rng.rand(), 0.3 * rng.rand(), 820 + rng.rand()], [0, 2880 + 10 * rng.rand(), 610 + rng.rand()], [0, 0, 1.0]])
        q, _r = np.linalg.qr(rng.randn(3, 3))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        C = rng.randn(3) * 2
        P = K @ np.concatenate([q, -(q @ C)[:, None]], 1)
        K2, R2, ch = I.decompose_projection_matrix(P)
        assert np.allclose(K2, K, atol=1e-9) and np.allclose(R2, q, atol=1e-12) and np.allclose((ch[:3] / ch[3])[:, 0], C, atol=1e-10)
        assert np.allclose(np.tril(K2, -1), 0) and np.all(np.diag(K2) > 0)
