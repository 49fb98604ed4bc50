"""Numpy restatements of the few OpenCV routines the reference's dataset layer calls while it READS a scene (host pipeline,
SURVEY.md 8(f) f-4): the image is not in this container (no cv2, no network), so the algorithms are restated from OpenCV's
published behaviour and used by neuray_amd/database.py for

    cv2.GaussianBlur(img, (k, k), sigma, borderType=BORDER_REFLECT101)   utils/base_utils.py:128-134 downsample_gaussian_blur
    cv2.resize(img, (w, h), interpolation=INTER_LINEAR)                  utils/base_utils.py:535-540 resize_img; database.py:203
    cv2.resize(..., interpolation=INTER_NEAREST)                         database.py:236,244,341,349 (masks, depth maps)
    cv2.resize(..., interpolation=INTER_AREA)                            database.py:96 (the LLFF cache, integer factor 4 / 8)
    cv2.decomposeProjectionMatrix(P)                                     database.py:166 (DTU cameras.npz world_mat_i)

PARITY UNPINNED against real OpenCV (nothing here could be run against cv2): the nearest / area resamplers and the projection-
matrix decomposition are exact restatements (integer index arithmetic, exact box means with cvRound's round-half-to-even,
a unique RQ factorisation); the Gaussian blur and the bilinear resize are evaluated in float64 and rounded half-up where
OpenCV's 8-bit paths use fixed-point weights (8 fractional bits for the blur, 11 for the resize), so an 8-bit result may differ
from OpenCV's by one grey level.  tests/test_imgproc.py checks the closed-form properties; tests/test_database.py runs the
REFERENCE's own database classes on top of these functions (as its `cv2`) and compares accessor by accessor with the
adapters."""
import numpy as np

INTER_NEAREST, INTER_LINEAR, INTER_AREA = 0, 1, 3


def gaussian_kernel(ksize, sigma):
    """cv2.getGaussianKernel for sigma > 0: exp(-(i - (ksize - 1) / 2)^2 / (2 sigma^2)), normalised to sum 1"""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def _reflect101(idx, n):
    """BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba (the edge sample is not repeated)"""
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.mod(idx, period)
    return np.where(idx >= n, period - idx, idx)


def _round_u8(a):
    return np.clip(np.floor(a + 0.5), 0, 255).astype(np.uint8)


def gaussian_blur(img, ksize, sigma):
    """separable Gaussian blur with reflect-101 borders; uint8 in -> uint8 out (float64 accumulation, round half up), float in ->
    float32 out"""
    k = gaussian_kernel(ksize, sigma)
    a = np.asarray(img)
    x = a.astype(np.float64)
    try:                                    # scipy's C loop when it is there ('mirror' = reflect-101); same sums, ~30x faster
        from scipy.ndimage import correlate1d
        for axis in (0, 1):
            x = correlate1d(x, k, axis=axis, mode='mirror')
    except ImportError:
        r = ksize // 2
        for axis in (0, 1):
            n = x.shape[axis]
            idx = _reflect101(np.arange(-r, n + r), n)
            padded = np.take(x, idx, axis=axis)
            out = np.zeros_like(x)
            for j in range(ksize):
                out += k[j] * np.take(padded, np.arange(j, j + n), axis=axis)
            x = out
    return _round_u8(x) if a.dtype == np.uint8 else x.astype(np.float32)


def downsample_gaussian_blur(img, ratio):
    """utils/base_utils.py:128-134"""
    sigma = (1 / ratio) / 3
    ksize = int(np.ceil(((sigma - 0.8) / 0.3 + 1) * 2 + 1))
    ksize = ksize + 1 if ksize % 2 == 0 else ksize
    return gaussian_blur(img, ksize, sigma)


def resize(img, dsize, interpolation=INTER_LINEAR):
    """cv2.resize(img, (w, h), interpolation=...) for the three modes the reference uses"""
    a = np.asarray(img)
    w, h = int(dsize[0]), int(dsize[1])
    sh, sw = a.shape[:2]
    if interpolation == INTER_NEAREST:
        # resizeNN: sx = min(floor(x * (src_w / dst_w)), src_w - 1), the scale taken in double
        ys = np.minimum(np.floor(np.arange(h) * (sh / h)).astype(np.int64), sh - 1)
        xs = np.minimum(np.floor(np.arange(w) * (sw / w)).astype(np.int64), sw - 1)
        return a[ys][:, xs]
    if interpolation == INTER_AREA:
        if sh % h or sw % w:
            raise NotImplementedError("neuray_amd.imgproc: INTER_AREA with a non-integer factor (%d x %d -> %d x %d)" % (sh, sw, h, w))
        fy, fx = sh // h, sw // w
        x = a.astype(np.float64).reshape((h, fy, w, fx) + a.shape[2:]).sum((1, 3)) / (fy * fx)
        if a.dtype != np.uint8:
            return x.astype(a.dtype)
        if fy == 2 and fx == 2:
            # OpenCV's 8-bit 2 x 2 fast path (ResizeAreaFastVec) is integer: (a + b + c + d + 2) >> 2, i.e. halves round UP
            # (from the OpenCV sources as remembered - unpinned against a real cv2, like the rest of this module)
            return np.clip(np.floor(x + 0.5), 0, 255).astype(np.uint8)
        return np.clip(np.rint(x), 0, 255).astype(np.uint8)      # cvRound (saturate_cast): half to even
    if interpolation != INTER_LINEAR:
        raise NotImplementedError(interpolation)
    # bilinear, half-pixel centres: src = (dst + 0.5) * scale - 0.5; taps clamped at the borders (resize.cpp resizeGeneric_)
    def taps(n_dst, n_src):
        f = (np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(f).astype(np.int64)
        t = f - i0
        t = np.where(i0 < 0, 0.0, t)
        i0 = np.maximum(i0, 0)
        t = np.where(i0 >= n_src - 1, 0.0, t)
        i0 = np.minimum(i0, n_src - 1)
        return i0, np.minimum(i0 + 1, n_src - 1), t
    y0, y1, ty = taps(h, sh)
    x0, x1, tx = taps(w, sw)
    x = a.astype(np.float64)
    tail = (1,) * (a.ndim - 2)
    rows = x[y0] * (1.0 - ty).reshape((h, 1) + tail) + x[y1] * ty.reshape((h, 1) + tail)
    out = rows[:, x0] * (1.0 - tx).reshape((1, w) + tail) + rows[:, x1] * tx.reshape((1, w) + tail)
    return _round_u8(out) if a.dtype == np.uint8 else out.astype(a.dtype)


def resize_img(img, ratio):
    """utils/base_utils.py:535-540: Gaussian pre-blur + bilinear resize to round(h * ratio) x round(w * ratio)"""
    h, w = img.shape[:2]
    hn, wn = int(np.round(h * ratio)), int(np.round(w * ratio))
    return resize(downsample_gaussian_blur(img, ratio), (wn, hn), INTER_LINEAR)


def decompose_projection_matrix(P):
    """cv2.decomposeProjectionMatrix(P)[:3] -> (K [3,3] upper triangular with a positive diagonal, R [3,3] rotation, C_h [4,1]
    homogeneous camera centre) with P[:, :3] = K R and P C_h = 0.  The RQ factorisation with a positive diagonal is unique, so any
    correct implementation returns OpenCV's K and R (it builds them from three Givens rotations); the centre's homogeneous scale
    and sign are arbitrary there too (an SVD null vector) - callers divide by the last entry."""
    P = np.asarray(P, np.float64)
    M = P[:, :3]
    # RQ via QR of the row-reversed transpose
    rev = np.eye(3)[::-1]
    q, r = np.linalg.qr((rev @ M).T)
    K = rev @ r.T @ rev
    R = rev @ q.T
    s = np.diag(np.sign(np.diag(K)) + (np.diag(K) == 0))
    K, R = K @ s, s @ R
    C = -np.linalg.solve(M, P[:, 3])
    return K, R, np.concatenate([C, [1.0]])[:, None]
