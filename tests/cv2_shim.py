"""TEST INFRASTRUCTURE: the slice of the `cv2` API the reference's dataset layer calls while reading a scene, implemented by
neuray_amd/imgproc.py (OpenCV is not in this image).  tests/test_database.py installs it as `cv2` inside the REFERENCE's
`dataset.database` / `utils.base_utils`, so the reference's own database classes (DTUTestDatabase, the `black_400` resize path,
the LLFF cache builder) run unmodified and can be compared accessor by accessor with neuray_amd/database.py.  Both sides then
share the resampling arithmetic: what the comparison pins is everything around it (file naming, camera algebra, masks,
ratios, call order) - the resamplers themselves are checked by tests/test_imgproc.py and stay unpinned against real OpenCV."""
import numpy as np

from neuray_amd import imgproc

INTER_NEAREST, INTER_LINEAR, INTER_AREA, INTER_CUBIC = imgproc.INTER_NEAREST, imgproc.INTER_LINEAR, imgproc.INTER_AREA, 2
BORDER_REFLECT101 = 4


def GaussianBlur(img, ksize, sigmaX, borderType=BORDER_REFLECT101):
    assert ksize[0] == ksize[1] and borderType == BORDER_REFLECT101
    return imgproc.gaussian_blur(img, ksize[0], sigmaX)


def resize(img, dsize, *args, interpolation=None, **_kw):
    # (utils/base_utils.py:539 passes the flag as the third POSITIONAL argument - cv2's `dst` slot - so cv2 itself falls back to
    # INTER_LINEAR there; the shim does the same)
    return imgproc.resize(img, dsize, INTER_LINEAR if interpolation is None else interpolation)


def decomposeProjectionMatrix(P):
    K, R, c = imgproc.decompose_projection_matrix(np.asarray(P))
    return K, R, c, None, None, None, None
