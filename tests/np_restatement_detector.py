"""Independent NumPy restatement of the reference's keyframe feature detector.

A SECOND reading of the reference, used only by tests: written from the reference's source files (`file:line` below,
paths relative to the reference tree) with different machinery from oracle/svo_oracle_detector.c and csrc/detector.hip --
whole-image arrays: the FAST segment test as sliding minima over the 16 ring differences, the non-maximum suppression
as a dense 3x3 comparison on a score image (the reference walks sorted corner lists with row pointers), the blur and
the Scharr derivative as shifted-array sums -- so that a line misread by the author of the C oracle and of the kernels
does not pass unnoticed because both sides share it.  tests/test_np_second_opinion_cpu.py compares the two.

Restated:
  FastDetector::detect / FastGradDetector::detect      src/svo_direct/src/feature_detection.cpp:113-194
  fd_utils::fillFeatures / fastDetector / edgeletDetector_V2 / getAngleAtPixelUsingHistogram / angle_hist::*
                                                        src/svo_direct/src/feature_detection_utils.cpp:72-195, 313-385,
                                                        831-839, 947-1009
  fast::fast_corner_detect_10(_sse2), fast_corner_score_10, fast_nonmax_3x3
                                                        src/fast_neon/src/faster_corner_10_sse.cpp:180-202,
                                                        fast_10_score.cpp:21-3180, nonmax_3x3.cpp:17-112
  OccupandyGrid2D::getCellIndex                          src/svo_common/include/svo/common/occupancy_grid_2d.h:82-94

Third-party arithmetic (OpenCV 4.x, not in the reference tree) from its published behaviour: cv::GaussianBlur(3x3,
sigma 0) on 8-bit images = the table kernel [1 2 1] / 4 in both directions, fixed point, one rounding (half up) of the
sum over 16; cv::Scharr(CV_16S) = [-1 0 1] along the derivative, [3 10 3] across, exact; BORDER_DEFAULT = reflect 101.

What the segment test means (fast_10_score.cpp is generated code: a decision tree per barrier): a pixel is a corner at
barrier b when 10 contiguous pixels of the 16-pixel ring are all > c + b or all < c - b; fast_corner_score_10 starts at
b + 1, raises b by the smallest margin of the tests it used while the pixel still passes, and returns b - 1 when it
fails: the largest barrier at which the pixel is a corner.
"""
import math

import numpy as np

# fast_10_score.cpp:3158-3175: pixel[i] = dx + stride * dy
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]

CORNER, EDGELET = 7, 6     # svo::FeatureType (src/svo_common/include/svo/common/types.h:60-73)
N_BINS = 36                # feature_detection_utils.h:168


def fast_score_image(img):
    """Largest barrier at which each pixel is a FAST-10 corner (-1: not even at barrier 0; -2: within 3 pixels of the
    border, where the detector does not look: faster_corner_10_sse.cpp loops y in [3, h-3), x in [3, w-3))."""
    im = img.astype(np.int32)
    h, w = im.shape
    out = np.full((h, w), -2, np.int32)
    if h < 7 or w < 7:
        return out
    c = im[3:h - 3, 3:w - 3]
    d = np.stack([im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] - c for dx, dy in RING])   # ring minus centre, 16 x H' x W'
    best = np.full(c.shape, -1000, np.int32)
    for sign in (1, -1):
        e = sign * d
        ee = np.concatenate([e, e[:9]])                         # circular
        for s in range(16):
            best = np.maximum(best, ee[s:s + 10].min(axis=0))   # all ten > b  <=>  b < their minimum
    out[3:h - 3, 3:w - 3] = np.where(best >= 1, best - 1, -1)   # corner at barrier 0 needs a margin >= 1
    return out


def fast_corners(img, barrier):
    """fast_corner_detect_10 + fast_corner_score_10 + fast_nonmax_3x3: (x, y, score) of the survivors in raster order.
    Non-maximum suppression: a corner goes when one of its 8 neighbours is a corner (at this barrier) with a score
    >= its own (nonmax_3x3.cpp:49-103: left, right, the three above, the three below)."""
    s = fast_score_image(img)
    if img.shape[1] >= 22 and img.shape[0] < 7:        # faster_corner_10_sse.cpp:193-195
        return []
    is_c = s >= barrier
    sc = np.where(is_c, s, -10).astype(np.int32)
    pad = np.pad(sc, 1, constant_values=-10)
    h, w = sc.shape
    keep = is_c.copy()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            keep &= ~(pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] >= sc)
    ys, xs = np.nonzero(keep)
    return [(int(x), int(y), int(s[y, x])) for y, x in zip(ys, xs)]


def cell_index(x, y, scale, cell_size, n_cols):   # occupancy_grid_2d.h:82-94, on doubles
    return int(math.floor(float(scale * y) / cell_size) * n_cols + math.floor(float(scale * x) / cell_size))


class Corner(object):      # feature_detection_types.h: x, y, score, level, angle
    __slots__ = ("x", "y", "score", "level", "angle")

    def __init__(self, x, y, score, level, angle):
        self.x, self.y, self.score, self.level, self.angle = x, y, np.float32(score), level, np.float32(angle)


def fast_detector(levels, threshold, border, min_level, max_level, corners, occupancy, cell_size, n_cols):
    """feature_detection_utils.cpp:145-195: the best corner per free cell, levels in ascending order."""
    for level in range(min_level, max_level + 1):
        img = levels[level]
        scale = 1 << level
        maxw, maxh = img.shape[1] - border, img.shape[0] - border
        for x, y, s in fast_corners(img, threshold):
            if x < border or y < border or x >= maxw or y >= maxh:
                continue
            k = cell_index(x, y, scale, cell_size, n_cols)
            if occupancy[k]:
                continue
            if np.float32(s) > corners[k].score:
                corners[k] = Corner(x * scale, y * scale, s, level, 0.0)


def _reflect101(a, n):
    return np.pad(a, n, mode="reflect")


def gaussian_blur_3x3(img):
    p = _reflect101(img.astype(np.int32), 1)
    hsum = p[:, :-2] + 2 * p[:, 1:-1] + p[:, 2:]
    v = hsum[:-2] + 2 * hsum[1:-1] + hsum[2:]
    return ((v + 8) >> 4).astype(np.uint8)


def scharr(img, x_derivative):
    p = _reflect101(img.astype(np.int32), 1)
    if x_derivative:
        d = p[:, 2:] - p[:, :-2]                      # [-1 0 1] along x
        return (3 * d[:-2] + 10 * d[1:-1] + 3 * d[2:]).astype(np.int16)
    d = p[2:] - p[:-2]
    return (3 * d[:, :-2] + 10 * d[:, 1:-1] + 3 * d[:, 2:]).astype(np.int16)


def angle_at_pixel_using_histogram(img, x, y, halfpatch):
    """feature_detection_utils.cpp:831-839, 947-1009."""
    hist = [0.0] * N_BINS
    rows, cols = img.shape
    for dy in range(-halfpatch, halfpatch + 1):
        for dx in range(-halfpatch, halfpatch + 1):
            xx, yy = x + dx, y + dy
            if yy > 0 and yy < rows - 1 and xx > 0 and xx < cols - 1:
                gx = float(int(img[yy, xx + 1]) - int(img[yy, xx - 1]))
                gy = float(int(img[yy + 1, xx]) - int(img[yy - 1, xx]))
                mag = math.sqrt(gx * gx + gy * gy)
                ang = math.atan2(gy, gx)
                b = int(round_half_away(N_BINS * (ang + math.pi) / (2.0 * math.pi)))
                b = b if b < N_BINS else 0
                hist[b] += mag
    prev, h0 = hist[N_BINS - 1], hist[0]
    for i in range(N_BINS):
        tmp = hist[i]
        hist[i] = 0.25 * prev + 0.5 * hist[i] + 0.25 * (h0 if i + 1 == N_BINS else hist[i + 1])
        prev = tmp
    max_bin, max_val = 0, hist[0]
    for i in range(1, N_BINS):
        if hist[i] > max_val:
            max_val, max_bin = hist[i], i
    return max_bin * 2.0 * math.pi / N_BINS


def round_half_away(v):        # std::round
    return math.floor(v + 0.5) if v >= 0 else math.ceil(v - 0.5)


def edgelet_detector_v2(levels, threshold, border, corners, occupancy, cell_size, n_cols):
    """feature_detection_utils.cpp:313-385: gradient magnitude of the blurred level 1, 8-neighbour suppression with
    the reference's asymmetric comparisons, the best edgelet per free cell, its direction from the angle histogram."""
    level, scale = 1, 2
    src = levels[level]
    img = gaussian_blur_3x3(src)
    dx, dy = scharr(img, True).astype(np.int32), scharr(img, False).astype(np.int32)
    rows, cols = src.shape
    score = np.zeros((rows, cols), np.float32)
    if rows - border > border and cols - border > border:
        sl = (slice(border, rows - border), slice(border, cols - border))
        mag = np.sqrt((dx[sl] * dx[sl] + dy[sl] * dy[sl]).astype(np.float64)).astype(np.float32)
        score[sl] = np.where(mag > np.float32(threshold), mag, np.float32(0.0))
    thr = np.float32(threshold)
    for y in range(border, rows - border):
        for x in range(border, cols - border):
            c = score[y, x]
            if c < thr:
                continue
            k = cell_index(x, y, scale, cell_size, n_cols)
            if occupancy[k]:
                continue
            if score[y, x + 1] >= c or score[y, x - 1] > c:
                continue
            if score[y + 1, x] >= c or score[y - 1, x] > c:
                continue
            if score[y + 1, x + 1] >= c or score[y + 1, x - 1] > c:
                continue
            if score[y - 1, x + 1] >= c or score[y - 1, x - 1] > c:
                continue
            if c > corners[k].score:
                corners[k] = Corner(x * scale, y * scale, c, level - 1, angle_at_pixel_using_histogram(src, x, y, 4))


def fill_features(corners, ftype, mask, threshold, max_n_features, out, occupancy, cell_size, n_cols):
    """feature_detection_utils.cpp:72-143.  std::sort is not stable: equal scores are ordered by cell here; callers that
    compare with another implementation must not cut inside a tie (or compare sets)."""
    cand = []
    for k, c in enumerate(corners):
        if float(c.score) > threshold:
            if mask is not None and mask[int(c.y), int(c.x)] == 0:
                continue
            cand.append((k, c))
            occupancy[cell_index(c.x, c.y, 1, cell_size, n_cols)] = 1
    cand.sort(key=lambda kc: (-float(kc[1].score), kc[0]))
    for k, c in cand[:max(0, max_n_features)]:
        out["px"].append((float(c.x), float(c.y)))
        out["score"].append(float(c.score))
        out["level"].append(int(c.level))
        out["grad"].append((float(np.cos(np.float32(c.angle))), float(np.sin(np.float32(c.angle)))))   # std::cos(float)
        out["type"].append(ftype)


def detect(levels, cell_size=30, max_level=2, min_level=0, border=8, detect_edgelets=False, threshold_primary=10.0,
           threshold_secondary=100.0, occupancy=None, mask=None, max_n_features=None):
    """FastDetector::detect (feature_detection.cpp:113-131) / FastGradDetector::detect (:155-194)."""
    h, w = levels[0].shape
    n_cols, n_rows = int(math.ceil(w / cell_size)), int(math.ceil(h / cell_size))
    n_cells = n_cols * n_rows
    occ = np.zeros(n_cells, np.uint8) if occupancy is None else np.array(occupancy, np.uint8).copy()
    if max_n_features is None:
        max_n_features = n_cells
    out = dict(px=[], score=[], level=[], grad=[], type=[])
    corners = [Corner(0, 0, threshold_primary, 0, 0.0) for _ in range(n_cells)]
    fast_detector(levels, int(threshold_primary), border, min_level, max_level, corners, occ, cell_size, n_cols)
    fill_features(corners, CORNER, mask, threshold_primary, max_n_features, out, occ, cell_size, n_cols)
    if detect_edgelets:
        max_features = max_n_features - len(out["px"])
        if max_features > 0:
            corners = [Corner(0, 0, threshold_secondary, 0, 0.0) for _ in range(n_cells)]
            edgelet_detector_v2(levels, int(threshold_secondary), border, corners, occ, cell_size, n_cols)
            fill_features(corners, EDGELET, mask, threshold_secondary, max_features, out, occ, cell_size, n_cols)
    return dict(px=np.array(out["px"], np.float64).reshape(-1, 2), score=np.array(out["score"], np.float64),
                level=np.array(out["level"], np.int32), grad=np.array(out["grad"], np.float64).reshape(-1, 2),
                type=np.array(out["type"], np.uint8))
