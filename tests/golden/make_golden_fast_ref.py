#!/usr/bin/env python3
"""Generates tests/golden/fast_ref.npz from the REFERENCE's own FAST code (oracle/_ref/libfast_ref.so, built by
oracle/ref_fast/Makefile from src/fast_neon/src/{faster_corner_10_sse,fast_10,fast_10_score,nonmax_3x3}.cpp where
they lie under /root/reference): the one part of SURVEY.md section 8's rows the reference can be compiled for in
this image.  This is what pins the FAST stage of row f-2 (fd_utils::fastDetector,
src/svo_direct/src/feature_detection_utils.cpp:145-195): the oracle (tests/test_fast_ref_cpu.py) and the HIP
detector (tests/test_fast_ref_gpu.py) are compared with THESE outputs, not with each other.

Inputs (stored as pixels; a fixture is data): the pyramid levels of two rendered scenes (640x480 pinhole, 752x480
EuRoC-like), the reference's own test image src/fast_neon/test/data/test.jpg (decoded once, here, with PIL; grey =
PIL's "L"), its two first half-samplings, and small / narrow / strided / saturated images for the edge paths
(width < 22 takes the plain detector, faster_corner_10_sse.cpp:189-195).  Thresholds 5 / 10 / 20 / 40.
Outputs per (image, threshold): corner count, survivor count and the SHA-256 of the three int32 arrays (corners in
the detector's order as x, y pairs; scores; indices of the 3x3 non-maximum survivors); the arrays themselves where
they are small (<= 6000 corners), so that a mismatch can be looked at; the survivors (x, y, score) always: they are
what fd_utils::fastDetector goes on with, and what the device's dense detector can be asked for.
Usage: python tests/golden/make_golden_fast_ref.py [--check]   (--check: regenerate in memory and compare)"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from svo_pro_universal_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fast_ref.npz")
THRESHOLDS = (5, 10, 20, 40)
FULL_LISTS_UP_TO = 6000
REF_JPG = "/root/reference/src/fast_neon/test/data/test.jpg"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.int32).tobytes()).hexdigest()


def input_images():
    """name -> u8 image (possibly a strided view: the callee gets its pitch)."""
    from PIL import Image
    out = {}
    sc = synth.make_align_scene(611, n_features=8)
    for lv, im in enumerate(orc.create_img_pyramid(sc.img_ref, 5)):
        out["s640_l%d" % lv] = im
    sc = synth.make_align_scene(612, n_features=8, cam=synth.Camera.euroc_like(752, 480))
    for lv, im in enumerate(orc.create_img_pyramid(sc.img_ref, 5)):
        if lv:   # level 0 of the 752-wide case is the reference's own photograph below
            out["s752_l%d" % lv] = im
    jpg = np.asarray(Image.open(REF_JPG).convert("L"), np.uint8)
    for lv, im in enumerate(orc.create_img_pyramid(jpg, 3)):
        out["refjpg_l%d" % lv] = im
    rng = np.random.RandomState(613)
    out["noise_17x9"] = rng.randint(0, 256, (9, 17)).astype(np.uint8)
    out["noise_21x30"] = rng.randint(0, 256, (30, 21)).astype(np.uint8)      # plain detector (width < 22)
    out["noise_22x7"] = rng.randint(0, 256, (7, 22)).astype(np.uint8)       # smallest image of the SSE2 path
    out["noise_23x6"] = rng.randint(0, 256, (6, 23)).astype(np.uint8)       # height < 7: nothing
    out["noise_131x97"] = (rng.randint(0, 256, (97, 131)) // 3 * 3).astype(np.uint8)
    out["checker_64x48"] = (np.kron(np.indices((6, 8)).sum(0) % 2, np.ones((8, 8))) * 255).astype(np.uint8)
    out["blobs_96x80"] = np.clip(128 + 120 * np.sin(np.arange(96)[None, :] * 0.9) * np.sin(np.arange(80)[:, None] * 0.7), 0, 255).astype(np.uint8)
    return out


def strided_cases(images):
    """the same pixels inside a wider row: pitch != width (a cv::Mat ROI)"""
    a = images["noise_131x97"]
    wide = np.zeros((a.shape[0], 160), np.uint8)
    wide[:, 7:7 + a.shape[1]] = a
    return {"noise_131x97_pitch160": wide[:, 7:7 + a.shape[1]]}


def generate():
    assert orc.ref_fast_available(), "needs /root/reference (oracle/ref_fast/Makefile)"
    images = input_images()
    out = {}
    names = []
    for name, im in images.items():
        out["img_" + name] = np.ascontiguousarray(im)
        names.append(name)
    for name, view in strided_cases(images).items():
        out["img_" + name] = np.ascontiguousarray(view.base)      # the parent; the test cuts the same view
        out["roi_" + name] = np.array([7, 0, view.shape[1], view.shape[0]], np.int32)
        names.append(name)
    out["names"] = np.array(names)
    out["thresholds"] = np.array(THRESHOLDS, np.int32)
    total = 0
    for name in names:
        img = out["img_" + name]
        if "roi_" + name in out:
            x0, y0, w, h = out["roi_" + name]
            img = img[y0:y0 + h, x0:x0 + w]
        for thr in THRESHOLDS:
            xy, sc, nm = orc.ref_fast_corners(img, thr)
            key = "%s_t%d" % (name, thr)
            out["n_" + key] = np.array([len(sc), len(nm)], np.int32)
            out["sha_" + key] = np.array([sha(xy), sha(sc), sha(nm)])
            out["sv_" + key] = np.concatenate([xy[nm], sc[nm, None]], axis=1).astype(np.int16).reshape(-1, 3)   # survivors: x, y, score
            if len(sc) <= FULL_LISTS_UP_TO:
                out["xy_" + key] = xy.astype(np.int16)
                out["sc_" + key] = sc.astype(np.int16)
                out["nm_" + key] = nm.astype(np.int32)
            total += len(sc)
    return out, total


def main():
    out, total = generate()
    if "--check" in sys.argv:
        z = np.load(PATH)
        assert sorted(z.files) == sorted(out.keys()), "key sets differ"
        for k in z.files:
            assert np.array_equal(z[k], out[k]), k
        print("fast_ref.npz reproduced from the reference build:", len(z.files), "arrays")
        return
    np.savez_compressed(PATH, **out)
    print("wrote", PATH, os.path.getsize(PATH), "bytes;", len(out["names"]), "images x", len(THRESHOLDS), "thresholds,", total, "corners in all")


if __name__ == "__main__":
    main()
