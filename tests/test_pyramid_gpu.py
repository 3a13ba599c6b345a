"""a-0 image pyramid on the device: bit-identical to the oracle (integer work)."""
import numpy as np
import pytest

from svo_pro_universal_amd import _capi as capi, frontend as fe

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,levels", [((480, 640), 5), ((480, 752), 5), ((37, 51), 3), ((64, 48), 4),
                                           ((2, 16), 2), ((481, 643), 4), ((1024, 1280), 6)])
@pytest.mark.parametrize("rounding", [capi.SVOH_HALFSAMPLE_REFERENCE, capi.SVOH_HALFSAMPLE_SCALAR])
def test_pyramid_bit_exact(gpu_ctx, oracle_lib, shape, levels, rounding):
    rng = np.random.RandomState(shape[0] * 7 + shape[1])
    img = rng.randint(0, 256, shape).astype(np.uint8)
    exp = oracle_lib.create_img_pyramid(img, levels, rounding)
    fr, got = gpu_ctx.build_pyramid(img, levels, rounding, return_levels=True)
    for l in range(levels):
        assert got[l].shape == exp[l].shape
        assert np.array_equal(got[l], exp[l]), "level %d differs" % l
        assert np.array_equal(gpu_ctx.download_level(fr, l), exp[l])
    gpu_ctx.release_frame(fr)


def test_forced_sse2_rule_on_16_multiple(gpu_ctx, oracle_lib):
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (96, 128)).astype(np.uint8)
    exp = oracle_lib.create_img_pyramid(img, 4, capi.SVOH_HALFSAMPLE_SSE2)  # 128, 64, 32 are all %16==0
    fr, got = gpu_ctx.build_pyramid(img, 4, capi.SVOH_HALFSAMPLE_SSE2, return_levels=True)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)
    # the two rules really differ on this image
    sca = oracle_lib.create_img_pyramid(img, 2, capi.SVOH_HALFSAMPLE_SCALAR)
    assert not np.array_equal(sca[1], exp[1])


def test_saturated_and_constant_images(gpu_ctx, oracle_lib):
    for val in (0, 1, 254, 255):
        img = np.full((48, 64), val, np.uint8)
        fr, got = gpu_ctx.build_pyramid(img, 3, return_levels=True)
        assert all(np.all(g == val) for g in got)


def test_batch_build_and_upload_roundtrip(gpu_ctx, oracle_lib):
    rng = np.random.RandomState(2)
    imgs = rng.randint(0, 256, (5, 120, 160)).astype(np.uint8)
    frames = gpu_ctx.build_pyramid_batch_host(imgs, 4)
    for i, fr in enumerate(frames):
        exp = oracle_lib.create_img_pyramid(imgs[i], 4)
        for l in range(4):
            assert np.array_equal(gpu_ctx.download_level(fr, l), exp[l])
    # upload an existing host pyramid (pitch != width on level 0)
    big = rng.randint(0, 256, (60, 100)).astype(np.uint8)
    view = big[:, :80]
    exp = oracle_lib.create_img_pyramid(np.ascontiguousarray(view), 3)
    fr = gpu_ctx.upload_pyramid([np.ascontiguousarray(view), exp[1], exp[2]])
    for l in range(3):
        assert np.array_equal(gpu_ctx.download_level(fr, l), exp[l])
    for fr in frames:
        gpu_ctx.release_frame(fr)


def test_bad_arguments(gpu_ctx):
    img = np.zeros((8, 8), np.uint8)
    with pytest.raises(fe.SvohError) as e:
        gpu_ctx.build_pyramid(img, 6)   # 8 >> 5 == 0
    assert e.value.code == -1
    with pytest.raises(fe.SvohError):
        gpu_ctx.download_level(123456, 0)


def test_released_slabs_are_reused_and_refilled(gpu_ctx, oracle_lib):
    """svoh_release_frame keeps a frame's allocation for the next frame of the same size (a pool of at most 8: the rest
    is freed).  Whatever the new frame gets -- a recycled slab, a fresh one -- its levels are the oracle's for ITS
    image, and other sizes in between are served as well."""
    rng = np.random.RandomState(11)
    for rnd in range(3):
        imgs = rng.randint(0, 256, (12, 120, 160)).astype(np.uint8)
        frames = [gpu_ctx.build_pyramid(imgs[k], 4) for k in range(12)]
        other = rng.randint(0, 256, (90, 130)).astype(np.uint8)
        fo = gpu_ctx.build_pyramid(other, 3)
        for k, fr in enumerate(frames):
            exp = oracle_lib.create_img_pyramid(imgs[k], 4)
            for l in range(4):
                assert np.array_equal(gpu_ctx.download_level(fr, l), exp[l]), (rnd, k, l)
        exp = oracle_lib.create_img_pyramid(other, 3)
        assert all(np.array_equal(gpu_ctx.download_level(fo, l), exp[l]) for l in range(3))
        for fr in frames + [fo]:
            gpu_ctx.release_frame(fr)
    with pytest.raises(fe.SvohError):
        gpu_ctx.download_level(frames[0], 0)   # the handle is gone, whatever became of its memory


def test_release_does_not_wait_and_does_not_disturb_queued_work(gpu_ctx):
    """A frame released while an alignment that reads it is still queued: the release returns at once (no wait for the
    device), the next frame of the same size takes the slab over -- written on the context's stream, BEHIND the queued
    kernel -- and the queued alignment delivers what the blocking call delivered."""
    import helpers
    sc = helpers.small_scene(321, n=300)
    other = helpers.small_scene(322, n=50)
    opt = capi.default_align_options(min_level=0)

    def problem():
        fr, fc = gpu_ctx.build_pyramid(sc.img_ref, 5), gpu_ctx.build_pyramid(sc.img_cur, 5)
        pb, keep = fe.make_align_problems([[(sc, fr, fc)]])
        return fr, fc, pb, keep

    fr, fc, pb, keep = problem()
    want = fe.se3_to_numpy(gpu_ctx.sparse_align(opt, pb)[0].T_icur_iref)
    gpu_ctx.release_frame(fr); gpu_ctx.release_frame(fc)
    for _ in range(4):
        fr, fc, pb, keep = problem()
        for _ in range(3):
            gpu_ctx.sparse_align_enqueue(opt, pb)           # three launches in the queue, all reading fr / fc
        gpu_ctx.release_frame(fr); gpu_ctx.release_frame(fc)
        # same size: these take the two slabs just released and overwrite them with other images
        f1, f2 = gpu_ctx.build_pyramid(other.img_ref, 5), gpu_ctx.build_pyramid(other.img_cur, 5)
        got = gpu_ctx.sparse_align_fetch(1)
        assert got[0].status == 0 and np.array_equal(fe.se3_to_numpy(got[0].T_icur_iref), want)
        gpu_ctx.release_frame(f1); gpu_ctx.release_frame(f2)
