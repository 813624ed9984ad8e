"""Caller-side hint preparation (reptext_amd.hints): Canny without OpenCV and the per-line masks of infer.py / infer_inpaint.py.

cv2 is not installed and the reference ships no edge-map fixture, so parity with cv2.Canny is UNPINNED; these tests pin the
properties the hot path's inputs depend on."""
import numpy as np
import pytest
from PIL import Image, ImageFont

from reptext_amd import hints


def test_canny_blank_and_inversion():
    z = np.zeros([64, 96, 3], dtype=np.uint8)
    assert hints.canny_edges(z).max() == 0
    h = hints.canny_hint(z)
    assert h.shape == (64, 96, 3) and h.dtype == np.uint8 and h.min() == 255       # no edges -> all white (infer.py:21)


def test_canny_rectangle_gives_closed_thin_contour():
    img = np.zeros([80, 120], dtype=np.uint8)
    img[20:60, 30:90] = 255
    e = hints.canny_edges(img, 50, 100)
    assert set(np.unique(e)) == {0, 255}
    ys, xs = np.nonzero(e)
    # edges hug the rectangle border (within one pixel) and nothing fires inside or far outside
    assert ys.min() >= 18 and ys.max() <= 61 and xs.min() >= 28 and xs.max() <= 91
    assert e[30:50, 40:80].max() == 0
    # every row crossing the rectangle has exactly one edge pixel per side: the contour is one pixel thick
    for y in range(25, 55):
        row = np.nonzero(e[y])[0]
        assert len(row) == 2 and row[0] in (29, 30) and row[1] in (89, 90)
    # closed: every edge pixel has at least two 8-neighbours on the contour
    p = np.pad(e > 0, 1)
    nb = sum(np.roll(np.roll(p, dy, 0), dx, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0))
    assert (nb[p] >= 2).all()


def test_canny_thresholds_and_hysteresis():
    # a faint step (contrast 8: Sobel magnitude 32) is below `low`; a medium one (contrast 20: 80) only survives next to a strong one
    img = np.zeros([40, 90], dtype=np.uint8)
    img[:, 30:] = 8
    assert hints.canny_edges(img, 50, 100).max() == 0
    img = np.zeros([40, 90], dtype=np.uint8)
    img[:, 30:] = 20
    assert hints.canny_edges(img, 50, 100).max() == 0            # weak only: no seed
    img[:20, 30:] = 60                                            # upper half of the same step is strong (240 > 100)
    e = hints.canny_edges(img, 50, 100)
    assert e[:18].any() and e[25:].any()                          # the weak part is kept through connectivity
    col = np.nonzero(e[25:].any(axis=0))[0]                       # away from the 60|20 seam (itself a strong horizontal edge)
    assert col.min() >= 28 and col.max() <= 31
    assert hints.canny_edges(img, 100, 50).tolist() == e.tolist()    # swapped thresholds are reordered, as cv::Canny does


def test_gray_conversion_weights():
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]]], dtype=np.uint8)
    assert hints.rgb_to_gray_u8(rgb).tolist() == [[76, 150, 29, 255]]


def test_build_text_hints_shapes_and_margins():
    font = ImageFont.truetype("DejaVuSans.ttf", 40)
    W, H = 512, 256
    imgs, pos, masks, glyph = hints.build_text_hints(["مرحبا", "RepText"], [(150, 40), (150, 140)], [(255, 255, 255), (0, 255, 0)], font, W, H)
    assert len(imgs) == len(pos) == len(masks) == 2
    for im, p, m in zip(imgs, pos, masks):
        assert im.size == (W, H) and im.mode == "RGB" and p.mode == "L" and m.mode == "L"
        a, pm, mm = np.array(im), np.array(p), np.array(m)
        assert (a == a[..., :1]).all()                               # three identical channels
        assert (a < 255).sum() > 100                                 # strokes produced edges
        ys, xs = np.nonzero(pm)
        y2, x2 = np.nonzero(mm)
        assert y2.min() == ys.min() - 5 and y2.max() == ys.max() + 5 and x2.min() == xs.min() - 5 and x2.max() == xs.max() + 5
        ey, ex = np.nonzero(a[..., 0] < 255)
        assert ey.min() >= y2.min() and ey.max() <= y2.max() and ex.min() >= x2.min() and ex.max() <= x2.max()   # edges inside the region mask
    assert glyph.size == (W, H) and np.array(glyph).max() == 255
    # inpaint script: position mask = bbox +- 5 (Q11)
    _, pos5, masks5, _ = hints.build_text_hints(["RepText"], [(150, 140)], [(0, 255, 0)], font, W, H, position_margin=5)
    assert np.array_equal(np.array(pos5[0]), np.array(masks5[0]))


def test_resize_img_matches_script_arithmetic():
    im = Image.new("RGB", (1500, 1000), (10, 20, 30))
    out = hints.resize_img(im)
    # short side 1000 -> 1024 (w 1536), long side 1536 -> 1280 (h 853), snapped to multiples of 64
    assert out.size == (1280, 832)
    assert hints.resize_img(im, size=(640, 320)).size == (640, 320)
    assert hints.resize_img(im, pad_to_max_side=True).size == (1280, 1280)


def test_canny_multichannel_takes_the_strongest_channel_not_luma():
    """infer.py:16-22 gives cv2.Canny the 3-channel glyph; cv::Canny with cn > 1 keeps, per pixel, the channel
    with the largest |dx|+|dy|. Pure blue text (0,0,128) has luma 15 — a gray conversion would lose the whole contour."""
    font = ImageFont.truetype("DejaVuSans.ttf", 48)
    from PIL import ImageDraw

    def glyph(color):
        im = Image.new("RGB", (256, 128), (0, 0, 0))
        ImageDraw.Draw(im).text((20, 30), "Text", font=font, fill=color)
        return np.array(im)

    e_white, e_blue, e_red = (hints.canny_edges(glyph(c)) for c in ((255, 255, 255), (0, 0, 128), (200, 0, 0)))
    assert e_blue.any() and e_red.any()
    # same glyph geometry -> (nearly) the same contour whatever the colour: the anti-aliased ramps scale with the colour value
    agree = lambda a, b: float(((a > 0) & (b > 0)).sum()) / float((a > 0).sum())
    assert agree(e_white, e_red) > 0.85 and agree(e_white, e_blue) > 0.8
    # a rectangle in one channel only: identical to the single-channel result of that channel
    img = np.zeros([80, 120, 3], dtype=np.uint8)
    img[20:60, 30:90, 2] = 128
    assert np.array_equal(hints.canny_edges(img), hints.canny_edges(img[..., 2]))
    # gray images (all channels equal) are unchanged by the multi-channel rule
    g = np.zeros([80, 120], dtype=np.uint8)
    g[20:60, 30:90] = 255
    assert np.array_equal(hints.canny_edges(np.stack([g] * 3, axis=2)), hints.canny_edges(g))
