"""SURVEY §8f row 3 on the device: rt_canny_u8 and rt_preprocess_u8 against their host restatements (hints.canny_edges,
VaeImageProcessor.preprocess) — byte / integer work, so the bar is bit-exactness. infer.py:16-22,98-100; PIPE:680,694,970."""
import numpy as np
import pytest
import torch
from PIL import Image, ImageDraw, ImageFont

pytestmark = pytest.mark.gpu

from reptext_amd import hints  # noqa: E402


def _glyph(width, height, text, pos, color, size):
    img = Image.new("RGB", (width, height), (0, 0, 0))
    d = ImageDraw.Draw(img)
    try:
        font = ImageFont.truetype("DejaVuSans.ttf", size)
    except Exception:
        font = ImageFont.load_default()
    d.text(pos, text, font=font, fill=color)
    return np.array(img), font


CASES = [
    ("white arabic 1024", 1024, 1024, "مرحبا", (370, 200), (255, 255, 255), 80),
    ("blue glyph (luma 15: a gray conversion would lose it)", 512, 384, "نص أزرق", (60, 120), (0, 0, 128), 64),
    ("two-colour overlap, odd size", 333, 257, "RepText", (11, 90), (255, 40, 0), 70),
    ("glyph touching the border", 256, 256, "مرحبا", (-20, -10), (200, 255, 10), 120),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_canny_on_device_is_bit_exact(gpu, case):
    from reptext_amd import ops

    _, w, h, text, pos, color, size = case
    g, _ = _glyph(w, h, text, pos, color, size)
    ref = hints.canny_edges(g, 50, 100)
    assert ref.max() == 255 and 0 < int((ref > 0).sum()) < ref.size // 4            # the fixture has edges
    dev = ops.canny_u8(torch.from_numpy(g).to(gpu), 50, 100)
    assert dev.shape == (h, w, 1) and torch.equal(dev[..., 0].cpu(), torch.from_numpy(ref))
    # the hint of infer.py:16-22: inverted, three channels
    hint = ops.canny_u8(torch.from_numpy(g).to(gpu), 50, 100, invert=True, out_channels=3)
    assert torch.equal(hint.cpu(), torch.from_numpy(hints.canny_hint(g)))
    # single-channel input, swapped thresholds (cv::Canny swaps them), other thresholds
    gray = hints.rgb_to_gray_u8(g)
    for lo, hi in ((100, 50), (20, 300), (0, 0)):
        assert torch.equal(ops.canny_u8(torch.from_numpy(gray).to(gpu), lo, hi)[..., 0].cpu(), torch.from_numpy(hints.canny_edges(gray, lo, hi))), (lo, hi)


def test_canny_hysteresis_long_chains_and_noise(gpu):
    """Hysteresis beyond glyphs: a weak spiral that only its innermost end makes strong (a chain hundreds of pixels long, crossing
    every region of the image), and dense random texture (many short chains, every NMS sector and tie rule exercised)."""
    from reptext_amd import ops

    H = W = 192
    img = np.zeros([H, W], dtype=np.uint8)
    y, x, dy, dx, n = H // 2, W // 2, 0, 1, 1
    val = 40                                                                 # step 40: Sobel magnitude 160 on straight runs ...
    steps = 0
    while 4 <= y < H - 4 and 4 <= x < W - 4:
        for _ in range(n):
            img[y, x] = val
            y, x = y + dy, x + dx
        dy, dx = dx, -dy
        steps += 1
        if steps % 2 == 0:
            n += 4
    img[H // 2 - 1:H // 2 + 2, W // 2 - 1:W // 2 + 2] = 255                  # ... one strong seed at the centre
    for lo, hi in ((50, 100), (100, 700), (150, 2000)):
        ref = hints.canny_edges(img, lo, hi)
        got = ops.canny_u8(torch.from_numpy(img).to(gpu), lo, hi)[..., 0].cpu()
        assert torch.equal(got, torch.from_numpy(ref)), (lo, hi)
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, size=(160, 224, 3), dtype=np.uint8)
    for lo, hi in ((50, 100), (300, 600), (700, 900)):
        ref = hints.canny_edges(noise, lo, hi)
        got = ops.canny_u8(torch.from_numpy(noise).to(gpu), lo, hi)[..., 0].cpu()
        assert torch.equal(got, torch.from_numpy(ref)), (lo, hi)
    # twice the same input: same bytes (the traversal order is free, the edge set is not)
    a = ops.canny_u8(torch.from_numpy(noise).to(gpu), 300, 600)
    assert torch.equal(a, ops.canny_u8(torch.from_numpy(noise).to(gpu), 300, 600))


def test_preprocess_and_build_text_hints_on_device(gpu):
    """rt_preprocess_u8 == VaeImageProcessor.preprocess bit for bit; build_text_hints(device=) returns the tensors prepare_image
    would have built from the PIL hints, and the pipeline's _prep_pixels takes them unchanged."""
    from reptext_amd import ops
    from reptext_amd.image_processor import VaeImageProcessor

    ip = VaeImageProcessor(vae_scale_factor=16)
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, size=(64, 80, 3), dtype=np.uint8)
    assert torch.equal(ops.preprocess_u8(torch.from_numpy(a).to(gpu)).cpu(), ip.preprocess(Image.fromarray(a)))
    assert torch.equal(ops.preprocess_u8(torch.from_numpy(a[..., 0]).to(gpu), normalize=False).cpu(), (torch.from_numpy(a[..., 0]).float() / 255.0)[None, None])
    _, font = _glyph(8, 8, "x", (0, 0), (255, 255, 255), 48)
    args = (["مرحبا", "RepText"], [(40, 30), (30, 140)], [(255, 255, 255), (0, 0, 128)], font, 320, 256)
    h_img, h_pos, h_mask, h_glyph = hints.build_text_hints(*args)
    d_img, d_pos, d_mask, d_glyph = hints.build_text_hints(*args, device=gpu)
    for hi, di in zip(h_img, d_img):
        assert di.is_cuda and di.shape == (1, 3, 256, 320) and torch.equal(di.cpu(), ip.preprocess(hi, height=256, width=320))
    for hp, dp in zip(h_pos, d_pos):
        assert dp.shape == (1, 1, 256, 320) and torch.equal(dp.cpu(), ip.preprocess(hp, height=256, width=320))
    assert all(np.array_equal(np.array(a_), np.array(b_)) for a_, b_ in zip(h_mask, d_mask)) and np.array_equal(np.array(h_glyph), np.array(d_glyph))
    # preprocess of an already normalised tensor is the identity (min < 0): the pipeline takes the device hints as they are
    assert torch.equal(ip.preprocess(d_img[0], height=256, width=320), d_img[0])
