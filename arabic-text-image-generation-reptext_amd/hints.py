"""Caller-side hint preparation of the RepText scripts without OpenCV.

What `infer.py:16-22,66-103` and `infer_inpaint.py:83-116` do per text line — render the glyph with PIL, take its bounding
box, build the position and regional masks, run Canny(50, 100) on the glyph and invert it — restated on numpy so that a host
without `cv2` can feed `FluxControlNetPipeline.__call__`. This runs once per image, outside the timed path; with `device=` the Canny map and the [0,255] -> [-1,1] preprocessing run on the
GPU (csrc/hints.hip: rt_canny_u8, rt_preprocess_u8), bit-identical to the numpy restatement below, which stays as their checker
and as the host path.

`canny_edges` follows the algorithm OpenCV documents for `cv::Canny(image, 50, 100)` with its defaults (3x3 Sobel with
replicated borders, L1 gradient magnitude, non-maximum suppression over four direction sectors split at tan 22.5° and
tan 67.5°, hysteresis with 8-connectivity). A 3-channel input is NOT converted to gray (infer.py:16-22 hands cv2.Canny the
3-channel glyph image): as cv::Canny does for cn > 1, the Sobel derivatives are taken per channel and, per pixel, the
channel with the largest |dx| + |dy| supplies (dx, dy) — so coloured glyphs (text_color_list is a user knob) give the same
contour as white ones, which a luma conversion would not (pure blue (0,0,128) has luma 15: gradient 60, below both thresholds). Parity with OpenCV is UNPINNED: cv2 is not installed here and the reference holds
no edge-map fixture; the tests pin the properties the downstream path relies on (closed one-pixel contours around glyph
strokes, threshold behaviour, shape/dtype, inversion).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image, ImageDraw

_TG22 = 0.41421356237       # tan 22.5°
_TG67 = 2.41421356237       # tan 67.5°


def rgb_to_gray_u8(rgb: np.ndarray) -> np.ndarray:
    """ITU-R 601 luma in 14-bit fixed point, the weights cvtColor uses (R 4899, G 9617, B 1868 of 16384), rounded."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def _sobel3(gray: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    p = np.pad(gray.astype(np.int32), 1, mode="edge")
    tl, tc, tr = p[:-2, :-2], p[:-2, 1:-1], p[:-2, 2:]
    ml, mr = p[1:-1, :-2], p[1:-1, 2:]
    bl, bc, br = p[2:, :-2], p[2:, 1:-1], p[2:, 2:]
    dx = (tr + 2 * mr + br) - (tl + 2 * ml + bl)
    dy = (bl + 2 * bc + br) - (tl + 2 * tc + tr)
    return dx, dy


def canny_edges(image: np.ndarray, low_threshold: float = 50, high_threshold: float = 100) -> np.ndarray:
    """uint8 [H, W] or [H, W, C] -> uint8 [H, W] edge map with values {0, 255}. Multi-channel: per pixel the channel with the
    largest L1 gradient magnitude supplies the gradient (cv::Canny with cn > 1); no gray conversion."""
    if low_threshold > high_threshold:
        low_threshold, high_threshold = high_threshold, low_threshold
    if image.ndim == 3:
        H, W, cn = image.shape
        dx = np.zeros([H, W], dtype=np.int32)
        dy = np.zeros([H, W], dtype=np.int32)
        best = np.full([H, W], -1, dtype=np.int32)
        for c in range(cn):                                          # first channel wins ties, as the strict '>' of cv::Canny's loop
            cx, cy = _sobel3(image[..., c].astype(np.uint8))
            m_c = np.abs(cx) + np.abs(cy)
            take = m_c > best
            dx, dy, best = np.where(take, cx, dx), np.where(take, cy, dy), np.where(take, m_c, best)
    else:
        H, W = image.shape
        dx, dy = _sobel3(image.astype(np.uint8))
    mag = np.abs(dx) + np.abs(dy)                                   # L1 norm (L2gradient=False)
    m = np.pad(mag, 1, mode="constant")                             # zero border: nothing outside the image is a maximum
    c = m[1:-1, 1:-1]
    ax, ay = np.abs(dx), np.abs(dy)
    horiz = ay < ax * _TG22                                          # gradient mostly along x -> compare left / right
    vert = ay > ax * _TG67                                           # gradient mostly along y -> compare up / down
    diag = ~(horiz | vert)
    same_sign = (dx ^ dy) >= 0                                       # gradient along the main diagonal
    left, right = m[1:-1, :-2], m[1:-1, 2:]
    up, down = m[:-2, 1:-1], m[2:, 1:-1]
    ul, dr = m[:-2, :-2], m[2:, 2:]
    ur, dl = m[:-2, 2:], m[2:, :-2]
    keep = np.zeros_like(c, dtype=bool)
    keep |= horiz & (c > left) & (c >= right)
    keep |= vert & (c > up) & (c >= down)
    keep |= diag & same_sign & (c > ul) & (c > dr)
    keep |= diag & ~same_sign & (c > ur) & (c > dl)
    cand = keep & (c > low_threshold)
    strong = cand & (c > high_threshold)
    # hysteresis: grow the strong set through 8-connected candidates until it stops changing
    edges = strong.copy()
    stack = list(zip(*np.nonzero(strong)))
    cand_rest = cand & ~strong
    while stack:
        y, x = stack.pop()
        y0, y1, x0, x1 = max(y - 1, 0), min(y + 2, H), max(x - 1, 0), min(x + 2, W)
        sub = cand_rest[y0:y1, x0:x1]
        if sub.any():
            ys, xs = np.nonzero(sub)
            for yy, xx in zip(ys + y0, xs + x0):
                cand_rest[yy, xx] = False
                edges[yy, xx] = True
                stack.append((yy, xx))
    return np.where(edges, 255, 0).astype(np.uint8)


def canny_hint(glyph_rgb: np.ndarray, low_threshold: float = 50, high_threshold: float = 100) -> np.ndarray:
    """`canny()` of infer.py:16-22: three identical channels of 255 - edges (black strokes outline on white)."""
    e = canny_edges(glyph_rgb, low_threshold, high_threshold)[:, :, None]
    return 255 - np.concatenate([e, e, e], axis=2)


def canny_hint_device(glyph_rgb: np.ndarray, device, low_threshold: float = 50, high_threshold: float = 100):
    """`canny()` of infer.py:16-22 followed by VaeImageProcessor.preprocess (PIPE:680), both on the device: the glyph's uint8
    pixels are uploaded once, rt_canny_u8 writes the inverted 3-channel edge hint, rt_preprocess_u8 turns it into the float32
    [1,3,H,W] tensor in [-1,1] that `prepare_image` would otherwise build on the host. Bit-identical to
    preprocess(Image.fromarray(canny_hint(glyph))) (tests/test_hints_gpu.py)."""
    import torch

    from . import ops

    g = torch.from_numpy(np.ascontiguousarray(glyph_rgb)).to(device)
    return ops.preprocess_u8(ops.canny_u8(g, low_threshold, high_threshold, invert=True, out_channels=3))


def build_text_hints(texts: Sequence[str], positions: Sequence[Tuple[int, int]], colors: Sequence[Tuple[int, int, int]], font,
                     width: int, height: int, position_margin: int = 0, mask_margin: int = 5, device=None):
    """Per text line: (canny hint RGB, position mask L, regional mask L) and the accumulated glyph image.

    `position_margin=0` is infer.py (tight bbox, infer.py:83); `position_margin=5` is infer_inpaint.py:99 (Q11). Returns
    (control_image_list, control_position_list, control_mask_list, control_glyph_all) as PIL images, the arguments
    `FluxControlNetPipeline.__call__` takes as control_image / control_position / control_mask / control_glyph.

    `device` (a HIP device): the edge hint and the position hint come back as preprocessed float32 tensors on that device —
    [1,3,H,W] / [1,1,H,W] in [-1,1], computed by rt_canny_u8 / rt_preprocess_u8 (only the glyph rendering stays PIL; the
    pipelines take such tensors as they are, PIPE:680,694) — the regional masks and the glyph stay PIL (the pipeline resizes the
    former on the device itself, the latter goes through its Lanczos-capable preprocess)."""
    images: List[Image.Image] = []
    pos_masks: List[Image.Image] = []
    reg_masks: List[Image.Image] = []
    glyph_all = np.zeros([height, width, 3], dtype=np.uint8)
    for text, pos, color in zip(texts, positions, colors):
        glyph = Image.new("RGB", (width, height), (0, 0, 0))
        draw = ImageDraw.Draw(glyph)
        draw.text(pos, text, font=font, fill=tuple(color))
        bbox = draw.textbbox(pos, text, font=font)

        def box(margin):
            m = np.zeros([height, width], dtype=np.uint8)
            m[max(bbox[1] - margin, 0) : bbox[3] + margin, max(bbox[0] - margin, 0) : bbox[2] + margin] = 255
            return Image.fromarray(m)

        reg_masks.append(box(mask_margin))
        g = np.array(glyph)
        glyph_all += g                                  # uint8 wrap-around on overlap, as the script's `+=` does
        if device is not None:
            import torch

            from . import ops

            images.append(canny_hint_device(g, device))
            pos_masks.append(ops.preprocess_u8(torch.from_numpy(np.array(box(position_margin))).to(device)))
        else:
            pos_masks.append(box(position_margin))
            images.append(Image.fromarray(canny_hint(g)))
    return images, pos_masks, reg_masks, Image.fromarray(glyph_all).convert("RGB")


def resize_img(img: Image.Image, max_side: int = 1280, min_side: int = 1024, size: Optional[Tuple[int, int]] = None,
               pad_to_max_side: bool = False, mode=Image.BILINEAR, base_pixel_number: int = 64) -> Image.Image:
    """infer_inpaint.py:25-46. Without `size`: bring the short side to `min_side`, then rescale so the LONG side is `max_side`
    (this second factor is applied whether it shrinks or enlarges), resize, snap both sides down to multiples of
    `base_pixel_number` and resize again (two resampling passes, as the script does). `pad_to_max_side` centres the result
    on a white max_side x max_side canvas."""
    w, h = img.size
    if size is not None:
        w_new, h_new = size
    else:
        r = min_side / min(h, w)
        w, h = round(r * w), round(r * h)
        r = max_side / max(h, w)
        w1, h1 = round(r * w), round(r * h)
        img = img.resize([w1, h1], mode)
        w_new, h_new = (w1 // base_pixel_number) * base_pixel_number, (h1 // base_pixel_number) * base_pixel_number
    img = img.resize([w_new, h_new], mode)
    if pad_to_max_side:
        canvas = np.full([max_side, max_side, 3], 255, dtype=np.uint8)
        ox, oy = (max_side - w_new) // 2, (max_side - h_new) // 2
        canvas[oy : oy + h_new, ox : ox + w_new] = np.array(img)
        img = Image.fromarray(canvas)
    return img
