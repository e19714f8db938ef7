"""TEST INFRASTRUCTURE ONLY - numpy restatement of Pillow's 8-bit antialiased bilinear resize (libImaging/Resample.c:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc, ImagingResampleVertical_8bpc) as
``Image.resize(size, BILINEAR)`` runs it - what torchvision's Resize(256) does to a PIL image (quant_train.py:428-440).
PINNED to real Pillow: tests/golden/pillow_resize.npz holds the output of Pillow 12.2 itself (tests/golden/make_pillow.py, nine
geometries + a decoded JPEG); tests/test_host_logic.py requires this file to reproduce every byte of it, and - where Pillow is
installed - compares live on further random geometries.  The HIP path (hawq_amd/image.py) is tested against the same fixture.
Independent of hawq_amd: its own coefficient code, whole-image passes, then the crop."""
import math

import numpy as np

BITS = 22


def _coeffs(in_size, out_size):
    scale = in_size / out_size
    fscale = scale if scale >= 1.0 else 1.0
    support = fscale   # bilinear support 1.0 x filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        ws = []
        for x in range(xmin, xmax):
            v = (x - center + 0.5) / fscale
            v = -v if v < 0 else v
            ws.append(1.0 - v if v < 1.0 else 0.0)
        tot = sum(ws)
        if tot != 0.0:
            ws = [w / tot for w in ws]
        out.append((xmin, [int(-0.5 + w * (1 << BITS)) if w < 0 else int(0.5 + w * (1 << BITS)) for w in ws]))
    return out


def _pass(img, coeffs, axis):
    img = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((len(coeffs),) + img.shape[1:], np.int64)
    for o, (x0, ks) in enumerate(coeffs):
        acc = np.full(img.shape[1:], 1 << (BITS - 1), np.int64)
        for i, k in enumerate(ks):
            acc += img[x0 + i] * k
        out[o] = np.clip(acc >> BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize(img, oh, ow):
    """uint8 HWC -> uint8 [oh, ow, C]; horizontal pass first, each pass skipped when the size does not change."""
    h, w = img.shape[:2]
    if ow != w:
        img = _pass(img, _coeffs(w, ow), 1)
    if oh != h:
        img = _pass(img, _coeffs(h, oh), 0)
    return img


def resize_center_crop(img, size=256, crop=224):
    h, w = img.shape[:2]
    if w <= h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    r = resize(img, oh, ow)
    top, left = int(round((oh - crop) / 2.0)), int(round((ow - crop) / 2.0))
    return r[top:top + crop, left:left + crop]
