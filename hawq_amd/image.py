"""Image pipeline in front of the uint8 input path (quant_train.py:428-440): ``transforms.Resize(256)`` and
``transforms.CenterCrop(224)`` on the decoded image, on the MI355X; ``ToTensor`` + ``Normalize`` + the input QuantAct are
the look-up table of ``IntegerEngine.forward_uint8``.

torchvision's Resize on a PIL image is ``Image.resize(..., BILINEAR)``: Pillow's antialiased separable resampling
(libImaging/Resample.c).  The coefficient construction below restates its ``precompute_coeffs`` (binary64, support scaled by
the down-sampling factor, taps normalised to sum 1) and ``normalize_coeffs_8bpc`` (22 fractional bits, round half away);
the two passes run in ``hawq_resample_u8``.  Pinned to REAL Pillow output: ``tests/golden/pillow_resize.npz``
(``make_pillow.py``, Pillow 12.2) holds what ``Image.resize(..., BILINEAR)`` + the crop produce for nine geometries and a JPEG;
``oracle/pil_resample.py`` and this device stage both reproduce it bit for bit.

JPEG decoding stays on the host, as in the reference (``datasets.ImageFolder``'s ``pil_loader`` inside DataLoader workers,
quant_train.py:428-445): ``decode_image`` / ``folder_loader`` below use Pillow when it is installed and say so when it is not;
everything after the decoded uint8 HWC pixels runs on the MI355X.
"""
from __future__ import annotations

import functools
import math

import numpy as np
import torch

from . import _lib

PRECISION_BITS = 32 - 8 - 2


def bilinear_coeffs(in_size: int, out_size: int):
    """Resample.c: precompute_coeffs (bilinear filter, support 1.0) + normalize_coeffs_8bpc.
    Returns (bounds int32 [out, 2] = (first input index, taps), coef int32 [out, ksize], ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax, dtype=np.float64)
        w = np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(w < 1.0, 1.0 - w, 0.0)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    coef = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64))
    return bounds, coef.astype(np.int32), ksize


def resize_crop_geometry(h: int, w: int, resize: int = 256, crop: int = 224):
    """torchvision Resize(int) (smaller edge -> resize, other edge int(resize * long / short)) + CenterCrop(crop):
    (resized h, resized w, crop top, crop left)."""
    if w <= h:
        ow, oh = resize, int(resize * h / w)
    else:
        oh, ow = resize, int(resize * w / h)
    return oh, ow, int(round((oh - crop) / 2.0)), int(round((ow - crop) / 2.0))


@functools.lru_cache(maxsize=256)   # ImageNet validation has thousands of distinct (h, w): bound the device allocations
def _coeffs_dev(in_size, out_size, lo, n, dev):
    """(bounds relative to the first input index read, coefficients, taps, first input index, one past the last) of output
    indices lo .. lo+n-1, on `dev`"""
    b, c, k = bilinear_coeffs(in_size, out_size)
    b = np.ascontiguousarray(b[lo:lo + n]).copy()
    first, last = int(b[:, 0].min()), int((b[:, 0] + b[:, 1]).max())
    b[:, 0] -= first
    return (torch.from_numpy(b).to(dev), torch.from_numpy(np.ascontiguousarray(c[lo:lo + n])).to(dev), k, first, last)


def resize_center_crop(img: torch.Tensor, resize: int = 256, crop: int = 224) -> torch.Tensor:
    """uint8 HWC image on the MI355X -> uint8 [crop, crop, C], equal to CenterCrop(crop)(Resize(resize)(PIL image)) as
    restated above.  Only the crop window is computed (same values: the passes are separable and local)."""
    if not img.is_cuda or img.dtype != torch.uint8 or img.dim() != 3:
        raise ValueError("expected a uint8 HWC tensor on the MI355X")
    h, w, ch = img.shape
    oh, ow, top, left = resize_crop_geometry(h, w, resize, crop)
    if oh < crop or ow < crop:
        raise ValueError("image too small for the crop (torchvision pads here; ImageNet validation images never need it)")
    img = img.contiguous()
    dev, sp = img.device, torch.cuda.current_stream(img.device).cuda_stream
    bv, cv, kv, y0, y1 = _coeffs_dev(h, oh, top, crop, str(dev))       # vertical taps of the crop rows: input rows y0 .. y1
    bh, chh, kh, x0, _ = _coeffs_dev(w, ow, left, crop, str(dev))       # horizontal taps: bounds relative to input column x0
    # Pillow runs the horizontal pass first, on the input rows the vertical pass will read
    tmp = torch.empty(y1 - y0, crop, ch, dtype=torch.uint8, device=dev)
    if ow != w:
        _lib.call("hawq_resample_u8", img.data_ptr() + x0 * ch, w, ch, bh.data_ptr(), chh.data_ptr(), kh, crop, 1, y1 - y0, y0, tmp.data_ptr(), sp)
    else:
        tmp.copy_(img[y0:y1, left:left + crop])
    if oh == h:
        return tmp[top - y0:top - y0 + crop].clone()
    out = torch.empty(crop, crop, ch, dtype=torch.uint8, device=dev)
    _lib.call("hawq_resample_u8", tmp.data_ptr(), crop, ch, bv.data_ptr(), cv.data_ptr(), kv, crop, 0, crop, 0, out.data_ptr(), sp)   # tmp starts at input row y0
    return out


def preprocess_batch(images, resize: int = 256, crop: int = 224) -> torch.Tensor:
    """List of decoded uint8 HWC images (any sizes, host or device) -> uint8 [N, crop, crop, C] for ``forward_uint8``."""
    return torch.stack([resize_center_crop(im.cuda() if not im.is_cuda else im, resize, crop) for im in images])


# ------------------------------------------------------------------------------------------------------------------
# Host side of the data path (quant_train.py:428-445): the files of an ImageFolder tree, decoded as its pil_loader does
IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def decode_image(source) -> torch.Tensor:
    """File path, bytes or file object -> uint8 HWC RGB tensor (host), decoded as torchvision's ``pil_loader`` does
    (``Image.open(f).convert('RGB')``).  Needs Pillow - the decoder is not part of the integer hot path."""
    try:
        from PIL import Image
    except ImportError as e:   # pragma: no cover - the build container has Pillow
        raise RuntimeError("hawq_amd.image.decode_image needs Pillow; pass decoded uint8 HWC tensors to preprocess_batch instead") from e
    import io
    if isinstance(source, (bytes, bytearray, memoryview)):
        source = io.BytesIO(bytes(source))
    with Image.open(source) as im:
        arr = np.array(im.convert("RGB"))   # a copy: Pillow's buffer is read-only
    return torch.from_numpy(arr)


def image_folder(root: str):
    """(path, class index) pairs of an ImageFolder tree: class = sub-directory, indices by sorted directory name, files in
    sorted walk order (torchvision.datasets.folder.make_dataset)."""
    import os
    classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
    if not classes:
        raise FileNotFoundError(f"no class directories under {root}")
    samples = []
    for idx, c in enumerate(classes):
        for dirpath, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
            samples += [(os.path.join(dirpath, f), idx) for f in sorted(files) if f.lower().endswith(IMG_EXTENSIONS)]
    return samples, classes


def folder_loader(root: str, batch_size: int = 128, resize: int = 256, crop: int = 224, device="cuda"):
    """Iterate an ImageFolder tree as ``(uint8 [n, crop, crop, 3] on the MI355X, int64 targets)`` batches - the validation loader of
    quant_train.py:428-445 (shuffle off) with everything behind the decoder on the device; feed it to ``api.validate(uint8=True)``."""
    samples, _ = image_folder(root)
    for i in range(0, len(samples), batch_size):
        part = samples[i:i + batch_size]
        imgs = [decode_image(p).to(device) for p, _ in part]
        yield preprocess_batch(imgs, resize, crop), torch.tensor([t for _, t in part], dtype=torch.int64)
