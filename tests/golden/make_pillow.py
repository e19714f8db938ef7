"""Fixture generator (build container, needs Pillow): pins the Resize(256) + CenterCrop(224) stage to REAL Pillow output.

torchvision's ``transforms.Resize(256)`` on a PIL image is ``img.resize((ow, oh), Image.BILINEAR)`` with the short edge at 256 and
the long edge ``int(256 * long / short)``; ``CenterCrop(224)`` cuts at ``int(round((size - 224) / 2.0))`` (quant_train.py:428-440
builds exactly this pipeline).  For a handful of geometries (down- and up-scaling, portrait / landscape, an axis that keeps its
size, a tiny and a large image) this script stores what Pillow itself produces: the 224 x 224 crop and a SHA-256 of the whole
resized image.  Inputs are regenerated from the seeds by the tests.  It also writes a small JPEG (encoded by Pillow from a synthetic
picture) whose DECODED pixels and pipeline output are pinned too: the file travels, so a box with Pillow checks its own decoder
against the build container's.

    python tests/golden/make_pillow.py        ->  tests/golden/pillow_resize.npz, tests/golden/sample_500x375.jpg
"""
import hashlib
import io
import os

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
GEOMS = [(375, 500), (500, 333), (90, 120), (256, 300), (300, 256), (224, 224), (1000, 1500), (257, 259), (37, 1024)]


def synth(h, w, seed):
    """noise + hard edges + smooth ramps: exercises the clipping at 0 / 255 and long filter supports"""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    img[::7, ::5] = 255
    img[3::11, 2::13] = 0
    yy, xx = np.mgrid[0:h, 0:w]
    img[h // 3: 2 * h // 3, :, 1] = ((yy[h // 3: 2 * h // 3] * 3 + xx[h // 3: 2 * h // 3] * 2) % 256).astype(np.uint8)
    return img


def pipeline(img):
    h, w = img.shape[:2]
    if w <= h:
        ow, oh = 256, int(256 * h / w)
    else:
        oh, ow = 256, int(256 * w / h)
    full = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    top, left = int(round((oh - 224) / 2.0)), int(round((ow - 224) / 2.0))
    return full, np.ascontiguousarray(full[top:top + 224, left:left + 224])


def main():
    out = {"pillow_version": np.array(PIL.__version__), "geoms": np.array(GEOMS, np.int32), "seeds": np.arange(len(GEOMS), dtype=np.int32) + 100}
    for i, (h, w) in enumerate(GEOMS):
        full, crop = pipeline(synth(h, w, 100 + i))
        out[f"crop_{i}"] = crop
        out[f"full_sha_{i}"] = np.array(hashlib.sha256(full.tobytes()).hexdigest())
        out[f"full_shape_{i}"] = np.array(full.shape[:2], np.int32)
    # a JPEG the pipeline starts from (ImageFolder's pil_loader: Image.open(f).convert('RGB'))
    yy, xx = np.mgrid[0:375, 0:500]
    pic = np.stack([(xx * 255 // 499), (yy * 255 // 374), ((xx + yy) * 255 // 873)], -1).astype(np.uint8)
    pic[100:200, 150:350] = synth(100, 200, 7)
    buf = io.BytesIO()
    Image.fromarray(pic).save(buf, format="JPEG", quality=90)
    with open(os.path.join(HERE, "sample_500x375.jpg"), "wb") as f:
        f.write(buf.getvalue())
    dec = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    out["jpeg_decoded_sha"] = np.array(hashlib.sha256(dec.tobytes()).hexdigest())
    out["jpeg_decoded"] = dec
    out["jpeg_crop"] = pipeline(dec)[1]
    np.savez_compressed(os.path.join(HERE, "pillow_resize.npz"), **out)
    print("wrote pillow_resize.npz (Pillow", PIL.__version__ + ")")


if __name__ == "__main__":
    main()
