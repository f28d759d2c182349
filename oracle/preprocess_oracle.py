"""CPU restatement of the reference's image preprocessor (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows crane-core/src/models/qwen3_5/processor.rs: smart_resize :64-88, process :114-210 (normalisation :126-137, patch
rows merge-block-major and (channel, temporal, y, x) inside a row :150-196).  The resize itself is delegated to PIL's
BICUBIC -- the arithmetic the reference says it mirrors ("resample: 3 = PIL BICUBIC, whose kernel (a = -0.5) is
Catmull-Rom", :104-111); the `image` crate the reference links is not in /root/reference, so equality with ITS rounding
is unpinned.  Pinned by the reference's own KATs (processor.rs:262-329), replayed in tests/test_preprocess.py."""
import math

import numpy as np


def _rust_round(x: float) -> float:                      # f64::round: half away from zero
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def smart_resize(h: int, w: int, factor: int, min_pixels: int, max_pixels: int):
    rb = lambda x: max(int(_rust_round(x / factor)), 1) * factor
    hb, wb = rb(h), rb(w)
    if hb * wb > max_pixels:
        beta = math.sqrt(h * w / max_pixels)
        hb = int(math.floor(h / beta / factor)) * factor
        wb = int(math.floor(w / beta / factor)) * factor
    elif hb * wb < min_pixels:
        beta = math.sqrt(min_pixels / (h * w))
        hb = int(math.ceil(h * beta / factor)) * factor
        wb = int(math.ceil(w * beta / factor)) * factor
    return max(hb, factor), max(wb, factor)


def process(img: np.ndarray, patch=16, tpatch=2, merge=2, min_pixels=65536, max_pixels=16777216,
            mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    from PIL import Image
    h, w = img.shape[:2]
    hn, wn = smart_resize(h, w, patch * merge, min_pixels, max_pixels)
    if (hn, wn) != (h, w):
        img = np.asarray(Image.fromarray(img, "RGB").resize((wn, hn), Image.BICUBIC))
    v = img.astype(np.float32) / np.float32(255.0)
    chw = ((v - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)).astype(np.float32).transpose(2, 0, 1)   # [3, H, W]
    hp, wp = hn // patch, wn // patch
    # [3, hb, m, p, wb, m, p] -> rows (hb, wb, m_row, m_col), row = (c, t, p_y, p_x)
    x = chw.reshape(3, hp // merge, merge, patch, wp // merge, merge, patch)
    x = x.transpose(1, 4, 2, 5, 0, 3, 6)                 # hb, wb, mr, mc, c, py, px
    x = np.repeat(x[:, :, :, :, :, None, :, :], tpatch, axis=5)
    return np.ascontiguousarray(x.reshape(hp * wp, 3 * tpatch * patch * patch)), (1, hp, wp)
