"""Golden vectors for the image pre-processing adapter from the REAL stack: the reference's vendored
CLIPImageProcessor (transformers/models/clip/image_processing_clip.py) on top of Pillow -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_preprocess.py

Input: seeded 224x224 RGB noise + a smooth gradient image (exercises clipping and the flat regions). Stores three 9-row strips of
the 672x672 result (uint8 after the resize, float32 after normalisation). Writes tests/golden/preprocess.npz.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ROWS = np.r_[0:9, 300:309, 663:672]


def inputs():
    rng = np.random.RandomState(7)
    noise = rng.randint(0, 256, (224, 224, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:224, 0:224]
    grad = np.stack([(xx * 255 // 223), (yy * 255 // 223), ((xx + yy) % 256)], -1).astype(np.uint8)
    hard = np.where(((xx // 8 + yy // 8) % 2)[..., None] > 0, 255, 0).astype(np.uint8).repeat(3, -1)     # checkerboard: overshoot clipping
    return {"noise": noise, "grad": grad, "hard": hard}


def main():
    ref_import.setup()
    from PIL import Image
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(do_resize=True, size=672, do_center_crop=True, crop_size=672, do_normalize=True, do_rescale=True)
    res = {}
    for name, img in inputs().items():
        pil = Image.fromarray(img)
        resized = np.array(pil.resize((672, 672), resample=Image.BICUBIC))
        pv = proc.preprocess(pil, return_tensors="np")["pixel_values"][0]
        res[f"{name}_u8_rows"] = resized[ROWS]
        res[f"{name}_f32_rows"] = pv[:, ROWS, :]
        res[f"{name}_u8_sum"] = np.array(resized.astype(np.int64).sum())
    res["rows"] = ROWS
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **res)
    print("preprocess.npz:", {k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
