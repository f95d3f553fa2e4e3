"""Capture the CROPPED pixel-mask path of the reference's VisionTokenizer (models/mla/image/vision_tokenizer.py:129-150) at B = 1.

Test infrastructure, build container only (imports the real reference from /root/reference; see ref_import.py). Writes
tests/golden/vision_crop.npz: for each mask case the mask rectangle, the reference's token count [h, w] and a 64-channel slice of
the projected tokens. B = 1 is the only batch size where the reference's crop path is self-consistent (the per-sample token counts
differ and PrismaticVLM hard-codes 256 image tokens downstream, models/vlm/prismatic.py:932-933); SURVEY 8c lists "all-ones and
cropped mask" for the a5 golden.

Cases (patch grid 48 x 48, conv_stride 3):
  rect_div   : ones on patch rows 6..41, cols 3..38   -> 36 x 36 patches -> 12 x 12 = 144 tokens
  rect_rem   : ones on patch rows 0..39, cols 5..47   -> 40 x 43 patches -> 13 x 14 = 182 tokens (avg-pool / unfold drop the remainder)
  all_zero   : mask == 0 everywhere                   -> the reference's 16 x 16 fallback (:131-132) -> 5 x 5 = 25 tokens
  full       : all ones (the shipped loaders' case)   -> 16 x 16 = 256 tokens (cross-check against components.npz)

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_crop.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import recipe, ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CASES = {"rect_div": (6, 41, 3, 38), "rect_rem": (0, 39, 5, 47), "all_zero": None, "full": (0, 47, 0, 47)}


def make_pixels(case):
    """[1, 4, 672, 672]: the recipe's first image with the mask channel replaced by the case's patch rectangle."""
    batch, _ = recipe.make_batch(R=1)
    px = batch["images"]["front_image"][:1].clone()
    px[:, 3] = 0.0
    rect = CASES[case]
    if rect is not None:
        r0, r1, c0, c1 = rect
        px[:, 3, r0 * 14:(r1 + 1) * 14, c0 * 14:(c1 + 1) * 14] = 1.0
    return px


def main():
    ref_import.setup()
    from models.mla.image.vision_tokenizer import MLP_GELU, VisionTokenizer
    vt = VisionTokenizer(1024)
    vt.load_state_dict({k: recipe.det_weight("vlm.vision_tower_2d." + k, v.shape) for k, v in vt.state_dict().items()})
    proj = MLP_GELU(1024, recipe.TOKEN_SIZE, 2)
    proj.load_state_dict({k: recipe.det_weight("vlm.projector_2d." + k, v.shape) for k, v in proj.state_dict().items()})
    res = {}
    for case in CASES:
        with torch.no_grad():
            toks, hw = vt(make_pixels(case), proj)
        assert len(toks) == 1
        res[f"{case}_hw"] = hw[0].numpy()
        res[f"{case}_tokens_slice"] = toks[0][:, :64].numpy()
        res[f"{case}_rect"] = np.array(CASES[case] if CASES[case] is not None else (-1, -1, -1, -1))
        print(case, "hw", hw[0].tolist(), "tokens", tuple(toks[0].shape))
    np.savez_compressed(os.path.join(OUT, "vision_crop.npz"), **res)


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
