"""GPU parity of the image pre-processing adapter: mla_clip_preprocess must be BIT-EXACT against vectors captured
from PIL.Image.resize(BICUBIC) + the reference's vendored CLIPImageProcessor (tests/golden/preprocess.npz) -- integer resize and
IEEE float32 normalisation -- and against the numpy oracle on the whole frame."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from test_oracle_golden import preprocess_inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_clip_preprocess_bit_exact(dev):
    from mla_amd.vision_tokenizer import ClipImagePreprocessor
    gold = np.load(os.path.join(G, "preprocess.npz"))
    rows = gold["rows"]
    proc = ClipImagePreprocessor(672, device=dev)
    imgs = preprocess_inputs()
    batch = torch.from_numpy(np.stack(list(imgs.values())))
    out = proc.preprocess(batch, mask_channel=True)["pixel_values"]
    assert out.shape == (3, 4, 672, 672) and out.dtype == torch.float32
    assert bool((out[:, 3] == 1).all())                                     # the all-ones mask channel (datasets.py:68-69)
    res = out[:, :3].cpu().numpy()
    for i, (name, img) in enumerate(imgs.items()):
        assert np.array_equal(res[i][:, rows, :], gold[f"{name}_f32_rows"]), name
        assert np.array_equal(res[i], O.clip_preprocess(img)), name        # whole frame vs the oracle
    single = proc.preprocess(imgs["noise"])["pixel_values"]                # HWC numpy frame, HF calling convention
    assert single.shape == (1, 3, 672, 672) and np.array_equal(single[0].cpu().numpy(), res[0])
    bf = proc.preprocess(batch, out_dtype=torch.bfloat16)["pixel_values"]
    assert torch.equal(bf, out[:, :3].to(torch.bfloat16))
