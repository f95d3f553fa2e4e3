"""Synthetic batches with the collator's schema (reference: util/data_utils.py:100-196 PaddedCollatorForActionPrediction,
producer vla/datasets/datasets.py:40-184) -- SURVEY 8d. The RLDS/TF input pipeline itself is out of scope."""
from __future__ import annotations

import torch

PAD_ID = 32000
IGNORE_INDEX = -100


def make_batch(B: int = 8, L_text: int = 32, seed: int = 42, device="cpu", ragged: bool = False, use_pointcloud: bool = True,
               vocab: int = 32000, pad_id: int = PAD_ID, img: int = 672, n_points: int = 1024, action_chunk: int = 1,
               with_next: bool = False):
    """input_ids = [1, prompt ids in [3, 31743], 29871, 32001, 32002, 2] (EOS last, no other id 2), right-padded when
    ragged; labels keep only the final </s>; images CLIP-normalised N(0,1) RGB + all-ones mask; points uniform in the
    RLBench workspace box; actions / proprio ~ U[-1, 1]."""
    g = torch.Generator().manual_seed(seed)
    rgb = torch.randn(B, 3, img, img, generator=g)
    images = torch.cat([rgb, torch.ones(B, 1, img, img)], dim=1)
    ids = torch.randint(3, min(31744, vocab - 4), (B, L_text), generator=g)
    ids[:, 0] = 1
    lens = [L_text - (3 * (b % 3) if ragged else 0) for b in range(B)]
    tail = [29871, 32001, 32002, 2] if vocab >= 32000 else [vocab - 3, vocab - 2, vocab - 1, 2]
    for b in range(B):
        ids[b, lens[b] - 4:lens[b]] = torch.tensor(tail)
        ids[b, lens[b]:] = pad_id
    labels = torch.full_like(ids, IGNORE_INDEX)
    for b in range(B):
        labels[b, lens[b] - 1] = 2
    lo, hi = torch.tensor([0.0, -0.4, 0.75]), torch.tensor([0.6, 0.4, 1.25])
    batch = dict(input_ids=ids, attention_mask=ids != pad_id, labels=labels, images={"front_image": images},
                 actions=torch.rand(B, action_chunk, 7, generator=g) * 2 - 1, proprio=torch.rand(B, 1, 7, generator=g) * 2 - 1,
                 action_masks=torch.ones(B, action_chunk, dtype=torch.bool), camera_name="rlbench_front")
    if use_pointcloud:
        batch["point_cloud"] = lo + (hi - lo) * torch.rand(B, n_points, 3, generator=g)
    if with_next:   # post-training targets (datasets.py next-frame fields): next RGB frame and next point cloud
        batch["next_images"] = torch.randn(B, 3, img, img, generator=g)
        batch["next_point_cloud"] = lo + (hi - lo) * torch.rand(B, n_points, 3, generator=g)
    dev = torch.device(device)
    mv = lambda v: ({k: x.to(dev) for k, x in v.items()} if isinstance(v, dict) else (v.to(dev) if torch.is_tensor(v) else v))  # noqa: E731
    return {k: mv(v) for k, v in batch.items()}
