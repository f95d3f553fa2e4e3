"""CPU oracle for the post-training generation heads: functional plain-PyTorch restatement.

TEST INFRASTRUCTURE ONLY (see oracle/torch_oracle.py). The reference builds these heads from torch's own
nn.TransformerDecoder / nn.MultiheadAttention (torch 2.x, a third-party dependency of the reference), so the restatement
spells out what those modules compute -- post-norm decoder layers, packed in-projection, scaled dot-product attention --
directly from a state dict. Dropout / DropPath are stochastic in train mode; the oracle models the deterministic network
(p = 0) with BatchNorm on batch statistics, which is how oracle/capture_golden.py runs the real reference
("parity tests must zero the dropout").

Pinned by tests/golden/generation.npz (captured from the imported reference, oracle/capture_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F


def _mha(x_q, x_kv, sd, pfx, nheads):
    """torch.nn.functional.multi_head_attention_forward with batch_first inputs, no masks, dropout 0."""
    E = x_q.shape[-1]
    W, b = sd[pfx + "in_proj_weight"], sd[pfx + "in_proj_bias"]
    q = F.linear(x_q, W[:E], b[:E])
    k = F.linear(x_kv, W[E:2 * E], b[E:2 * E])
    v = F.linear(x_kv, W[2 * E:], b[2 * E:])
    B, Sq, _ = q.shape
    Sk = k.shape[1]
    hd = E // nheads
    q = q.view(B, Sq, nheads, hd).transpose(1, 2)
    k = k.view(B, Sk, nheads, hd).transpose(1, 2)
    v = v.view(B, Sk, nheads, hd).transpose(1, 2)
    p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Sq, E)
    return F.linear(o, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])


def _ln(x, sd, pfx, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + "weight"], sd[pfx + "bias"], eps)


def decoder_layer_postnorm(tgt, memory, sd, pfx, nheads):
    """nn.TransformerDecoderLayer.forward with norm_first=False, activation gelu (erf), as instantiated at
    models/mla/generation/models.py:103-122."""
    x = _ln(tgt + _mha(tgt, tgt, sd, pfx + "self_attn.", nheads), sd, pfx + "norm1.")
    x = _ln(x + _mha(x, memory, sd, pfx + "multihead_attn.", nheads), sd, pfx + "norm2.")
    ff = F.linear(F.gelu(F.linear(x, sd[pfx + "linear1.weight"], sd[pfx + "linear1.bias"])), sd[pfx + "linear2.weight"],
                  sd[pfx + "linear2.bias"])
    return _ln(x + ff, sd, pfx + "norm3.")


def transformer_decoder(tgt, memory, sd, pfx, nlayers, nheads):
    x = tgt
    for i in range(nlayers):
        x = decoder_layer_postnorm(x, memory, sd, f"{pfx}layers.{i}.", nheads)
    return x


def images_to_patches(images, ps=42):
    """models/mla/generation/utils.py:7-18: [B, 3, H, W] -> [B, (H/ps)(W/ps), 3*ps*ps] (patch-major, then channel, row, col)."""
    B, C, H, W = images.shape
    g = H // ps
    x = images.reshape(B, C, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, g * g, C * ps * ps)


def image_generation(hidden, sd, pfx, nheads, n_intent_layers=2, n_mae_layers=3, clip=5.0):
    """ImageGenerationModule.forward models.py:158-224 for use_roi=False (all-true mask): returns delta_all = clip*tanh(head)."""
    B = hidden.shape[0]
    queries = sd[pfx + "image_gen_queries"].expand(B, -1, -1)
    intent = transformer_decoder(queries, hidden, sd, pfx + "intent_decoder.", n_intent_layers, nheads)
    tokens = (sd[pfx + "mae_mask_token"] + sd[pfx + "mae_pos_embed"]).expand(B, -1, -1)     # masked everywhere, :183-186
    feats = transformer_decoder(tokens, intent, sd, pfx + "mae_decoder.", n_mae_layers, nheads)
    fn = _ln(feats, sd, pfx + "mae_patch_norm.")
    delta = torch.tanh(F.linear(fn, sd[pfx + "mae_delta_head.weight"], sd[pfx + "mae_delta_head.bias"])) * clip
    return delta


def image_generation_loss(delta_all, curr_images, next_images, ps=42):
    """_generate_generated_patches models.py:264-283 with mask == 1 (roi_pred = 0.05*(curr+delta) + 0.95*delta, alpha = 1)
    + compute_generation_losses prismatic.py:780-816 (empty background set)."""
    curr = images_to_patches(curr_images[:, :3], ps)
    nxt = images_to_patches(next_images, ps)
    gen = 0.05 * (curr + delta_all) + 0.95 * delta_all
    mse = F.mse_loss(gen, nxt)
    l1 = F.l1_loss(gen, nxt)
    reward = -0.1 * delta_all.abs().mean()
    return mse + 0.5 * l1 + reward, dict(mse=mse, l1=l1, delta_abs=delta_all.abs().mean())


def transformer_block_prenorm(x, pos, sd, pfx, nheads):
    """TransformerBlock.forward models.py:56-65 (DropPath / Dropout off)."""
    xn = _ln(x + pos, sd, pfx + "norm1.")
    x = x + _mha(xn, xn, sd, pfx + "attn.", nheads)
    h = _ln(x, sd, pfx + "norm2.")
    h = F.linear(F.gelu(F.linear(h, sd[pfx + "mlp.0.weight"], sd[pfx + "mlp.0.bias"])), sd[pfx + "mlp.3.weight"], sd[pfx + "mlp.3.bias"])
    return x + h


def pointcloud_generation(hidden, sd, pfx, nheads, depth, num_groups, group_size, bn_eps=1e-5):
    """PointCloudGenerationModule.forward models.py:351-386 with current_pointcloud=None (prismatic.py:1098)."""
    B = hidden.shape[0]
    proj = F.linear(hidden, sd[pfx + "feature_projector.weight"], sd[pfx + "feature_projector.bias"])
    agg = proj.mean(dim=1)
    C = agg.shape[-1]
    x = F.linear(agg, sd[pfx + "seq_to_patch.weight"], sd[pfx + "seq_to_patch.bias"]).reshape(B, num_groups, C)
    pos = sd[pfx + "pos_embed"].expand(B, -1, -1)
    for i in range(depth):
        x = transformer_block_prenorm(x, pos, sd, f"{pfx}decoder_blocks.{i}.", nheads)
    rows = x.reshape(B * num_groups, C)
    h = F.linear(rows, sd[pfx + "future_predictor.0.weight"].squeeze(-1), sd[pfx + "future_predictor.0.bias"])
    mean, var = h.mean(0), h.var(0, unbiased=False)                                           # BatchNorm1d, training statistics
    h = (h - mean) / torch.sqrt(var + bn_eps) * sd[pfx + "future_predictor.1.weight"] + sd[pfx + "future_predictor.1.bias"]
    h = F.relu(h)
    d = F.linear(h, sd[pfx + "future_predictor.3.weight"].squeeze(-1), sd[pfx + "future_predictor.3.bias"])
    return d.reshape(B, num_groups * group_size, 3)


def chamfer_distance_l2(pred, gt):
    """models/mla/generation/gen_loss.py:12-18."""
    d = torch.cdist(pred, gt)
    return (d.min(dim=2)[0].mean(dim=1) + d.min(dim=1)[0].mean(dim=1)).mean()


def generation_losses(hidden, curr_images, next_images, next_pc, sd: Dict[str, torch.Tensor], cfg: dict, pfx="generation_manager."):
    """MultimodalGenerationManager.forward + compute_generation_losses for gen_image / gen_pointcloud. Returns
    (image_gen_loss, point_cloud_gen_loss, extras)."""
    delta = image_generation(hidden, sd, pfx + "image_gen_module.", cfg["image_heads"], 2, cfg["image_layers"])
    img_loss, parts = image_generation_loss(delta, curr_images, next_images)
    pts = pointcloud_generation(hidden, sd, pfx + "pointcloud_gen_module.", cfg["pc_heads"], cfg["pc_layers"], cfg["pc_groups"],
                                cfg["pc_group_size"])
    pc_loss = chamfer_distance_l2(pts.float(), next_pc.float())
    return img_loss, pc_loss, dict(delta_all=delta, points=pts, **parts)


def image_generation_roi(hidden, image_features, roi_mask_2d, sd, pfx, nheads, n_intent_layers=2, n_mae_layers=3, clip=5.0, shift=8.0,
                         dilation=3):
    """ImageGenerationModule.forward models.py:158-224 with use_roi=True: dilated ROI (utils.py:35-44), mask tokens on ROI positions,
    delta / alpha / offset heads. Returns (delta_all, alpha_all, offset_all, roi_flat)."""
    B = hidden.shape[0]
    intent = transformer_decoder(sd[pfx + "image_gen_queries"].expand(B, -1, -1), hidden, sd, pfx + "intent_decoder.", n_intent_layers, nheads)
    pad = (dilation - 1) // 2
    roi = (F.max_pool2d(roi_mask_2d.float().unsqueeze(1), dilation, 1, pad) > 0).view(B, -1)
    tokens = torch.where(roi.unsqueeze(-1), sd[pfx + "mae_mask_token"].view(1, 1, -1), image_features) + sd[pfx + "mae_pos_embed"]
    feats = transformer_decoder(tokens, intent, sd, pfx + "mae_decoder.", n_mae_layers, nheads)
    fn = _ln(feats, sd, pfx + "mae_patch_norm.")
    delta = torch.tanh(F.linear(fn, sd[pfx + "mae_delta_head.weight"], sd[pfx + "mae_delta_head.bias"])) * clip
    alpha = torch.sigmoid(F.linear(fn, sd[pfx + "mae_alpha_head.weight"], sd[pfx + "mae_alpha_head.bias"]).squeeze(-1))
    offset = torch.tanh(F.linear(fn, sd[pfx + "mae_offset_head.weight"], sd[pfx + "mae_offset_head.bias"])) * shift
    return delta, alpha, offset, roi


def image_generation_roi_loss(delta, alpha, offset, roi, curr_images, next_images, ps=42):
    """_generate_generated_patches models.py:226-286 (translation warp through affine_grid / grid_sample, align_corners=True,
    border padding) + the image terms of compute_generation_losses prismatic.py:780-816."""
    curr = images_to_patches(curr_images[:, :3], ps)
    nxt = images_to_patches(next_images, ps)
    B, P, _ = curr.shape
    cimg = curr.reshape(B * P, 3, ps, ps)
    off = offset.reshape(B * P, 2)
    theta = torch.zeros(B * P, 2, 3)
    theta[:, 0, 0] = 1.0
    theta[:, 1, 1] = 1.0
    theta[:, 0, 2] = 2.0 * off[:, 0] / float(ps - 1)
    theta[:, 1, 2] = 2.0 * off[:, 1] / float(ps - 1)
    grid = F.affine_grid(theta, size=(B * P, 3, ps, ps), align_corners=True)
    warped = F.grid_sample(cimg.float(), grid, mode="bilinear", padding_mode="border", align_corners=True)
    dimg = delta.reshape(B * P, 3, ps, ps)
    roi_pred = 0.05 * (cimg + dimg) + 0.95 * dimg
    non_roi = warped + dimg
    m = roi.reshape(B * P, 1, 1, 1)
    pred = torch.where(m, roi_pred, non_roi)
    a = torch.where(roi, torch.ones_like(alpha), alpha).reshape(B * P, 1, 1, 1)
    gen = (a * pred + (1.0 - a) * cimg).reshape(B, P, -1)
    total = 0.0
    parts = {}
    if roi.any():
        parts["roi"] = F.mse_loss(gen[roi], nxt[roi]) + 0.5 * F.l1_loss(gen[roi], nxt[roi])
        total = total + parts["roi"]
    if (~roi).any():
        parts["bg"] = 0.01 * F.l1_loss(gen[~roi], nxt[~roi])
        total = total + parts["bg"]
    parts["delta"] = -0.1 * delta.abs().mean()
    return total + parts["delta"], parts
