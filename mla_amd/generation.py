"""Post-training generation heads (BASELINE config[3]).

Mirrors models/mla/generation/models.py of the reference: same class names, constructor arguments and state-dict keys
(``image_gen_module.intent_decoder.layers.0.self_attn.in_proj_weight`` ...), so a reference checkpoint's
``generation_manager`` entry loads with ``load_state_dict``. The parameter containers are torch's own nn.TransformerDecoder /
nn.MultiheadAttention / nn.Conv1d / nn.BatchNorm1d classes (identical names, shapes and default init); every forward is
re-implemented on the HIP kernels: MFMA GEMMs for the projections, batched MFMA GEMMs + a masked-softmax/dropout kernel for
the attention products, LayerNorm / BatchNorm / dropout / chamfer / image-loss kernels from csrc/gen.hip.

Scope notes
* scripts/post_rlbench.sh:26 ships USE_ROI=false: every patch is inside the ROI, the warp / alpha / offset branches are dead and the
  generated patch is ``0.05 * current + 5 * tanh(delta)``; the image head returns the raw delta logits and one loss kernel fuses
  tanh, the blend, images_to_patches addressing and the three reductions (mla_imgloss_*). ``use_roi=True`` (dilated ROI from the
  projected point centres, translation warp via affine_grid/grid_sample, alpha blend, background loss) runs through a second pair
  of kernels (mla_imgroi_*: one block per patch, bilinear sample with border clamp, gradients for delta / alpha / offsets).
* TactileGenerationModule (models.py:389-430) is built although no BASELINE config enables it (GEN_TAC=false: no tactile sensor in
  the simulator); its single-query attention runs on the SIMT GEMM fallback.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip, ops

_KEY_ALIGN = 32   # key/value rows are padded to the MFMA reduction granularity; the softmax kernel masks the padding


def _pad_keys(mem: torch.Tensor):
    S = mem.shape[1]
    Sp = (S + _KEY_ALIGN - 1) // _KEY_ALIGN * _KEY_ALIGN
    return (mem if Sp == S else F.pad(mem, (0, 0, 0, Sp - S))), S


def _self_attention(mha: nn.MultiheadAttention, x: torch.Tensor, training: bool) -> torch.Tensor:
    E = mha.embed_dim
    qkv = ops.linear(x, mha.in_proj_weight, mha.in_proj_bias)                                   # [B, S, 3E]
    o = ops.mha_core(qkv, None, mha.num_heads, x.shape[1], mha.dropout, training)
    assert E % mha.num_heads == 0       # sequence lengths that break the MFMA alignment (the 1-query tactile head) use the SIMT GEMM
    return ops.linear(o, mha.out_proj.weight, mha.out_proj.bias)


def _cross_attention(mha: nn.MultiheadAttention, x: torch.Tensor, mem_padded: torch.Tensor, nvalid: int, training: bool):
    E = mha.embed_dim
    q = ops.linear(x, ops.param_view(mha.in_proj_weight, 0, E), ops.param_view(mha.in_proj_bias, 0, E))
    kv = ops.linear(mem_padded, ops.param_view(mha.in_proj_weight, E, 3 * E), ops.param_view(mha.in_proj_bias, E, 3 * E))
    o = ops.mha_core(q, kv, mha.num_heads, nvalid, mha.dropout, training)
    return ops.linear(o, mha.out_proj.weight, mha.out_proj.bias)


class TransformerDecoderLayer(nn.TransformerDecoderLayer):
    """nn.TransformerDecoderLayer(batch_first=True, norm_first=False, activation='gelu') -- post-norm:
    x = norm1(x + drop(self_attn(x))); x = norm2(x + drop(cross_attn(x, memory))); x = norm3(x + drop(ffn(x)))."""

    def forward(self, tgt, memory, memory_valid: Optional[int] = None):   # type: ignore[override]
        tr = self.training
        if self.norm_first:
            raise NotImplementedError("norm_first decoder layers are not used by the reference")
        mem, nvalid = (memory, memory_valid) if memory_valid is not None else _pad_keys(memory)
        x = tgt
        sa = _self_attention(self.self_attn, x, tr)
        x = ops.layernorm(ops.dropout_add(sa, x, self.dropout1.p, tr), self.norm1.weight, self.norm1.bias, self.norm1.eps)
        ca = _cross_attention(self.multihead_attn, x, mem, nvalid, tr)
        x = ops.layernorm(ops.dropout_add(ca, x, self.dropout2.p, tr), self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = ops.act(ops.linear(x, self.linear1.weight, self.linear1.bias), hip.ACT_GELU_ERF)
        h = ops.dropout_add(h, None, self.dropout.p, tr)
        h = ops.linear(h, self.linear2.weight, self.linear2.bias)
        return ops.layernorm(ops.dropout_add(h, x, self.dropout3.p, tr), self.norm3.weight, self.norm3.bias, self.norm3.eps)


class TransformerDecoder(nn.TransformerDecoder):
    def forward(self, tgt, memory):   # type: ignore[override]
        mem, nvalid = _pad_keys(memory)     # pad once for all layers
        x = tgt
        for layer in self.layers:
            x = layer(x, mem, nvalid)
        if self.norm is not None:
            x = ops.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return x


def _decoder(d_model, nhead, dim_feedforward, num_layers):
    layer = TransformerDecoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward, dropout=0.1, activation="gelu",
                                    batch_first=True)
    return TransformerDecoder(layer, num_layers=num_layers)


class TransformerBlock(nn.Module):
    """models/mla/generation/models.py:39-65 (pre-norm block with DropPath; `qkv_bias` is accepted and ignored there too)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0.):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = nn.MultiheadAttention(dim, num_heads, dropout=attn_drop, batch_first=True)
        self.drop_path_prob = float(drop_path)
        self.drop_path = nn.Identity()      # parameter-free in the reference as well (timm DropPath)
        self.norm2 = nn.LayerNorm(dim)
        hidden = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(drop), nn.Linear(hidden, dim), nn.Dropout(drop))

    def forward(self, x, pos=None):
        tr = self.training
        xin = x + pos if pos is not None else x
        xn = ops.layernorm(xin, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        a = _self_attention(self.attn, xn, tr)
        x = ops.dropout_add(ops.drop_path(a, self.drop_path_prob, tr), x, 0.0, tr)
        h = ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = ops.act(ops.linear(h, self.mlp[0].weight, self.mlp[0].bias), hip.ACT_GELU_ERF)
        h = ops.dropout_add(h, None, self.mlp[2].p, tr)
        h = ops.linear(h, self.mlp[3].weight, self.mlp[3].bias)
        h = ops.dropout_add(h, None, self.mlp[4].p, tr)
        return ops.dropout_add(ops.drop_path(h, self.drop_path_prob, tr), x, 0.0, tr)


class ImageGenerationModule(nn.Module):
    """models/mla/generation/models.py:68-286."""

    def __init__(self, token_size: int = 4096, num_gen_queries: int = 64, decoder_layers: int = 3, decoder_heads: int = 8,
                 image_patch_size: int = 42, use_roi: bool = True, roi_dilation_kernel_size: int = 3, gen_delta_clip: float = 5.0,
                 max_patch_shift_pixels: int = 8, use_patch_offset: bool = True, image_num_patches: int = 256):
        super().__init__()
        self.token_size, self.num_gen_queries, self.image_patch_size = token_size, num_gen_queries, image_patch_size
        self.use_roi, self.roi_dilation_kernel_size = use_roi, roi_dilation_kernel_size
        self.gen_delta_clip, self.max_patch_shift_pixels, self.use_patch_offset = gen_delta_clip, max_patch_shift_pixels, use_patch_offset
        self.image_num_patches = image_num_patches
        self.image_gen_queries = nn.Parameter(torch.zeros(1, num_gen_queries, token_size))
        self.mae_mask_token = nn.Parameter(torch.zeros(1, 1, token_size))
        self.mae_pos_embed = nn.Parameter(torch.zeros(1, self.image_num_patches, token_size))
        self.intent_decoder = _decoder(token_size, decoder_heads, token_size * 2, 2)
        self.mae_decoder = _decoder(token_size, decoder_heads, token_size * 4, decoder_layers)
        patch_dim = image_patch_size ** 2 * 3
        self.mae_patch_norm = nn.LayerNorm(token_size)
        self.mae_delta_head = nn.Linear(token_size, patch_dim)
        self.mae_alpha_head = nn.Linear(token_size, 1)
        self.mae_offset_head = nn.Linear(token_size, 2)
        self._initialize_weights()

    def _initialize_weights(self):
        nn.init.normal_(self.image_gen_queries, std=0.02)
        nn.init.normal_(self.mae_mask_token, std=0.02)
        nn.init.normal_(self.mae_pos_embed, std=0.02)
        nn.init.normal_(self.mae_delta_head.weight, std=0.02)
        nn.init.constant_(self.mae_delta_head.bias, 0.0)
        nn.init.normal_(self.mae_alpha_head.weight, std=0.02)
        nn.init.constant_(self.mae_alpha_head.bias, -3.0)
        nn.init.normal_(self.mae_offset_head.weight, std=0.001)
        nn.init.constant_(self.mae_offset_head.bias, 0.0)

    def forward(self, llm_hidden_states, current_image_features=None, current_images_patches=None, roi_mask_2d=None):
        B = llm_hidden_states.shape[0]
        dt = llm_hidden_states.dtype
        intent = self.intent_decoder(self.image_gen_queries.to(dt).expand(B, -1, -1).contiguous(), llm_hidden_states)
        out = {}
        if self.use_roi:
            # models.py:172-186: dilate the ROI (3x3 max-pool on the 16x16 patch grid), put the mask token on ROI positions and keep
            # the current image features elsewhere (those keep their gradient path). [B, 256]-sized mask logic: torch.
            pad = (self.roi_dilation_kernel_size - 1) // 2
            roi = (F.max_pool2d(roi_mask_2d.float().unsqueeze(1), self.roi_dilation_kernel_size, 1, pad) > 0).view(B, -1)
            tokens = torch.where(roi.unsqueeze(-1), self.mae_mask_token.to(dt), current_image_features.to(dt)) + self.mae_pos_embed.to(dt)
            tokens = tokens.contiguous()
        else:
            # every patch is inside the ROI -> every decoder input token is the mask token (+ position), models.py:183-186
            roi = torch.ones((B, self.image_num_patches), dtype=torch.bool, device=llm_hidden_states.device)
            tokens = (self.mae_mask_token + self.mae_pos_embed).to(dt).expand(B, -1, -1).contiguous()
        feats = self.mae_decoder(tokens, intent)
        fn = ops.layernorm(feats, self.mae_patch_norm.weight, self.mae_patch_norm.bias, self.mae_patch_norm.eps)
        # [B, 256, ld]: 3*ps*ps = 5292 valid columns, padded to ld = 5312 so every GEMM of the head stays on the MFMA path
        out["delta_raw"] = ops.PaddedLinearFn.apply(fn, self.mae_delta_head.weight, self.mae_delta_head.bias)
        if self.use_roi:
            # alpha / offset heads (1 and 2 outputs, models.py:199-200) act on the non-ROI patches; padded to 64 columns like the delta head
            out["alpha_raw"] = ops.PaddedLinearFn.apply(fn, self.mae_alpha_head.weight, self.mae_alpha_head.bias)
            out["offset_raw"] = ops.PaddedLinearFn.apply(fn, self.mae_offset_head.weight, self.mae_offset_head.bias)
        # with the all-true mask (use_roi False) the alpha / offset heads feed nothing, get no gradient in the reference either,
        # and are not evaluated
        out["generation_roi_mask"] = roi
        out["norm_features"] = fn
        return out

    @torch.no_grad()
    def materialize(self, outputs: Dict[str, torch.Tensor], current_images_patches: torch.Tensor) -> Dict[str, torch.Tensor]:
        """The reference's full output dict (image_generation / delta_all / alpha_all / offset_all, models.py:218-224) for
        visualisation; not on the training path."""
        fn = outputs["norm_features"]
        pd = 3 * self.image_patch_size ** 2
        delta_all = torch.tanh(outputs["delta_raw"][..., :pd].float()) * self.gen_delta_clip
        alpha = torch.sigmoid(ops.linear(fn, self.mae_alpha_head.weight, self.mae_alpha_head.bias).squeeze(-1).float())
        offset = torch.tanh(ops.linear(fn, self.mae_offset_head.weight, self.mae_offset_head.bias).float()) * float(self.max_patch_shift_pixels)
        gen = 0.05 * current_images_patches.float() + delta_all
        return {"image_generation": gen, "generation_roi_mask": outputs["generation_roi_mask"], "delta_all": delta_all,
                "alpha_all": alpha, "offset_all": offset}


class PointCloudGenerationModule(nn.Module):
    """models/mla/generation/models.py:289-386."""

    def __init__(self, prismatic_hidden_dim: int = 4096, trans_dim: int = 1024, decoder_depth: int = 4, decoder_num_heads: int = 8,
                 group_size: int = 32, num_groups: int = 128, loss: str = "cdl2", use_geometric_prior: bool = True):
        super().__init__()
        self.prismatic_hidden_dim, self.trans_dim, self.decoder_depth = prismatic_hidden_dim, trans_dim, decoder_depth
        self.decoder_num_heads, self.group_size, self.num_groups, self.loss = decoder_num_heads, group_size, num_groups, loss
        self.use_geometric_prior = use_geometric_prior
        self.feature_projector = nn.Linear(prismatic_hidden_dim, trans_dim)
        self.seq_to_patch = nn.Linear(trans_dim, num_groups * trans_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, num_groups, trans_dim))
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        self.decoder_blocks = nn.ModuleList([
            TransformerBlock(dim=trans_dim, num_heads=decoder_num_heads, mlp_ratio=4.0, qkv_bias=True, drop=0.1, attn_drop=0.1,
                             drop_path=0.1) for _ in range(decoder_depth)])
        self.future_predictor = nn.Sequential(nn.Conv1d(trans_dim, trans_dim, 1), nn.BatchNorm1d(trans_dim), nn.ReLU(inplace=True),
                                              nn.Conv1d(trans_dim, 3 * group_size, 1))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, last_hidden, current_pointcloud: Optional[torch.Tensor] = None, vis: bool = False):
        B = last_hidden.shape[0]
        G, C, M = self.num_groups, self.trans_dim, self.group_size
        proj = ops.linear(last_hidden, self.feature_projector.weight, self.feature_projector.bias)      # [B, S, C]
        agg = ops.SeqMeanFn.apply(proj)                                                                 # [B, C]
        x = ops.linear(agg, self.seq_to_patch.weight, self.seq_to_patch.bias).view(B, G, C)
        pos = self.pos_embed.to(x.dtype).expand(B, -1, -1)
        for blk in self.decoder_blocks:
            x = blk(x, pos)
        conv0, bn, _, conv1 = self.future_predictor
        rows = x.reshape(B * G, C)
        h = ops.linear(rows, ops.param_view(conv0.weight, shape=(C, C)), conv0.bias)                   # Conv1d(k=1) == per-row Linear
        if bn.training:
            h, mean, var = ops.BatchNormTrainFn.apply(h, bn.weight, bn.bias, bn.eps, True)
            with torch.no_grad():
                n = B * G
                mom = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
                bn.running_var.mul_(1 - mom).add_((var * (n / max(n - 1, 1))).to(bn.running_var.dtype), alpha=mom)
                bn.num_batches_tracked += 1
        else:
            h = hip.bn_apply(h, bn.running_mean.float(), bn.running_var.float(), bn.weight, bn.bias, bn.eps, relu=True)
        deltas = ops.linear(h, ops.param_view(conv1.weight, shape=(3 * M, C)), conv1.bias)              # [B*G, 3M]
        pts = deltas.view(B * G, M, 3)
        if self.use_geometric_prior and current_pointcloud is not None:
            xyz = current_pointcloud.to(torch.float32).contiguous()
            start = torch.randint(0, xyz.shape[1], (B,), dtype=torch.long, device=xyz.device)           # FPSSampling models.py:21
            idx = hip.fps(xyz, start, G)
            centers = hip.gather_rows_f32(xyz, idx)                                                      # [B, G, 3]
            pts = pts + centers.reshape(B * G, 1, 3).to(pts.dtype)
        return {"pointcloud_coord_generation": pts.reshape(B, G * M, 3)}


class TactileGenerationModule(nn.Module):
    """models/mla/generation/models.py:389-430: one learned query cross-attends to the (projected) LLM states through a 2-layer
    post-norm decoder; a linear head predicts the next tactile reading."""

    def __init__(self, token_size: int = 4096, tactile_dim: int = 128, decoder_layers: int = 2, decoder_heads: int = 4):
        super().__init__()
        self.token_size, self.tactile_dim = token_size, tactile_dim
        self.feature_projector = nn.Linear(token_size, token_size)
        self.tactile_query = nn.Parameter(torch.zeros(1, 1, token_size))
        nn.init.normal_(self.tactile_query, std=0.02)
        self.decoder = _decoder(token_size, decoder_heads, token_size * 2, decoder_layers)
        self.output_head = nn.Linear(token_size, tactile_dim)

    def forward(self, llm_hidden_states):
        B = llm_hidden_states.shape[0]
        query = self.tactile_query.to(llm_hidden_states.dtype).expand(B, -1, -1).contiguous()
        memory = ops.linear(llm_hidden_states, self.feature_projector.weight, self.feature_projector.bias)
        decoded = self.decoder(query, memory)                                                          # [B, 1, H]
        return {"tactile_generation": ops.linear(decoded.squeeze(1), self.output_head.weight, self.output_head.bias)}


class MultimodalGenerationManager(nn.Module):
    """models/mla/generation/models.py:433-539."""

    def __init__(self, token_size: int = 4096, use_image_generation: bool = False, num_image_gen_queries: int = 64,
                 image_decoder_layers: int = 3, image_decoder_heads: int = 8, image_patch_size: int = 42, use_roi: bool = True,
                 roi_dilation_kernel_size: int = 3, use_pointcloud_generation: bool = False, pointcloud_trans_dim: int = 1024,
                 pointcloud_decoder_layers: int = 4, pointcloud_decoder_heads: int = 8, pointcloud_group_size: int = 16,
                 pointcloud_num_groups: int = 64, use_tactile_generation: bool = False, tactile_dim: int = 128,
                 tactile_decoder_layers: int = 2, tactile_decoder_heads: int = 4, image_num_patches: int = 256):
        super().__init__()
        self.use_image_generation, self.use_pointcloud_generation = use_image_generation, use_pointcloud_generation
        self.use_tactile_generation = use_tactile_generation
        if use_image_generation:
            self.image_gen_module = ImageGenerationModule(
                token_size=token_size, num_gen_queries=num_image_gen_queries, decoder_layers=image_decoder_layers,
                decoder_heads=image_decoder_heads, image_patch_size=image_patch_size, use_roi=use_roi,
                roi_dilation_kernel_size=roi_dilation_kernel_size, image_num_patches=image_num_patches)
        if use_pointcloud_generation:
            self.pointcloud_gen_module = PointCloudGenerationModule(
                prismatic_hidden_dim=token_size, trans_dim=pointcloud_trans_dim, decoder_depth=pointcloud_decoder_layers,
                decoder_num_heads=pointcloud_decoder_heads, group_size=pointcloud_group_size, num_groups=pointcloud_num_groups,
                loss="cdl2", use_geometric_prior=True)
        if use_tactile_generation:
            self.tactile_gen_module = TactileGenerationModule(token_size=token_size, tactile_dim=tactile_dim,
                                                              decoder_layers=tactile_decoder_layers, decoder_heads=tactile_decoder_heads)

    def forward(self, llm_hidden_states, current_image_features=None, current_images_patches=None, current_point_cloud=None,
                roi_mask_2d=None) -> Dict[str, torch.Tensor]:
        out: Dict[str, torch.Tensor] = {}
        if self.use_image_generation:
            out.update(self.image_gen_module(llm_hidden_states=llm_hidden_states, current_image_features=current_image_features,
                                             current_images_patches=current_images_patches, roi_mask_2d=roi_mask_2d))
        if self.use_pointcloud_generation:
            out.update(self.pointcloud_gen_module(last_hidden=llm_hidden_states, current_pointcloud=current_point_cloud))
        if self.use_tactile_generation:
            out.update(self.tactile_gen_module(llm_hidden_states=llm_hidden_states))
        return out

    def get_module_keys(self) -> list:
        keys = []
        if self.use_image_generation:
            keys.append("image_gen_module")
        if self.use_pointcloud_generation:
            keys.append("pointcloud_gen_module")
        if self.use_tactile_generation:
            keys.append("tactile_gen_module")
        return keys


def chamfer_distance_l2(pred, gt):
    """models/mla/generation/gen_loss.py:12-18."""
    return ops.ChamferFn.apply(pred, gt)
