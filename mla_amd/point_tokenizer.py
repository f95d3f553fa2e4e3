"""Point-cloud tokenizer: Point-PN 'scan' encoder (reference: models/mla/pointcloud/backbone/{pointvit,Point_PN}.py).

Same module tree / state-dict names as the reference (patch_embed.EncP.raw_point_embed.net.{0,1}, patch_embed.EncP.
LGA_list.{i}.linear2.{j}.net{1,2}.{0,1}, proj, cls_token, pos_embed, norm); the forward pass is a fixed pipeline of HIP
kernels (FPS, kNN, gather + positional embedding, 1x1-conv GEMMs, train-mode BatchNorm, max-pool). The tower is frozen on
the SFT / post-training path, yet -- exactly like the reference, which leaves it in train() mode
(training/strategies/base_strategy_mla.py:291) -- BatchNorm normalises with batch statistics and keeps updating its
running statistics. Stage "pretrain" trains it: `_forward_trainable` is the same pipeline through autograd ops.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import hip
from .llama import Linear


class Linear1Layer(nn.Module):  # Point_PN.py:173-186
    def __init__(self, in_channels, out_channels, kernel_size=1, bias=True):
        super().__init__()
        self.act = nn.ReLU(inplace=True)
        self.net = nn.Sequential(nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, bias=bias),
                                 nn.BatchNorm1d(out_channels), self.act)


class Linear2Layer(nn.Module):  # Point_PN.py:189-219
    def __init__(self, in_channels, kernel_size=1, groups=1, bias=True, adapter_layer=0):
        super().__init__()
        self.act = nn.ReLU(inplace=True)
        mid = 32 if adapter_layer == 2 else int(in_channels / 2)
        self.net1 = nn.Sequential(nn.Conv2d(in_channels, mid, kernel_size=kernel_size, groups=groups, bias=bias),
                                  nn.BatchNorm2d(mid), self.act)
        self.net2 = nn.Sequential(nn.Conv2d(mid, in_channels, kernel_size=kernel_size, bias=bias), nn.BatchNorm2d(in_channels))


class LGA(nn.Module):  # Point_PN.py:98-158
    def __init__(self, out_dim, alpha, beta, block_num, dim_expansion, type, adapter_layer=0):
        super().__init__()
        self.type, self.out_dim, self.alpha, self.beta = type, out_dim, alpha, beta
        self.adapter_layer = adapter_layer
        self.linear2 = nn.Sequential(*[Linear2Layer(out_dim, bias=True, adapter_layer=adapter_layer) for _ in range(block_num)])


class EncP(nn.Module):  # Point_PN.py:251-298
    def __init__(self, in_channels, input_points, num_stages, embed_dim, k_neighbors, alpha, beta, LGA_block, dim_expansion, type):
        super().__init__()
        self.input_points, self.num_stages, self.embed_dim = input_points, num_stages, embed_dim
        self.alpha, self.beta, self.k_neighbors = alpha, beta, k_neighbors
        self.raw_point_embed = Linear1Layer(in_channels, embed_dim, bias=False)
        self.LGA_list = nn.ModuleList()
        self.group_nums = []
        out_dim, group_num = embed_dim, input_points
        for i in range(num_stages):
            out_dim *= dim_expansion[i]
            group_num //= 2
            self.group_nums.append(group_num)
            self.LGA_list.append(LGA(out_dim, alpha, beta, LGA_block[i], dim_expansion[i], type, adapter_layer=i))


class Point_PN_scan(nn.Module):  # Point_PN.py:301-315
    def __init__(self, in_channels=3, class_num=15, input_points=1024, num_stages=2, embed_dim=96, k_neighbors=81, beta=100,
                 alpha=1000, LGA_block=(2, 1, 1, 1), dim_expansion=(2, 2, 2, 1), type="scan"):
        super().__init__()
        self.EncP = EncP(in_channels, input_points, num_stages, embed_dim, k_neighbors, alpha, beta, LGA_block, dim_expansion, type)
        self.out_channels = embed_dim
        for i in dim_expansion:
            self.out_channels *= i


def _bn_train(x2d: torch.Tensor, bn: nn.modules.batchnorm._BatchNorm, residual=None, relu=False):
    """BatchNorm with batch statistics over all rows (= over B x spatial), running stats updated like torch does."""
    if not bn.training:      # inference (vlm.eval(), model_mla.py:619): normalise with the running statistics
        return hip.bn_apply(x2d, bn.running_mean.float(), bn.running_var.float(), bn.weight, bn.bias, bn.eps, residual=residual, relu=relu)
    mean, var = hip.colstats(x2d)
    y = hip.bn_apply(x2d, mean, var, bn.weight, bn.bias, bn.eps, residual=residual, relu=relu)
    if bn.track_running_stats and bn.running_mean is not None:
        n = x2d.shape[0]
        m = bn.momentum if bn.momentum is not None else 0.1
        bn.running_mean.mul_(1 - m).add_(mean.to(bn.running_mean.dtype), alpha=m)
        bn.running_var.mul_(1 - m).add_((var * (n / max(n - 1, 1))).to(bn.running_var.dtype), alpha=m)
        bn.num_batches_tracked += 1
    return y


class PointTokenizer(nn.Module):
    """pointvit.py:17-82. forward(p [B, 1024, 3]) -> (tokens [B, 256, 768], centres [B, 256, 3])."""

    def __init__(self, in_channels=3, embed_dim=768, depth=12, num_heads=6, mlp_ratio=4.0, target_token_count=256,
                 norm_args=None, **kwargs):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = Point_PN_scan()
        self.proj = Linear(384, 768)
        self.cls_token = nn.Parameter(torch.randn(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, target_token_count + 1, embed_dim))
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)   # create_norm({'norm': 'ln', 'eps': 1e-6}) -- never called in forward
        self.fps_starts_override = None  # tests: list of int64 [B] start indices per stage
        self.last_indices = None         # (fps_idx, knn_idx) per stage of the last forward (parity tests)
        self.initialize_weights()

    def initialize_weights(self):
        torch.nn.init.normal_(self.cls_token, std=0.02)
        torch.nn.init.normal_(self.pos_embed, std=0.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d, nn.BatchNorm1d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, p, x=None, **kwargs):
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            return self._forward_trainable(p)     # stage "pretrain" (base_strategy_mla.py / prismatic.py freeze_backbones)
        enc = self.patch_embed.EncP
        xyz = p.float().contiguous()                      # pointvit.py:66-74 forces fp32 coordinates
        B, N, _ = xyz.shape
        idx_dbg = []
        with torch.no_grad():
            conv, bn = enc.raw_point_embed.net[0], enc.raw_point_embed.net[1]
            x0 = hip.gemm(xyz.reshape(B * N, 3).to(torch.bfloat16), conv.weight.reshape(conv.out_channels, 3).contiguous())
            feats = _bn_train(x0, bn, relu=True).view(B, N, -1)
            for i in range(enc.num_stages):
                G, K = enc.group_nums[i], enc.k_neighbors
                fps_idx, knn_idx = self._group(xyz, i, G, K)
                idx_dbg.append((fps_idx, knn_idx))
                lga = enc.LGA_list[i]
                rows, lc_xyz = hip.lga_prep(xyz, feats.contiguous(), fps_idx, knn_idx, lga.alpha, lga.beta)
                for blk in lga.linear2:
                    c1, b1, c2, b2 = blk.net1[0], blk.net1[1], blk.net2[0], blk.net2[1]
                    y = hip.gemm(rows, c1.weight.reshape(c1.out_channels, c1.in_channels).contiguous(), bias=c1.bias)
                    y = _bn_train(y, b1, relu=True)
                    y = hip.gemm(y, c2.weight.reshape(c2.out_channels, c2.in_channels).contiguous(), bias=c2.bias)
                    rows = _bn_train(y, b2, residual=rows, relu=True)
                feats = hip.maxpool_k(rows, B * G, K).view(B, G, -1)
                xyz = lc_xyz
            tokens = hip.gemm(feats.reshape(B * xyz.shape[1], -1), self.proj.weight, bias=self.proj.bias).view(B, xyz.shape[1], -1)
        self.last_indices = idx_dbg
        return tokens, xyz

    def _group(self, xyz, stage, G, K):
        """Farthest-point centres + their K nearest neighbours (index work, no gradient): Point_PN.py:8-27, :49-74."""
        with torch.no_grad():
            if self.fps_starts_override is not None:
                start = self.fps_starts_override[stage].to(xyz.device)
            else:
                start = torch.randint(0, xyz.shape[1], (xyz.shape[0],), dtype=torch.long, device=xyz.device)  # Point_PN.py:10
            fps_idx = hip.fps(xyz, start.contiguous(), G)
            centers = hip.gather_rows_f32(xyz, fps_idx)
            knn_idx = hip.knn(xyz, centers, K)
        return fps_idx, knn_idx

    def _forward_trainable(self, p):
        """The same pipeline with autograd: 1x1 convolutions as GEMMs with weight-gradient delivery, train-mode BatchNorm
        backward, gather / max-pool backward. Coordinates, FPS / kNN indices and the sin/cos embedding carry no gradient."""
        from . import ops
        enc = self.patch_embed.EncP
        xyz = p.float().contiguous()
        B, N, _ = xyz.shape
        idx_dbg = []

        def bn_rows(x2, bn, relu):
            y, mean, var = ops.BatchNormTrainFn.apply(x2, bn.weight, bn.bias, bn.eps, relu)
            if bn.track_running_stats and bn.running_mean is not None:
                with torch.no_grad():
                    n, m = x2.shape[0], (bn.momentum if bn.momentum is not None else 0.1)
                    bn.running_mean.mul_(1 - m).add_(mean.to(bn.running_mean.dtype), alpha=m)
                    bn.running_var.mul_(1 - m).add_((var * (n / max(n - 1, 1))).to(bn.running_var.dtype), alpha=m)
                    bn.num_batches_tracked += 1
            return y

        conv, bn = enc.raw_point_embed.net[0], enc.raw_point_embed.net[1]
        x0 = ops.linear(xyz.reshape(B * N, 3).to(torch.bfloat16), ops.param_view(conv.weight, shape=(conv.out_channels, 3)))
        feats = bn_rows(x0, bn, True).view(B, N, -1)
        for i in range(enc.num_stages):
            G, K = enc.group_nums[i], enc.k_neighbors
            fps_idx, knn_idx = self._group(xyz, i, G, K)
            idx_dbg.append((fps_idx, knn_idx))
            lga = enc.LGA_list[i]
            rows, lc_xyz = ops.LgaPrepFn.apply(xyz, feats, fps_idx, knn_idx, lga.alpha, lga.beta)
            for blk in lga.linear2:
                c1, b1, c2, b2 = blk.net1[0], blk.net1[1], blk.net2[0], blk.net2[1]
                y = ops.linear(rows, ops.param_view(c1.weight, shape=(c1.out_channels, c1.in_channels)), c1.bias)
                y = bn_rows(y, b1, True)
                y = ops.linear(y, ops.param_view(c2.weight, shape=(c2.out_channels, c2.in_channels)), c2.bias)
                y = bn_rows(y, b2, False)
                rows = ops.act(ops.dropout_add(y, rows, 0.0, False), hip.ACT_RELU)       # act(net2(net1(x)) + x), Point_PN.py:218
            feats = ops.MaxPoolKFn.apply(rows, B * G, K).view(B, G, -1)
            xyz = lc_xyz
        tokens = ops.linear(feats.reshape(B * xyz.shape[1], -1), self.proj.weight, self.proj.bias).view(B, xyz.shape[1], -1)
        self.last_indices = idx_dbg
        return tokens, xyz
