"""Camera projection + multimodal-alignment contrastive heads (reference: models/mla/fuser/{camera,contrastive}.py)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn

from . import hip, ops


@dataclass
class CameraParams:
    K: torch.Tensor
    R: torch.Tensor
    t: torch.Tensor
    image_size_orig: tuple


# models/mla/fuser/camera.py:12-52; image_size_orig from the per-camera projection functions (contrastive.py:8,51,94)
CAMERA_CONFIGS = {
    "rlbench_front": CameraParams(
        K=torch.tensor([[-307.7174807, 0.0, 112.0], [0.0, -307.7174807, 112.0], [0.0, 0.0, 1.0]], dtype=torch.float32),
        R=torch.tensor([[1.19209290e-07, -4.22617942e-01, -9.06307936e-01], [-1.00000000e+00, -5.96046448e-07, 1.49011612e-07],
                        [-5.66244125e-07, 9.06307936e-01, -4.22617912e-01]], dtype=torch.float32),
        t=torch.tensor([1.34999919e+00, 3.71546562e-08, 1.57999933e+00], dtype=torch.float32), image_size_orig=(224, 224)),
    "franka_right": CameraParams(
        K=torch.tensor([[387.414794921875, 0.0, 319.47052001953125], [0.0, 386.8714904785156, 241.13287353515625],
                        [0.0, 0.0, 1.0]], dtype=torch.float32),
        R=torch.tensor([[0.91300858, 0.26157042, -0.31304353], [0.39730357, -0.7442472, 0.53688545],
                        [-0.09254842, -0.61455433, -0.78342694]], dtype=torch.float32),
        t=torch.tensor([0.8591219242556176, -0.5851783639922448, 0.7535876808722389], dtype=torch.float32),
        image_size_orig=(480, 640)),
    "franka_front": CameraParams(
        K=torch.tensor([[388.2638244628906, 0.0, 328.3757019042969], [0.0, 387.84130859375, 240.24295043945312],
                        [0.0, 0.0, 1.0]], dtype=torch.float32),
        R=torch.tensor([[-0.01750229, 0.95018522, -0.31119403], [0.99984609, 0.01625676, -0.00659609],
                        [-0.0012085, -0.31126158, -0.95032351]], dtype=torch.float32),
        t=torch.tensor([0.8545415959817313, 0.5748472977587156, 1.0411478820663598], dtype=torch.float32),
        image_size_orig=(720, 1280)),
}


def get_camera_params(config_name="default", device=None) -> CameraParams:
    """camera.py:54-66 -- without the reference's in-place mutation of the global config."""
    if config_name not in CAMERA_CONFIGS:
        raise ValueError(f"Unknown camera config: {config_name}. Available configs: {list(CAMERA_CONFIGS.keys())}")
    p = CAMERA_CONFIGS[config_name]
    if device is None:
        return p
    return CameraParams(p.K.to(device), p.R.to(device), p.t.to(device), p.image_size_orig)


def projection_constants(camera_name: str, image_size_resize=(672, 672)):
    """Host-side (fp32, torch CPU) folding of K scaling and the world->camera transform exactly as
    project_3d_to_2d_672_* does it (contrastive.py:13-27): returns Rw [3,3], tw [3], Ks [3,3] as flat fp32 lists."""
    p = CAMERA_CONFIGS[camera_name]
    oh, ow = p.image_size_orig
    sx, sy = image_size_resize[1] / ow, image_size_resize[0] / oh
    Ks = p.K.clone()
    Ks[0, 0] *= sx
    Ks[1, 1] *= sy
    Ks[0, 2] *= sx
    Ks[1, 2] *= sy
    Rw = p.R.T.contiguous()
    tw = -Rw @ p.t
    return Rw, tw, Ks


def get_projection_func(camera_name: str):
    if camera_name not in CAMERA_CONFIGS:
        raise ValueError(f"Unknown projection func for camera {camera_name}. Available: {list(CAMERA_CONFIGS.keys())}")

    def project(xyz_3d, K=None, R=None, t=None, image_size_resize=(672, 672), vision_strides=None):
        vs = vision_strides or {"patch_stride": 14, "conv_stride": 2}
        return project_points(xyz_3d, camera_name, image_size_resize, vs["patch_stride"] * vs["conv_stride"])

    return project


_CONSTS_ON_DEVICE: dict = {}


def project_points(xyz: torch.Tensor, camera_name: str, image_size_resize=(672, 672), total_stride: int = 42):
    """contrastive.py:5-45: pinhole projection of point centres to the patch grid. xyz [..., 3] fp32 on the GPU.
    Returns (patch_idx [..., 2] int64 (row, col), valid bool [...])."""
    key = (camera_name, tuple(image_size_resize), xyz.device)
    consts = _CONSTS_ON_DEVICE.get(key)
    if consts is None:     # uploaded once: a pageable host-to-device copy per step is a host synchronisation in the front end
        Rw, tw, Ks = projection_constants(camera_name, image_size_resize)
        consts = _CONSTS_ON_DEVICE[key] = torch.cat([Rw.reshape(-1), tw.reshape(-1), Ks.reshape(-1)]).to(xyz.device)
    flat = xyz.reshape(-1, 3).float().contiguous()
    n = flat.shape[0]
    idx = torch.empty((n, 2), dtype=torch.int64, device=xyz.device)
    valid = torch.empty((n,), dtype=torch.bool, device=xyz.device)
    hip.project_points(flat, consts, idx, valid, float(image_size_resize[1]), float(image_size_resize[0]), float(total_stride),
                       image_size_resize[0] // total_stride, image_size_resize[1] // total_stride)
    return idx.view(*xyz.shape[:-1], 2), valid.view(xyz.shape[:-1])


def _head(feature_dim, projection_dim):
    from .llama import Linear
    return nn.Sequential(Linear(feature_dim, feature_dim), nn.ReLU(inplace=True), Linear(feature_dim, projection_dim))


def _run_head(head, x):
    h = head[0](x)
    h = ops.act(h, hip.ACT_RELU)
    return head[2](h)


class CoordinateAwareContrastiveLoss(nn.Module):
    """contrastive.py:170-215. Negatives are local to the rank (no cross-rank gather)."""

    def __init__(self, feature_dim, projection_dim=256, temperature=0.07):
        super().__init__()
        self.temperature = temperature
        self.image_projection_head = _head(feature_dim, projection_dim)
        self.pointcloud_projection_head = _head(feature_dim, projection_dim)

    def forward(self, image_features, pointcloud_features, patch_indices, valid_mask):
        B, n_patches, _ = image_features.shape
        img_proj = ops.l2_normalize(_run_head(self.image_projection_head, image_features.contiguous()))
        pc_proj = ops.l2_normalize(_run_head(self.pointcloud_projection_head, pointcloud_features.contiguous()))
        patch_w = int(n_patches ** 0.5)
        linear = patch_indices[:, :, 0] * patch_w + patch_indices[:, :, 1]          # [B, N_points]
        D = img_proj.shape[-1]
        # row gathers / compaction are index plumbing (torch); all arithmetic below runs in the HIP kernels
        base = (torch.arange(B, device=linear.device) * n_patches)[:, None]
        vidx = getattr(valid_mask, "_mla_valid_index", None)      # taken early by PrismaticVLM.forward (no sync behind the decoder)
        if vidx is None:
            vidx = torch.nonzero(valid_mask.reshape(-1), as_tuple=False).squeeze(-1)   # host sync, like the reference's mask index
        M = int(vidx.numel())
        if M == 0:
            return torch.tensor(0.0, device=image_features.device, requires_grad=True)
        Mp = ((M + 127) // 128) * 128
        tgt_rows = (base + linear).reshape(-1)[vidx]
        # HIP row gathers into the zero-padded [Mp, D] operands; tgt_rows repeats (several centres per patch): the backward sums the
        # duplicates in a fixed order (ops.GatherRowsSumFn), so the whole step is bit-reproducible
        a = ops.gather_rows_sum(pc_proj.reshape(-1, D), vidx, Mp)
        b = ops.gather_rows_sum(img_proj.reshape(-1, D), tgt_rows, Mp)
        return ops.info_nce(a, b, M, self.temperature)


class TactileContrastiveLoss(nn.Module):
    """contrastive.py:219-258: tactile tokens against ALL 256 point-cloud tokens and ALL 256 image tokens of their sample; the
    positives are the point centre nearest to the gripper and the image patch it projects to. Parameters are always
    constructed (the reference builds the module even when tactile is off, modeling_llama.py:1148-1156)."""

    def __init__(self, feature_dim, projection_dim=256, temperature=0.07):
        super().__init__()
        self.temperature = temperature
        self.tactile_projection_head = _head(feature_dim, projection_dim)
        self.pointcloud_projection_head = _head(feature_dim, projection_dim)
        self.image_projection_head = _head(feature_dim, projection_dim)

    def forward(self, tac_features, pc_features, img_features, positive_pc_indices, linear_positive_img_indices):
        if tac_features.shape[0] == 0:
            return torch.tensor(0.0, device=tac_features.device, requires_grad=True)
        tac = ops.l2_normalize(_run_head(self.tactile_projection_head, tac_features.contiguous()))
        pc = ops.l2_normalize(_run_head(self.pointcloud_projection_head, pc_features.contiguous()))
        img = ops.l2_normalize(_run_head(self.image_projection_head, img_features.contiguous()))
        inv_t = 1.0 / self.temperature
        logits_pc = ops.BmmNTFn.apply(tac, pc, inv_t)                     # [B, n_arms, 256] fp32
        logits_img = ops.BmmNTFn.apply(tac, img, inv_t)
        loss_pc = ops.cross_entropy(logits_pc.reshape(-1, pc.shape[1]), positive_pc_indices.reshape(-1))
        loss_img = ops.cross_entropy(logits_img.reshape(-1, img.shape[1]), linear_positive_img_indices.reshape(-1))
        return (loss_pc + loss_img) / 2
