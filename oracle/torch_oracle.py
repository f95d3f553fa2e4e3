"""CPU oracle: plain-PyTorch fp32 restatement of the reference's arithmetic for the MLA training hot path.

TEST INFRASTRUCTURE ONLY. Nothing under mla_amd/ may import this module; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU baseline.

Pinned against the real reference: oracle/capture_golden.py imports /root/reference (in the build container) and
stores input/output vectors under tests/golden/; tests/test_oracle_golden.py replays them through this file.

Each function cites the reference lines it restates (paths relative to the reference repo root).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------- Llama decoder
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """transformers/models/llama/modeling_llama.py:85-90 -- cast back to the input dtype BEFORE the weight multiply."""
    in_dtype = x.dtype
    x32 = x.to(torch.float32)
    var = x32.pow(2).mean(-1, keepdim=True)
    x32 = x32 * torch.rsqrt(var + eps)
    return weight * x32.to(in_dtype)


def rope_tables(seq_len: int, dim: int, base: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """modeling_llama.py:96-145: inv_freq = base^(-2i/dim); cos/sin of outer(position, inv_freq), fp32. Returns the
    half tables [S, dim/2] (the reference concatenates the half with itself)."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    pos = torch.arange(seq_len, dtype=torch.float32)
    freqs = torch.outer(pos, inv_freq)
    return freqs.cos(), freqs.sin()


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """modeling_llama.py:177-181"""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q: torch.Tensor, k: torch.Tensor, cos_half: torch.Tensor, sin_half: torch.Tensor):
    """modeling_llama.py:184-208; q,k: [B, H, S, D]; tables [S, D/2]."""
    cos = torch.cat([cos_half, cos_half], -1)[None, None].to(q.dtype)
    sin = torch.cat([sin_half, sin_half], -1)[None, None].to(q.dtype)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def causal_attention(q, k, v, seqlens: Optional[torch.Tensor] = None, zero_pad_rows: bool = True):
    """LlamaAttention.forward math, modeling_llama.py:371-380 (scores/sqrt(D) + causal mask, fp32 softmax, @V).
    q,k,v: [B, H, S, D].  With right padding (seqlens) the flash path un-pads, so pad query rows are ZERO after
    pad_input (modeling_llama.py:531-553) -- reproduced when zero_pad_rows."""
    B, H, S, D = q.shape
    scores = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(D)
    mask = torch.full((S, S), float("-inf"), dtype=scores.dtype).triu(1)
    scores = scores + mask
    if seqlens is not None and not zero_pad_rows:
        # eager path: _update_causal_mask (modeling_llama.py:1081-1103) also masks padded KEY columns for every query
        keypad = torch.arange(S)[None, :] >= seqlens[:, None]               # [B, S]
        scores = scores.masked_fill(keypad[:, None, None, :], float("-inf"))
    p = torch.softmax(scores.float(), dim=-1).to(q.dtype)
    out = torch.matmul(p, v)
    if seqlens is not None and zero_pad_rows:
        valid = (torch.arange(S)[None, :] < seqlens[:, None]).to(out.dtype)  # [B, S]
        out = out * valid[:, None, :, None]
    return out


def swiglu_mlp(x, w_gate, w_up, w_down):
    """LlamaMLP.forward, modeling_llama.py:240 (pretraining_tp == 1 branch)."""
    return F.linear(F.silu(F.linear(x, w_gate)) * F.linear(x, w_up), w_down)


def decoder_layer(x, p: dict, cos_half, sin_half, n_heads: int, eps: float, seqlens=None, zero_pad_rows: bool = True):
    """LlamaDecoderLayer.forward, modeling_llama.py:695-767 (pre-norm residual block). p holds the 9 weights with
    the reference's leaf names (input_layernorm.weight, self_attn.{q,k,v,o}_proj.weight, post_attention_layernorm.weight,
    mlp.{gate,up,down}_proj.weight)."""
    B, S, Hd = x.shape
    D = Hd // n_heads
    h = rmsnorm(x, p["input_layernorm.weight"], eps)
    q = F.linear(h, p["self_attn.q_proj.weight"]).view(B, S, n_heads, D).transpose(1, 2)
    k = F.linear(h, p["self_attn.k_proj.weight"]).view(B, S, n_heads, D).transpose(1, 2)
    v = F.linear(h, p["self_attn.v_proj.weight"]).view(B, S, n_heads, D).transpose(1, 2)
    q, k = apply_rope(q, k, cos_half, sin_half)
    a = causal_attention(q, k, v, seqlens, zero_pad_rows).transpose(1, 2).reshape(B, S, Hd)
    x = x + F.linear(a, p["self_attn.o_proj.weight"])
    h = rmsnorm(x, p["post_attention_layernorm.weight"], eps)
    return x + swiglu_mlp(h, p["mlp.gate_proj.weight"], p["mlp.up_proj.weight"], p["mlp.down_proj.weight"])


def shifted_cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """modeling_llama.py:1255-1269: logits.float(), shift by one, CrossEntropyLoss (mean over labels != -100)."""
    logits = logits.float()
    sl = logits[..., :-1, :].contiguous().view(-1, logits.shape[-1])
    tl = labels[..., 1:].contiguous().view(-1)
    return F.cross_entropy(sl, tl, ignore_index=-100)


# ------------------------------------------------------------------------------------------------- heads / embedders
def mlp_gelu_tanh(x, fc1_w, fc1_b, fc2_w, fc2_b):
    """timm 0.9.10 Mlp(fc1 -> GELU(tanh) -> fc2) as instantiated by ActionEmbedder / FinalLayer,
    models/diffusion/models.py:112-123, 173-189."""
    return F.linear(F.gelu(F.linear(x, fc1_w, fc1_b), approximate="tanh"), fc2_w, fc2_b)


def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: float = 10000.0) -> torch.Tensor:
    """TimestepEmbedder.timestep_embedding, models/diffusion/models.py:42-60: [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def timestep_embedder(t, w0, b0, w2, b2):
    """TimestepEmbedder.forward models/diffusion/models.py:62-65: Linear -> SiLU -> Linear on the sinusoid."""
    return F.linear(F.silu(F.linear(timestep_embedding(t).to(w0.dtype), w0, b0)), w2, b2)


def timm_rms_norm(x, weight, eps: float = 1e-6):
    """timm==0.9.10 RmsNorm.forward (timm v0.9.10 timm/layers/fast_norm.py::rms_norm, reached from
    models/diffusion/models.py:177,187): v = torch.var(x, dim=-1, keepdim=True) -- unbiased and mean-subtracted --
    then x * rsqrt(v + eps) * weight. Rounding points as in the reference's bf16-autocast run: var and var + eps stay in
    x's dtype, torch.rsqrt is on autocast's fp32 list, so the two products are fp32 (identity for fp32 inputs)."""
    v = torch.var(x, dim=-1, keepdim=True)
    r = torch.rsqrt((v + eps).float())
    return x.float() * r * weight.float()


def final_layer(x, norm_w, fc1_w, fc1_b, fc2_w, fc2_b, eps: float = 1e-6):
    """FinalLayer.forward models/diffusion/models.py:186-189: timm 0.9.10 RmsNorm (torch.var based) -> Mlp."""
    n = timm_rms_norm(x, norm_w, eps).to(x.dtype)
    return mlp_gelu_tanh(n, fc1_w, fc1_b, fc2_w, fc2_b)


def mlp_projector(x, w0, b0, w2, b2):
    """MLPProjector util/nn_utils.py:21-34 and MLP_GELU (depth 2) models/mla/image/vision_tokenizer.py:79-89:
    Linear -> GELU(erf) -> Linear."""
    return F.linear(F.gelu(F.linear(x, w0, b0)), w2, b2)


# ------------------------------------------------------------------------------------------------- diffusion
def cosine_beta_schedule(num_steps: int = 100, max_beta: float = 0.999) -> np.ndarray:
    """get_named_beta_schedule('squaredcos_cap_v2') + betas_for_alpha_bar, models/diffusion/gaussian_diffusion.py:115-140
    (float64)."""
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
    return np.array([min(1 - ab((i + 1) / num_steps) / ab(i / num_steps), max_beta) for i in range(num_steps)],
                    dtype=np.float64)


def respaced_betas(num_steps: int = 100) -> np.ndarray:
    """SpacedDiffusion.__init__ models/diffusion/respace.py:75-89 with use_timesteps = all steps: the betas are
    RE-DERIVED from the base process' cumulative products (1 - ac_i / ac_{i-1}), which differs from the base betas in
    the last float64 bits -- and these are the betas the training tables are built from."""
    base_ac = np.cumprod(1.0 - cosine_beta_schedule(num_steps), axis=0)
    last, out = 1.0, []
    for a in base_ac:
        out.append(1 - a / last)
        last = a
    return np.array(out)


def diffusion_tables(num_steps: int = 100):
    """GaussianDiffusion.__init__ tables, gaussian_diffusion.py:166-184 (float64), on the respaced betas."""
    ac = np.cumprod(1.0 - respaced_betas(num_steps), axis=0)
    return np.sqrt(ac), np.sqrt(1.0 - ac)


def ddim_schedule(n_ddim: int, num_steps: int = 100):
    """space_timesteps('ddimN') respace.py:33-43 + SpacedDiffusion.__init__ :75-89: kept base timesteps (first integer stride
    giving exactly N steps) and the cumulative alpha products of the respaced process. Returns (timestep_map, acp, acp_prev)."""
    keep = None
    for stride in range(1, num_steps):
        if len(range(0, num_steps, stride)) == n_ddim:
            keep = list(range(0, num_steps, stride))
            break
    base_ac = np.cumprod(1.0 - cosine_beta_schedule(num_steps), axis=0)
    last, betas = 1.0, []
    for i in keep:
        betas.append(1 - base_ac[i] / last)
        last = base_ac[i]
    acp = np.cumprod(1.0 - np.array(betas), axis=0)
    return keep, acp, np.append(1.0, acp[:-1])


def ddim_sample_loop(eps_model, noise: torch.Tensor, n_ddim: int = 8, clip_denoised: bool = False, num_steps: int = 100):
    """GaussianDiffusion.ddim_sample_loop(eta=0) gaussian_diffusion.py:520-568, 608-688 with the timestep mapping of
    _WrappedModel respace.py:118-129. eps_model(x, base_timesteps) -> epsilon. Tables float64 -> fp32 per element (:869-881)."""
    tmap, acp, acp_prev = ddim_schedule(n_ddim, num_steps)
    x = noise
    f = lambda a, i: torch.tensor(a[i], dtype=torch.float64).float()  # noqa: E731
    for i in reversed(range(n_ddim)):
        t_base = torch.full((x.shape[0],), tmap[i], dtype=torch.long)
        eps_out = eps_model(x, t_base).float()
        rec, recm1 = f(np.sqrt(1.0 / acp), i), f(np.sqrt(1.0 / acp - 1), i)
        x0 = rec * x - recm1 * eps_out
        if clip_denoised:
            x0 = x0.clamp(-1, 1)
        eps = (rec * x - x0) / recm1
        ab_prev = f(acp_prev, i)
        x = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev) * eps
    return x


def q_sample(x0: torch.Tensor, t: torch.Tensor, noise: torch.Tensor, num_steps: int = 100) -> torch.Tensor:
    """GaussianDiffusion.q_sample gaussian_diffusion.py:214-229 with _extract_into_tensor :869-881 (tables -> fp32)."""
    sa, s1 = diffusion_tables(num_steps)
    a = torch.from_numpy(sa)[t].float()
    b = torch.from_numpy(s1)[t].float()
    while a.dim() < x0.dim():
        a, b = a[..., None], b[..., None]
    return a * x0 + b * noise


# ------------------------------------------------------------------------------------------------- action tokenizer
class ActionTokenizerOracle:
    """vla/action_tokenizer.py:13-75 (numpy, float64 bin edges) -- ids only; the string round trip needs the real
    Llama tokenizer and is out of scope."""

    def __init__(self, vocab_size: int = 32000, bins: int = 256, min_action: float = -1.0, max_action: float = 1.0):
        self.vocab_size, self.n_bins = vocab_size, bins
        self.min_action, self.max_action = min_action, max_action
        self.bins = np.linspace(min_action, max_action, bins)
        self.bin_centers = (self.bins[:-1] + self.bins[1:]) / 2.0

    def encode_ids(self, action: np.ndarray) -> np.ndarray:
        a = np.clip(action, a_min=float(self.min_action), a_max=float(self.max_action))
        return self.vocab_size - np.digitize(a, self.bins)

    def decode_token_ids_to_actions(self, ids: np.ndarray) -> np.ndarray:
        d = self.vocab_size - ids
        d = np.clip(d - 1, a_min=0, a_max=self.bin_centers.shape[0] - 1)
        return self.bin_centers[d]


# ------------------------------------------------------------------------------------------------- camera / contrastive
CAMERAS = {  # models/mla/fuser/camera.py:12-52 (K, R, t) and the original image size each projection assumes
    "rlbench_front": dict(
        K=[[-307.7174807, 0.0, 112.0], [0.0, -307.7174807, 112.0], [0.0, 0.0, 1.0]],
        R=[[1.19209290e-07, -4.22617942e-01, -9.06307936e-01], [-1.00000000e+00, -5.96046448e-07, 1.49011612e-07],
           [-5.66244125e-07, 9.06307936e-01, -4.22617912e-01]],
        t=[1.34999919e+00, 3.71546562e-08, 1.57999933e+00], orig=(224, 224)),
    "franka_right": dict(
        K=[[387.414794921875, 0.0, 319.47052001953125], [0.0, 386.8714904785156, 241.13287353515625], [0.0, 0.0, 1.0]],
        R=[[0.91300858, 0.26157042, -0.31304353], [0.39730357, -0.7442472, 0.53688545],
           [-0.09254842, -0.61455433, -0.78342694]],
        t=[0.8591219242556176, -0.5851783639922448, 0.7535876808722389], orig=(480, 640)),
    "franka_front": dict(
        K=[[388.2638244628906, 0.0, 328.3757019042969], [0.0, 387.84130859375, 240.24295043945312], [0.0, 0.0, 1.0]],
        R=[[-0.01750229, 0.95018522, -0.31119403], [0.99984609, 0.01625676, -0.00659609],
           [-0.0012085, -0.31126158, -0.95032351]],
        t=[0.8545415959817313, 0.5748472977587156, 1.0411478820663598], orig=(720, 1280)),
}


def project_points(xyz: torch.Tensor, camera: str, resize=(672, 672), total_stride: int = 42):
    """project_3d_to_2d_672_* models/mla/fuser/contrastive.py:5-131 as called from prismatic.py:611-619
    (patch_stride 14 x conv_stride 3 = 42). Returns (patch_idx [.., 2] int64 (row, col), valid bool)."""
    cam = CAMERAS[camera]
    K = torch.tensor(cam["K"], dtype=torch.float32)
    R = torch.tensor(cam["R"], dtype=torch.float32)
    t = torch.tensor(cam["t"], dtype=torch.float32)
    oh, ow = cam["orig"]
    sx, sy = resize[1] / ow, resize[0] / oh
    Ks = K.clone()
    Ks[0, 0] *= sx; Ks[1, 1] *= sy; Ks[0, 2] *= sx; Ks[1, 2] *= sy
    Rw = R.T
    tw = -Rw @ t
    cam_xyz = xyz @ Rw.T + tw
    uvw = cam_xyz @ Ks.T
    z = uvw[..., 2:]
    xy = uvw[..., :2] / (z + 1e-6)
    row = (xy[..., 1] / total_stride).floor().long()
    col = (xy[..., 0] / total_stride).floor().long()
    ph, pw = resize[0] // total_stride, resize[1] // total_stride
    valid = (z.squeeze(-1) > 0) & (xy[..., 0] >= 0) & (xy[..., 0] < resize[1]) & (xy[..., 1] >= 0) & (xy[..., 1] < resize[0])
    return torch.stack([row.clamp(0, ph - 1), col.clamp(0, pw - 1)], dim=-1), valid


def coordinate_contrastive_loss(img_feat, pc_feat, patch_idx, valid, heads: dict, temperature: float = 0.07):
    """CoordinateAwareContrastiveLoss.forward contrastive.py:185-215. heads: {img,pc}_{0,2}_{w,b}."""
    ip = F.linear(F.relu(F.linear(img_feat, heads["img_0_w"], heads["img_0_b"])), heads["img_2_w"], heads["img_2_b"])
    pp = F.linear(F.relu(F.linear(pc_feat, heads["pc_0_w"], heads["pc_0_b"])), heads["pc_2_w"], heads["pc_2_b"])
    ip = F.normalize(ip, p=2, dim=-1)
    pp = F.normalize(pp, p=2, dim=-1)
    pw = int(img_feat.shape[1] ** 0.5)
    lin = patch_idx[:, :, 0] * pw + patch_idx[:, :, 1]
    tgt = torch.gather(ip, 1, lin.unsqueeze(-1).expand(-1, -1, ip.shape[-1]))
    vp, vt = pp[valid], tgt[valid]
    if vp.shape[0] == 0:
        return torch.tensor(0.0)
    logits = vp @ vt.t() / temperature
    labels = torch.arange(vp.shape[0])
    return (F.cross_entropy(logits, labels) + F.cross_entropy(logits.t(), labels)) / 2


# ------------------------------------------------------------------------------------------------- vision tokenizer
def local_attention(features, w: dict, conv_stride: int = 3, num_heads: int = 8):
    """LocalAttention.forward models/mla/image/vision_tokenizer.py:26-47; features [B, C, H, W]; scale = C**-0.5 (:19).
    w: q_ln_{w,b}, q_w, kv_ln_{w,b}, kv_w, proj_{w,b}."""
    B, C, H, W = features.shape
    cs = conv_stride
    red = F.avg_pool2d(features, kernel_size=cs, stride=cs)
    h, w_ = red.shape[-2:]
    N = cs * cs
    red = red.flatten(2).transpose(-2, -1)
    q = F.linear(F.layer_norm(red, (C,), w["q_ln_w"], w["q_ln_b"]), w["q_w"])
    q = q.reshape(B, h * w_, num_heads, -1).permute(0, 2, 1, 3).unsqueeze(-2)
    f = features.unfold(2, cs, cs).unfold(3, cs, cs).contiguous().view(B, C, h * w_, cs, cs)
    kv_in = f.flatten(3).permute(0, 2, 3, 1)
    kv = F.linear(F.layer_norm(kv_in, (C,), w["kv_ln_w"], w["kv_ln_b"]), w["kv_w"])
    kv = kv.reshape(B, h * w_, N, 2, num_heads, -1).permute(3, 0, 4, 1, 2, 5)
    attn = (q * (C ** -0.5) * kv[0]).sum(-1).softmax(dim=-1)
    agg = (attn.unsqueeze(-1) * kv[1]).sum(-2).transpose(1, 2).reshape(B, h * w_, -1)
    return red + F.linear(agg, w["proj_w"], w["proj_b"])


def vision_tokenizer(pixel_values, w: dict, proj: dict, patch: int = 14):
    """VisionTokenizer.forward vision_tokenizer.py:119-152 for the all-ones-mask case (full 48x48 grid -> 256 tokens),
    followed by MLP_GELU projector_2d. GlobalAttention's result is discarded by the reference (:142,149) and skipped.
    pixel_values [B, 4, 672, 672] (RGB + mask)."""
    rgb = pixel_values[:, :-1]
    pe = F.conv2d(rgb, w["patch_w"], stride=patch)
    toks = local_attention(pe, w)  # [B, 256, C]
    return mlp_projector(toks, proj["w0"], proj["b0"], proj["w2"], proj["b2"])


def vision_tokenizer_cropped(pixel_values, w: dict, proj: dict, patch: int = 14, conv_stride: int = 3):
    """VisionTokenizer.forward vision_tokenizer.py:119-150 INCLUDING the per-sample crop (:124-137): the pixel mask (last channel) is
    average-pooled to the patch grid; a sample whose patch mask is all zero keeps the top-left 16 x 16 patches (:131-132), any other keeps
    the rectangle spanned by the first and the last non-zero patch in row-major order (:134-137). Local attention then runs on that
    sub-grid (floor(H/3) x floor(W/3) tokens). Returns (list of [h*w, token] tensors, list of [h, w]) like the reference."""
    rgb, mask = pixel_values[:, :-1], pixel_values[:, -1:]
    pe = F.conv2d(rgb, w["patch_w"], stride=patch)
    pm = F.avg_pool2d(mask, kernel_size=patch, stride=patch)
    assert len(torch.where(pm % 1)[0]) == 0          # :127 patch masks are whole patches
    toks, hws = [], []
    for i in range(pe.shape[0]):
        if pm[i, 0].sum() == 0:
            sub = pe[i, :, :16, :16]
        else:
            nz = torch.nonzero(pm[i, 0], as_tuple=False)
            (h1, w1), (h2, w2) = nz[0], nz[-1]
            sub = pe[i, :, h1:h2 + 1, w1:w2 + 1]
        Hs, Ws = sub.shape[1:]
        t = local_attention(sub.unsqueeze(0), w, conv_stride)[0]
        toks.append(mlp_projector(t, proj["w0"], proj["b0"], proj["w2"], proj["b2"]))
        hws.append(torch.tensor([Hs // conv_stride, Ws // conv_stride]))
    return toks, hws


# ------------------------------------------------------------------------------------------------- point tokenizer
def furthest_point_sample(xyz: torch.Tensor, npoint: int, start: torch.Tensor) -> torch.Tensor:
    """models/mla/pointcloud/backbone/Point_PN.py:6-21 with the random start index made an explicit input."""
    B, N, _ = xyz.shape
    idx = torch.zeros(B, npoint, dtype=torch.long)
    far = start.clone()
    dist = torch.ones(B, N) * 1e10
    ar = torch.arange(B)
    for i in range(npoint):
        idx[:, i] = far
        c = xyz[ar, far, :].view(B, 1, 3)
        d = torch.sum((xyz - c) ** 2, -1)
        dist = torch.minimum(dist, d)
        far = torch.max(dist, -1)[1]
    return idx


def square_distance(src, dst):
    """Point_PN.py:23-42"""
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d = d + torch.sum(src ** 2, -1)[:, :, None]
    return d + torch.sum(dst ** 2, -1)[:, None, :]


def knn_point(k: int, xyz, new_xyz):
    """Point_PN.py:62-73 (index order inside a group is unspecified: sorted=False)"""
    return torch.topk(square_distance(new_xyz, xyz), k, dim=-1, largest=False, sorted=False)[1]


def index_points(points, idx):
    """Point_PN.py:44-60"""
    B = points.shape[0]
    view = [B] + [1] * (idx.dim() - 1)
    bi = torch.arange(B).view(view).expand_as(idx)
    return points[bi, idx, :]


def _bn_train(x, w, b, eps=1e-5):
    """BatchNorm in train mode (batch statistics, biased variance): the frozen point tower still runs
    self.vlm.train() (training/strategies/base_strategy_mla.py:291). x: [B, C, ...]."""
    dims = [0] + list(range(2, x.dim()))
    mean = x.mean(dims, keepdim=True)
    var = x.var(dims, unbiased=False, keepdim=True)
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean) / torch.sqrt(var + eps) * w.view(shape) + b.view(shape)


def pos_embed_geo(knn_xyz, out_dim: int, alpha: float = 1000.0, beta: float = 100.0):
    """PosE_Geo.forward Point_PN.py:231-249 (position part only). knn_xyz [B, 3, G, K] -> [B, out_dim, G, K]."""
    B, _, G, K = knn_xyz.shape
    fd = out_dim // 6
    rng = torch.arange(fd, dtype=torch.float32)
    dim_embed = torch.pow(torch.tensor(alpha), rng / fd)
    div = (beta * knn_xyz.unsqueeze(-1)) / dim_embed
    pe = torch.cat([torch.sin(div), torch.cos(div)], -1)
    return pe.permute(0, 1, 4, 2, 3).contiguous().view(B, out_dim, G, K)


def point_tokenizer(xyz: torch.Tensor, w: dict, fps_starts, k: int = 81):
    """PointTokenizer.forward pointvit.py:59-82 -> Point_PN_scan/EncP Point_PN.py:284-315 (2 stages, type='scan',
    embed 96, dim_expansion 2,2, LGA blocks 2,1), BatchNorm in train mode. xyz [B, 1024, 3] fp32.
    w keys: raw.{conv_w, bn_w, bn_b}; s{i}.b{j}.{c1_w,c1_b,bn1_w,bn1_b,c2_w,c2_b,bn2_w,bn2_b}; proj_{w,b}.
    Returns tokens [B, 256, 768], centres [B, 256, 3], and the (fps_idx, knn_idx) lists."""
    x = xyz.transpose(1, 2)  # [B, 3, N]
    x = F.relu(_bn_train(F.conv1d(x, w["raw.conv_w"]), w["raw.bn_w"], w["raw.bn_b"]))  # Linear1Layer :173-186
    blocks = [2, 1]
    out_dim, group = 96, xyz.shape[1]
    dbg = []
    for i in range(2):
        out_dim *= 2
        group //= 2
        feats = x.permute(0, 2, 1)  # [B, N, C]
        fps = furthest_point_sample(xyz, group, fps_starts[i])
        lc_xyz, lc_x = index_points(xyz, fps), index_points(feats, fps)
        knn = knn_point(k, xyz, lc_xyz)
        knn_xyz, knn_x = index_points(xyz, knn), index_points(feats, knn)
        dbg.append((fps, knn))
        # LGA 'scan' normalisation, Point_PN.py:125-134
        kx = knn_xyz.permute(0, 3, 1, 2) - lc_xyz.permute(0, 2, 1).unsqueeze(-1)
        mx = torch.abs(kx).max(dim=-1, keepdim=True)[0].clamp(min=1e-6)
        kx = kx / mx  # [B, 3, G, K]
        B, G, K, C = knn_x.shape
        kf = torch.cat([knn_x, lc_x.reshape(B, G, 1, -1).repeat(1, 1, K, 1)], dim=-1).permute(0, 3, 1, 2)
        f = kf + pos_embed_geo(kx, out_dim)
        for j in range(blocks[i]):  # Linear2Layer :189-219
            pfx = f"s{i}.b{j}."
            y = F.relu(_bn_train(F.conv2d(f, w[pfx + "c1_w"][:, :, None, None], w[pfx + "c1_b"]), w[pfx + "bn1_w"], w[pfx + "bn1_b"]))
            y = _bn_train(F.conv2d(y, w[pfx + "c2_w"][:, :, None, None], w[pfx + "c2_b"]), w[pfx + "bn2_w"], w[pfx + "bn2_b"])
            f = F.relu(y + f)
        x = f.max(-1)[0]  # Pooling :166-169 -> [B, C, G]
        xyz = lc_xyz
    tokens = F.linear(x.transpose(1, 2), w["proj_w"], w["proj_b"])
    return tokens, xyz, dbg


# ------------------------------------------------------------------------------------------------- sequence splice
def splice_sequence(input_ids, attention_mask, labels, n_fused: int, n_action: int, eos_id: int = 2):
    """Index arithmetic of PrismaticVLM.forward models/vlm/prismatic.py:981-1038 (training: tag_0 = 2, :882-884).
    Returns per-row gather plan: position k (in the fused sequence) where [proprio, t, x..] is inserted, plus the
    spliced mask and labels. Sequence = [BOS | fused n_fused | text[1:k'] | proprio | t | x(n_action) | text[k':]]."""
    B, L = input_ids.shape
    ins = 2 + n_action
    ks, masks, labs = [], [], []
    for i in range(B):
        pos = torch.where(input_ids[i] == eos_id)[0][-1].item()
        k = pos + n_fused
        ks.append(k)
        m = torch.cat([attention_mask[i, :1], torch.ones(n_fused, dtype=attention_mask.dtype), attention_mask[i, 1:pos],
                       torch.ones(ins, dtype=attention_mask.dtype), attention_mask[i, pos:]])
        lb = torch.cat([labels[i, :1], torch.full((n_fused,), -100, dtype=labels.dtype), labels[i, 1:pos],
                        torch.full((ins,), -100, dtype=labels.dtype), labels[i, pos:]])
        masks.append(m)
        labs.append(lb)
    return torch.tensor(ks), torch.stack(masks), torch.stack(labs)


# ------------------------------------------------------------------------------------------------- image preprocessing (8f-4)
def pil_bicubic_resize_u8(img: np.ndarray, out_size: int) -> np.ndarray:
    """PIL.Image.resize(..., BICUBIC) for uint8 HWC images, restated from Pillow's libImaging/Resample.c (third-party dependency of the
    reference's CLIPImageProcessor; the reference preprocesses every frame with it, vision_tokenizer.py:98-105, datasets.py:52-69):
    per axis, taps of the a = -0.5 cubic around centre (xx + 0.5) * in / out over support 2 (no filter scaling when up-sampling),
    normalised, quantised to 2^22 fixed point; horizontal pass then vertical pass, each rounded to uint8."""
    def taps(n_in, n_out):
        scale = n_in / n_out
        fs = max(scale, 1.0)
        sup = 2.0 * fs
        out = []
        for xx in range(n_out):
            c = (xx + 0.5) * scale
            lo = max(int(c - sup + 0.5), 0)
            hi = min(int(c + sup + 0.5), n_in)
            w = []
            for x in range(lo, hi):
                t = abs((x - c + 0.5) / fs)
                w.append(((1.5 * t - 2.5) * t * t + 1) if t < 1 else ((((t - 5) * t + 8) * t - 4) * -0.5 if t < 2 else 0.0))
            tot = sum(w)
            w = [v / tot for v in w] if tot != 0.0 else w
            out.append((lo, np.array([int(-0.5 + v * 4194304) if v < 0 else int(0.5 + v * 4194304) for v in w], dtype=np.int64)))
        return out
    H, W, _ = img.shape
    tmp = np.zeros((H, out_size, 3), dtype=np.uint8)
    for xx, (lo, k) in enumerate(taps(W, out_size)):
        acc = (img[:, lo:lo + len(k), :].astype(np.int64) * k[None, :, None]).sum(1) + (1 << 21)
        tmp[:, xx, :] = np.clip(acc >> 22, 0, 255)
    out = np.zeros((out_size, out_size, 3), dtype=np.uint8)
    for yy, (lo, k) in enumerate(taps(H, out_size)):
        acc = (tmp[lo:lo + len(k)].astype(np.int64) * k[:, None, None]).sum(0) + (1 << 21)
        out[yy] = np.clip(acc >> 22, 0, 255)
    return out


def clip_preprocess(img_u8: np.ndarray, size: int = 672) -> np.ndarray:
    """CLIPImageProcessor.preprocess for a square uint8 frame: resize (bicubic), rescale by 1/255 (float64 product, then float32),
    normalise in float32 (transformers/image_transforms.py:92-125, 347-400), channels first."""
    r = pil_bicubic_resize_u8(img_u8, size)
    v = (r * 0.00392156862745098).astype(np.float32)
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
    std = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)
    return ((v - mean) / std).transpose(2, 0, 1)
