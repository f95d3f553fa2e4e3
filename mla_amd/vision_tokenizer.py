"""Encoder-free vision tokenizer (reference: models/mla/image/vision_tokenizer.py:14-159), batched on HIP kernels.

Same module tree / parameter names as the reference (patch_embedding, class_embedding, split_embedding,
local_attention.{q,kv,proj}, global_attention.{q,kv,proj}); GlobalAttention's output is discarded by the reference
(:142,149) so it is never computed here -- its parameters exist for state-dict parity and receive no gradient.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip, ops
from .llama import Linear


class _AttnParams(nn.Module):
    def __init__(self, input_size, with_stride=None, num_heads=8):
        super().__init__()
        if with_stride is not None:
            self.conv_stride = with_stride
        self.num_heads = num_heads
        self.scale = input_size ** -0.5   # NB: input_size, not head_dim (vision_tokenizer.py:19,54)
        self.q = nn.Sequential(nn.LayerNorm(input_size), Linear(input_size, input_size, bias=False))
        self.kv = nn.Sequential(nn.LayerNorm(input_size), Linear(input_size, input_size * 2, bias=False))
        self.proj = Linear(input_size, input_size)


class LocalAttention(_AttnParams):
    def __init__(self, input_size, conv_stride, num_heads=8):
        super().__init__(input_size, conv_stride, num_heads)


class GlobalAttention(_AttnParams):
    def __init__(self, input_size, num_heads=8):
        super().__init__(input_size, None, num_heads)


class MLP_GELU(nn.Module):
    """vision_tokenizer.py:79-89 (projector_2d = MLP_GELU(1024, token_size, 2)); trainable -> autograd ops."""

    def __init__(self, input_size, hidden_size, depth):
        super().__init__()
        layers = [Linear(input_size, hidden_size)]
        for _ in range(1, depth):
            layers.append(nn.GELU())
            layers.append(Linear(hidden_size, hidden_size))
        self.mlp = nn.Sequential(*layers)

    def forward(self, x):
        for m in self.mlp:
            x = ops.act(x, hip.ACT_GELU_ERF) if isinstance(m, nn.GELU) else m(x)
        return x


def pil_resample_tables(in_size: int, out_size: int):
    """Taps of PIL's bicubic resize for one axis, as PIL computes them for 8-bit images (libImaging/Resample.c precompute_coeffs +
    normalize_coeffs_8bpc): bounds [out, 2] = (first input index, tap count), coef [out, ksize] int32 in 2^22 fixed point."""
    import math
    import numpy as np

    def bicubic(x, a=-0.5):
        x = abs(x)
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)               # C (int) cast: truncation toward zero
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k)
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            coef[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, coef


class ClipImagePreprocessor:
    """Stand-in for the reference's CLIPImageProcessor(size=672, crop_size=672, rescale, normalise) (vision_tokenizer.py:98-105) with
    the HF calling convention ``preprocess(image, return_tensors='pt')['pixel_values']``: uint8 frames are resized with PIL-exact
    8-bit bicubic resampling, rescaled and normalised by one HIP kernel (hip.clip_preprocess). Square inputs only (the RLDS pipeline
    delivers 224 x 224): shortest-edge resize to 672 then makes the centre crop the identity."""

    def __init__(self, size=672, device="cuda"):
        self.size = self.crop_size = size
        self.do_resize = self.do_center_crop = self.do_normalize = self.do_rescale = True
        self.image_mean = [0.48145466, 0.4578275, 0.40821073]
        self.image_std = [0.26862954, 0.26130258, 0.27577711]
        self.device = torch.device(device)
        self._tables = {}

    def tables(self, in_size):
        if in_size not in self._tables:
            b, c = pil_resample_tables(in_size, self.size)
            self._tables[in_size] = (torch.from_numpy(b).to(self.device), torch.from_numpy(c).to(self.device))
        return self._tables[in_size]

    def preprocess(self, images, return_tensors="pt", mask_channel=False, out_dtype=torch.float32, **kwargs):
        """images: PIL image / HWC uint8 array / uint8 tensor [H, W, 3] or a batch [B, H, W, 3]. Returns {"pixel_values": [B, 3(+1), 672, 672]}."""
        import numpy as np
        if not torch.is_tensor(images):
            arr = np.asarray(images)
            images = torch.from_numpy(np.ascontiguousarray(arr))
        if images.dim() == 3:
            images = images.unsqueeze(0)
        if images.dtype != torch.uint8 or images.shape[-1] != 3:
            raise TypeError("ClipImagePreprocessor expects uint8 RGB frames in HWC layout")
        if images.shape[1] != images.shape[2]:
            raise NotImplementedError("non-square frames (shortest-edge resize + centre crop): the RLDS pipeline delivers 224 x 224")
        bt, ct = self.tables(int(images.shape[1]))
        out = hip.clip_preprocess(images.to(self.device), bt, ct, bt, ct, self.size, self.size, self.image_mean, self.image_std,
                                  out_dtype=out_dtype, mask_channel=mask_channel)
        return {"pixel_values": out}

    __call__ = preprocess


class VisionTokenizer(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.is_loaded = True
        self.hidden_size = input_size
        # the reference holds a CLIPImageProcessor(size=672, crop 672, rescale, normalise) here (:98-105); same settings, PIL-exact
        # arithmetic, on the GPU
        self.image_processor = ClipImagePreprocessor(672)
        self.patch_stride = 14
        self.conv_stride = 3
        self.patch_embedding = nn.Conv2d(3, input_size, kernel_size=14, stride=14, bias=False)
        self.class_embedding = nn.Parameter(torch.randn(input_size))
        self.split_embedding = nn.Parameter(torch.randn(input_size))
        self.local_attention = LocalAttention(input_size, self.conv_stride)
        self.global_attention = GlobalAttention(input_size)

    @property
    def dtype(self):
        return self.patch_embedding.weight.dtype

    @property
    def device(self):
        return self.patch_embedding.weight.device

    def tokens(self, pixel_values: torch.Tensor, check_mask: bool = True) -> torch.Tensor:
        """[B, 4, Hi, Wi] (RGB + mask channel, fp32 or bf16; 672 x 672 on the training path) -> [B, (Hi/42) * (Wi/42), C] bf16 tokens
        (before the projector). The mask channel is NOT applied here: this is the whole-grid dataflow."""
        B, CT, Hi, Wi = pixel_values.shape
        P, cs, C = self.patch_stride, self.conv_stride, self.hidden_size
        gh, gw = Hi // P, Wi // P
        # the batched path is the all-ones-mask dataflow; the verdict is read back by assert_masks_ok() -- called by the owner at its next
        # host synchronisation point (PrismaticVLM.forward, right after the contrastive index) so that the check does not stall an
        # empty launch queue at the very start of the step. Cropped masks go through forward(..., allow_crop=True) / crop_boxes().
        if check_mask:
            self._mask_ok = (pixel_values[:, -1] == 1).all()
        kreal = 3 * P * P
        kpad = ((kreal + 31) // 32) * 32
        la = self.local_attention
        trainable = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if trainable:
            # stage "pretrain" (models/vlm/prismatic.py:415-447): the same dataflow on autograd ops -- Conv2d(k = s = 14) as a GEMM
            # over im2col rows (its weight gradient flows back through the zero-padding of the reduction axis), LayerNorm /
            # Linear backward, and the two dedicated backward kernels (window attention, average pooling)
            rows = hip.im2col_patch(pixel_values.contiguous(), P, kpad)
            w = F.pad(self.patch_embedding.weight.reshape(C, kreal), (0, kpad - kreal))
            pe = ops.linear(rows, w)
            red = ops.AvgPoolTokensFn.apply(pe, B, gh, gw, cs)
            qv = ops.linear(ops.layernorm(red, la.q[0].weight, la.q[0].bias, la.q[0].eps), la.q[1].weight)
            kv = ops.linear(ops.layernorm(pe, la.kv[0].weight, la.kv[0].bias, la.kv[0].eps), la.kv[1].weight)
            agg = ops.LocalAttnFn.apply(qv, kv, B, gh, gw, cs, la.num_heads, la.scale)
            tok = ops.linear(agg, la.proj.weight, la.proj.bias, residual=red)
            return tok.view(B, (gh // cs) * (gw // cs), C)
        with torch.no_grad():
            rows = hip.im2col_patch(pixel_values.contiguous(), P, kpad)
            w = F.pad(self.patch_embedding.weight.reshape(C, kreal), (0, kpad - kreal)).contiguous()
            pe = hip.gemm(rows, w)                                             # [B*gh*gw, C]
            red = hip.avgpool_tokens(pe, B, gh, gw, cs)                        # [B*256, C]
            qv = hip.gemm(hip.layernorm_fwd(red, la.q[0].weight, la.q[0].bias, la.q[0].eps), la.q[1].weight)
            kv = hip.gemm(hip.layernorm_fwd(pe, la.kv[0].weight, la.kv[0].bias, la.kv[0].eps), la.kv[1].weight)
            agg = hip.local_attn(qv, kv, B, gh, gw, cs, la.num_heads, la.scale)
            tok = hip.gemm(agg, la.proj.weight, bias=la.proj.bias, residual=red)
        return tok.view(B, (gh // cs) * (gw // cs), C)

    def assert_masks_ok(self) -> None:
        ok = getattr(self, "_mask_ok", None)
        if ok is not None:
            self._mask_ok = None
            if not bool(ok):
                raise NotImplementedError("cropped pixel masks inside PrismaticVLM: only the all-ones mask yields the 256 tokens the "
                                          "reference's N_img = 256 layout needs (models/vlm/prismatic.py:932-933); the tokenizer itself "
                                          "handles them: VisionTokenizer.forward(pixel_values, projector, allow_crop=True)")

    def crop_boxes(self, pixel_values: torch.Tensor):
        """Per-sample patch rectangles of the reference's crop (models/mla/image/vision_tokenizer.py:124-137): the pixel mask pooled to
        the patch grid must consist of whole patches (:127); an all-zero mask keeps the top-left 16 x 16 patches (:131-132), any other
        the rectangle spanned by its first and last non-zero patch in row-major order (:134-137). Host-side (one device read):
        returns [(h1, h2, w1, w2)] inclusive, or None when every mask is all ones (the batched path applies)."""
        P = self.patch_stride
        pm = F.avg_pool2d(pixel_values[:, -1:].float(), kernel_size=P, stride=P)[:, 0]
        if bool((pm == 1).all()):
            return None
        pm = pm.cpu()
        if bool((pm % 1 != 0).any()):
            raise ValueError("pixel mask is not made of whole 14 x 14 patches (the reference asserts this, vision_tokenizer.py:127)")
        boxes = []
        for m in pm:
            if float(m.sum()) == 0:
                boxes.append((0, 15, 0, 15))
            else:
                nz = torch.nonzero(m, as_tuple=False)
                boxes.append((int(nz[0, 0]), int(nz[-1, 0]), int(nz[0, 1]), int(nz[-1, 1])))
        return boxes

    def forward(self, pixel_values, modules, repeat: int = 1, allow_crop: bool = False):
        """Reference signature (pixel_values, projector) -> (list of [h * w, token_size] tokens, list of [h, w]).
        ``repeat``: the batch is ``repeat`` tiled copies of its first B/repeat samples (MLA.forward tiles every input
        R times, models/mla/model_mla.py:159-169); the frozen, deterministic tower then runs once per distinct image.
        ``allow_crop``: honour cropped pixel masks like the reference's per-sample loop (:129-150) -- one host read of the pooled
        masks, then every cropped sample runs the same kernels on its own patch rectangle (floor(H/3) x floor(W/3) tokens; the
        average pool and the window unfold both drop the remainder rows / columns, so the rectangle is cut to multiples of 3).
        PrismaticVLM never sets it: its N_img = 256 layout (prismatic.py:932-933) only fits the all-ones mask."""
        B = pixel_values.shape[0]
        boxes = self.crop_boxes(pixel_values) if allow_crop else None
        if boxes is not None:
            P, cs = self.patch_stride, self.conv_stride
            outs, hws = [], []
            for i, (h1, h2, w1, w2) in enumerate(boxes):
                h, w = (h2 - h1 + 1) // cs, (w2 - w1 + 1) // cs
                sub = pixel_values[i:i + 1, :, h1 * P:(h1 + h * cs) * P, w1 * P:(w1 + w * cs) * P]
                outs.append(modules(self.tokens(sub, check_mask=False))[0])
                hws.append(torch.tensor([h, w], dtype=torch.long, device=pixel_values.device))
            return outs, hws
        if repeat > 1:
            tok = self.tokens(pixel_values[: B // repeat]).repeat(repeat, 1, 1)
        else:
            tok = self.tokens(pixel_values)
        out = modules(tok)
        h = w = int(round(out.shape[1] ** 0.5))
        # cached per (h, w, device): building it from a Python list is a pageable host-to-device copy, i.e. a host synchronisation
        # 0.9 ms into every step that leaves the rest of the front end launch-bound
        key = (h, w, out.device)
        cache = self.__dict__.setdefault("_hw_cache", {})
        if key not in cache:
            cache[key] = torch.tensor([h, w], dtype=torch.long, device=out.device)
        hw = cache[key]
        return list(out.unbind(0)), [hw] * B
