"""MLA wrapper (reference: models/mla/model_mla.py:47-309): diffusion branch of forward + wrap policy + freeze."""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

import torch
import torch.nn as nn

from .action_tokenizer import ActionTokenizer
from .diffusion import create_diffusion
from .prismatic import PrismaticVLM

IGNORE_INDEX = -100


class MLA(nn.Module):
    def __init__(self, vlm: PrismaticVLM, action_tokenizer: Optional[ActionTokenizer] = None, token_size: int = 4096,
                 action_dim: int = 7, future_action_window_size: int = 15, past_action_window_size: int = 0, use_ema: bool = False,
                 norm_stats=None, use_diff: bool = False, use_pointcloud: bool = False, use_tactile: bool = False,
                 use_contrastive: bool = False, use_generation: bool = False, gen_image: bool = False, use_roi: bool = False,
                 gen_pointcloud: bool = False, gen_tactile: bool = False, **kwargs) -> None:
        super().__init__()
        self.action_tokenizer = action_tokenizer
        self.use_diff, self.use_pointcloud, self.use_tactile = use_diff, use_pointcloud, use_tactile
        self.use_contrastive, self.use_generation = use_contrastive, use_generation
        self.gen_image, self.use_roi, self.gen_pointcloud, self.gen_tactile = gen_image, use_roi, gen_pointcloud, gen_tactile
        self.vlm = vlm
        self.future_action_window_size = future_action_window_size
        self.vlm.future_action_window_size = future_action_window_size
        self.past_action_window_size = past_action_window_size
        self.all_module_keys = ["vlm." + k for k in self.vlm.all_module_keys]
        if use_ema:
            raise NotImplementedError("use_ema is non-functional in the reference (no ema_diffusion is ever built, "
                                      "model_mla.py:47-97, 305-309); keep it False")
        self.use_ema = use_ema
        self.norm_stats = norm_stats
        self._trainable_module_keys: List[str] = []
        self.last_diff_mse = None
        if self.use_diff:
            self.ddim_diffusion = None
            # round 6, opt-in: run the R diffusion copies of a sample as ONE [prefix | R suffix groups] sequence where the prefix does not depend
            # on the copy (PrismaticVLM.shared_prefix_ok(): the scripts/pretrain.sh configuration); see _forward_shared_prefix
            self.share_prefix = False
            # opt-in (round 6): in the diffusion branch only the action read-out rows of the final hidden state are read; with this
            # flag the last decoder layer computes its row-wise half (o_proj, MLP) on those rows alone (ops.ReadoutLayerFn)
            self.readout_rows_only = os.environ.get("MLA_READOUT_ROWS", "0") != "0"
            self.diffusion_steps = 100
            self.diffusion = create_diffusion(timestep_respacing="", noise_schedule="squaredcos_cap_v2", diffusion_steps=100,
                                              sigma_small=True, learn_sigma=False)

    @classmethod
    def from_pretrained(cls, action_tokenizer, pretrained_checkpoint, model_id: str, llm_backbone, enable_mixed_precision_training: bool = True,
                        arch_specifier: str = "gelu-mlp", freeze_weights: bool = True, action_dim: int = 7,
                        future_action_window_size: int = 15, past_action_window_size: int = 0, use_ema: bool = False, norm_stats=None,
                        class_dropout_prob: float = 0.0, use_diff: bool = False, use_pointcloud: bool = False, use_tactile: bool = False,
                        use_contrastive: bool = False, use_generation: bool = False, gen_image: bool = False, use_roi: bool = False,
                        gen_pointcloud: bool = False, gen_tactile: bool = False, **kwargs) -> "MLA":
        """model_mla.py:311-492: build the VLM, then load the per-module state dicts of a ``{"model": {...}}`` checkpoint with
        the reference's rules (missing optional modules keep their initialisation; the LLM loads non-strictly; embedders load
        only when their input width matches ``action_dim``; generation sub-modules load by key prefix)."""
        token_size = llm_backbone.llm.lm_head.in_features
        vlm = PrismaticVLM(model_id, llm_backbone, enable_mixed_precision_training=enable_mixed_precision_training,
                           class_dropout_prob=class_dropout_prob, use_diff=use_diff, action_dim=action_dim, token_size=token_size,
                           use_pointcloud=use_pointcloud, use_tactile=use_tactile, use_contrastive=use_contrastive,
                           use_generation=use_generation, gen_image=gen_image, use_roi=use_roi, gen_pointcloud=gen_pointcloud,
                           gen_tactile=gen_tactile, **kwargs)
        sd = torch.load(pretrained_checkpoint, map_location="cpu")["model"]
        loaded = []

        def load(name, module, strict=True):
            module.load_state_dict(sd[name], strict=strict)
            loaded.append(name)

        if "vision_tower_2d" in sd:
            load("vision_tower_2d", vlm.vision_tower_2d)
        if "projector_2d" in sd:
            load("projector_2d", vlm.projector_2d)
        if use_pointcloud and "vision_tower_3d" in sd:
            load("vision_tower_3d", vlm.vision_tower_3d)
        if use_pointcloud and "projector_3d" in sd:
            load("projector_3d", vlm.projector_3d)
        assert "llm_backbone" in sd, "PrismaticVLM `from_pretrained` expects checkpoint with keys for `llm_backbone`!"
        load("llm_backbone", vlm.llm_backbone, strict=False)
        if "proprio_embedder" in sd and sd["proprio_embedder"]["mlp.fc1.weight"].shape[-1] == action_dim:
            load("proprio_embedder", vlm.proprio_embedder)
        tactile_dim = 24 if action_dim == 14 else 12                                         # model_mla.py:405-409
        if use_tactile and "tactile_embedder" in sd and sd["tactile_embedder"]["mlp.fc1.weight"].shape[-1] == tactile_dim:
            load("tactile_embedder", vlm.tactile_embedder)
        if use_diff and all(k in sd for k in ("x_embedder", "t_embedder", "final_layer")):
            if sd["x_embedder"]["mlp.fc1.weight"].shape[-1] == action_dim:
                for k in ("x_embedder", "t_embedder", "final_layer"):
                    load(k, getattr(vlm, k))
        if use_generation and "generation_manager" in sd:
            for flag, sub in ((gen_image, "image_gen_module"), (gen_pointcloud, "pointcloud_gen_module"), (gen_tactile, "tactile_gen_module")):
                part = {k[len(sub) + 1:]: v for k, v in sd["generation_manager"].items() if k.startswith(sub + ".")}
                if flag and part:
                    getattr(vlm.generation_manager, sub).load_state_dict(part)
                    loaded.append("generation_manager." + sub)
        if freeze_weights:
            vlm.requires_grad_(False)
            vlm.eval()
        model = cls(vlm, action_tokenizer, token_size=token_size, action_dim=action_dim,
                    future_action_window_size=future_action_window_size, past_action_window_size=past_action_window_size, use_ema=use_ema,
                    norm_stats=norm_stats, use_diff=use_diff, use_pointcloud=use_pointcloud, use_tactile=use_tactile,
                    use_contrastive=use_contrastive, use_generation=use_generation, gen_image=gen_image, use_roi=use_roi,
                    gen_pointcloud=gen_pointcloud, gen_tactile=gen_tactile)
        model.loaded_module_keys = loaded
        return model

    @property
    def trainable_module_keys(self) -> List[str]:
        return ["vlm." + k for k in self.vlm.trainable_module_keys] + self._trainable_module_keys

    @property
    def llm_backbone(self):
        return self.vlm.llm_backbone

    def freeze_backbones(self, stage):
        self.vlm.freeze_backbones(stage)

    def get_fsdp_wrapping_policy(self) -> Callable:
        """model_mla.py:279-303 (same class sets as the VLM's policy)."""
        return self.vlm.get_fsdp_wrapping_policy()

    def forward(self, input_ids=None, attention_mask=None, images=None, next_images=None, camera_name=None, point_cloud=None,
                next_point_cloud=None, tactile=None, next_tactile=None, labels=None, actions=None, proprio=None, gripper_xyz=None,
                inputs_embeds=None, past_key_values=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, repeated_diffusion_steps: int = 4, action_masks=None, use_diff: Optional[bool] = None,
                noise: Optional[torch.Tensor] = None, timestep: Optional[torch.Tensor] = None) -> Tuple[Dict, object]:
        """model_mla.py:118-234. ``noise`` / ``timestep`` (not in the reference signature) let tests inject the random
        draws; when omitted they are drawn in the reference's order: randn_like(actions_future) then randint (:178-179)."""
        if use_diff is not None:
            self.use_diff = use_diff
        if not self.use_diff:
            raise NotImplementedError("the autoregressive branch is dead code in the reference (SURVEY Appendix A #15)")
        R = repeated_diffusion_steps
        rep = lambda v: v.repeat(R, *([1] * (v.ndimension() - 1)))  # noqa: E731
        if getattr(self, "share_prefix", False) and R > 1 and self.training and self.vlm.shared_prefix_ok():
            return self._forward_shared_prefix(input_ids, attention_mask, images, camera_name, labels, actions, proprio, R, noise, timestep)
        self.vlm.readout_rows_only = bool(getattr(self, "readout_rows_only", False))
        proprio = rep(proprio)
        actions = rep(actions)
        actions_future = actions[:, -(self.future_action_window_size + 1):, :]
        input_ids, attention_mask, labels = rep(input_ids), rep(attention_mask), rep(labels)
        if action_masks is not None:
            action_masks = rep(action_masks)
        images = {k_: rep(v) for k_, v in images.items()} if isinstance(images, dict) else rep(images)
        if self.use_generation and self.gen_image:
            next_images = rep(next_images)
        if self.use_pointcloud:
            point_cloud = rep(point_cloud)
        if self.use_pointcloud and self.use_generation and self.gen_pointcloud:
            next_point_cloud = rep(next_point_cloud)
        if self.use_tactile:                                              # model_mla.py:172-176
            tactile, gripper_xyz = rep(tactile), rep(gripper_xyz)
        if self.use_generation and self.gen_tactile:                     # (the reference tiles it only when use_tactile is set too)
            next_tactile = rep(next_tactile)
        if noise is None:
            noise = torch.randn_like(actions_future)
        if timestep is None:
            timestep = torch.randint(0, self.diffusion.num_timesteps, (actions_future.size(0),), device=actions.device)
        x = self.diffusion.q_sample(actions_future, timestep, noise)

        self.vlm.image_repeat_hint = R
        try:
            output, noise_pred, generation_outputs, generation_losses = self.vlm(
                input_ids=input_ids, attention_mask=attention_mask, images=images, next_images=next_images,
                camera_name=camera_name, point_cloud=point_cloud if self.use_pointcloud else None,
                next_point_cloud=next_point_cloud, tactile=tactile, next_tactile=next_tactile, labels=labels, x=x, t=timestep,
                proprio=proprio, gripper_xyz=gripper_xyz, use_cache=use_cache, output_attentions=output_attentions,
                output_hidden_states=output_hidden_states, return_dict=return_dict, use_diff=self.use_diff)
        finally:
            self.vlm.image_repeat_hint = 1
        assert noise_pred.shape == noise.shape == actions_future.shape
        zero = lambda: torch.tensor(0, dtype=torch.float32)  # noqa: E731
        loss_dict = {"total_loss": zero(), "img_pc_contrastive_loss": zero(), "tactile_contrastive_loss": zero(),
                     "diff_loss": zero(), "image_gen_loss": zero(), "point_cloud_gen_loss": zero(), "tactile_gen_loss": zero()}
        diff_loss = ((noise_pred.float() - noise.float()) ** 2).mean()
        self.last_diff_mse = diff_loss.detach().clone()
        total = diff_loss
        if self.use_generation and self.gen_image:                       # model_mla.py:218-223 (generation terms come first)
            loss_dict["image_gen_loss"] = generation_losses["image_gen_loss"]
            total = total + generation_losses["image_gen_loss"].float()
        if self.use_generation and self.gen_pointcloud:
            loss_dict["point_cloud_gen_loss"] = generation_losses["point_cloud_gen_loss"]
            total = total + generation_losses["point_cloud_gen_loss"].float()
        if self.use_generation and self.gen_tactile:
            loss_dict["tactile_gen_loss"] = generation_losses["tactile_gen_loss"]
            total = total + generation_losses["tactile_gen_loss"].float()
        if self.use_contrastive:
            loss_dict["img_pc_contrastive_loss"] = output.img_pc_contrastive_loss
            total = total + output.img_pc_contrastive_loss.float()
            if self.use_tactile:
                loss_dict["tactile_contrastive_loss"] = output.tactile_contrastive_loss
                total = total + output.tactile_contrastive_loss.float()
        # the reference's `total_loss` and `diff_loss` are one tensor mutated in place (model_mla.py:215-229), so the
        # reported diff_loss equals total_loss; the true diffusion MSE is kept in self.last_diff_mse
        loss_dict["total_loss"] = total
        loss_dict["diff_loss"] = total
        return loss_dict, output

    def _forward_shared_prefix(self, input_ids, attention_mask, images, camera_name, labels, actions, proprio, R, noise, timestep):
        """Opt-in (`mla.share_prefix = True`; round 6): the diffusion branch without tiling the sample R times. Only the actions are
        tiled -- noise and timesteps are drawn exactly like in forward() (randn_like(actions_future) then randint, :178-179), so the
        same RNG stream gives the same x_t per copy -- and PrismaticVLM.forward_shared_prefix runs [prefix | R suffix groups] once per
        sample. The loss dict is forward()'s for this configuration (no contrastive / generation terms); `output` is in the shared
        layout."""
        rep = lambda v: v.repeat(R, *([1] * (v.ndimension() - 1)))  # noqa: E731
        actions_future = rep(actions)[:, -(self.future_action_window_size + 1):, :]
        if noise is None:
            noise = torch.randn_like(actions_future)
        if timestep is None:
            timestep = torch.randint(0, self.diffusion.num_timesteps, (actions_future.size(0),), device=actions.device)
        x = self.diffusion.q_sample(actions_future, timestep, noise)
        self.vlm.image_repeat_hint = 1
        output, noise_pred = self.vlm.forward_shared_prefix(x, timestep, R, proprio, input_ids, attention_mask, images, camera_name, labels)
        assert noise_pred.shape == noise.shape == actions_future.shape
        zero = lambda: torch.tensor(0, dtype=torch.float32)  # noqa: E731
        loss_dict = {"total_loss": zero(), "img_pc_contrastive_loss": zero(), "tactile_contrastive_loss": zero(),
                     "diff_loss": zero(), "image_gen_loss": zero(), "point_cloud_gen_loss": zero(), "tactile_gen_loss": zero()}
        total = ((noise_pred.float() - noise.float()) ** 2).mean()
        self.last_diff_mse = total.detach().clone()
        loss_dict["total_loss"] = total
        loss_dict["diff_loss"] = total
        return loss_dict, output

    # ------------------------------------------------------------------------------------------ inference
    def create_ddim(self, ddim_step=10, noise_schedule="squaredcos_cap_v2", diffusion_steps=100):
        """model_mla.py:1166-1173."""
        self.ddim_diffusion = create_diffusion(timestep_respacing="ddim" + str(ddim_step), noise_schedule=noise_schedule,
                                               diffusion_steps=diffusion_steps, sigma_small=True, learn_sigma=False)
        return self.ddim_diffusion

    @staticmethod
    def _check_unnorm_key(norm_stats, unnorm_key):
        if unnorm_key is None:
            assert len(norm_stats) == 1, ("Your model was trained on more than one dataset, please pass a `unnorm_key` from the "
                                          f"following options to choose the statistics used for un-normalizing actions: {norm_stats.keys()}")
            unnorm_key = next(iter(norm_stats.keys()))
        assert unnorm_key in norm_stats, f"The `unnorm_key` you chose is not in the set of available dataset statistics, please choose from: {norm_stats.keys()}"
        return unnorm_key

    def get_action_dim(self, unnorm_key=None):
        return len(self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]["q01"])

    def get_proprio_stats(self, unnorm_key=None):
        return self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["proprio"]

    def get_action_stats(self, unnorm_key=None):
        return self.norm_stats[self._check_unnorm_key(self.norm_stats, unnorm_key)]["action"]

    def normalize_proprio(self, cur_robot_state, unnorm_key=None) -> np.ndarray:
        """model_mla.py:667-677: q01/q99 -> [-1, 1] on the masked dimensions, clipped."""
        st = self.get_proprio_stats(unnorm_key)
        mask = st.get("mask", np.ones_like(st["q01"], dtype=bool))
        hi, lo = np.array(st["q99"]), np.array(st["q01"])
        return np.clip(np.where(mask, 2 * (cur_robot_state - lo) / (hi - lo + 1e-8) - 1, cur_robot_state), -1, 1)

    def unnormalize_actions(self, normalized_actions: np.ndarray, unnorm_key=None) -> np.ndarray:
        """model_mla.py:679-704: clip to [-1, 1], binarise the gripper channel(s) at 0.5, map back through q01/q99."""
        st = self.get_action_stats(unnorm_key)
        mask = st.get("mask", np.ones_like(st["q01"], dtype=bool))
        hi, lo = np.array(st["q99"]), np.array(st["q01"])
        a = np.clip(normalized_actions, -1, 1)
        width = a.shape[-1] if a.ndim >= 1 else 0
        for g in ((6,) if width == 7 else (6, 13) if width == 14 else ()):
            a[..., g] = np.where(a[..., g] < 0.5, 0, 1)
        return np.where(mask, 0.5 * (a + 1) * (hi - lo) + lo, a)

    @torch.inference_mode()
    def predict_action_diff(self, image=None, pointcloud=None, instruction: Optional[str] = None, cur_robot_state=None,
                            unnorm_key: Optional[str] = None, cfg_scale: float = 0.0, use_ddim: bool = True, num_ddim_steps: int = 8,
                            action_dim: int = 7, *, input_ids: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                            camera_name: str = "rlbench_front", reuse_prefix: bool = True, **kwargs) -> np.ndarray:
        """model_mla.py:592-775: 8-step DDIM (eta = 0) over the action chunk with the VLM as the epsilon model, then
        un-normalisation.
        * ``image`` is a PIL image / uint8 HWC frame (pre-processed here like the reference does, :656-660) or an already
          pre-processed float tensor [3|4, 672, 672]; a ones mask channel is appended when missing;
        * the prompt is built from ``instruction`` with the backbone's prompt builder and tokenizer like the reference does
          (:626-632; the Llama tokenizer files are not in this image, so a tokenizer has to be attached), or arrives tokenised as
          ``input_ids`` [1, L]; [29871, 32001, 32002, 29871] is appended unless the last id already is 29871 and the last three ids
          are dropped again (:640-645, :711-713).
        ``reuse_prefix`` (default, round 6): the encoders and the decoder rows in front of the [t, x] tokens are computed ONCE per action
        chunk and every sampler step runs over the 1 + T suffix rows against the cached keys / values (mla_amd/infer.py: same function,
        FPS start indices drawn once per chunk instead of once per step); False = the reference's control flow, a whole forward per step.
        ``noise`` optionally fixes the initial sample (the reference draws it with torch.randn, :707). ``camera_name``: the shipped
        method does not forward it, so the reference's get_camera_params(None) raises (camera.py:54-56); it is an explicit
        argument here (the evaluation scripts use the RLBench front camera)."""
        self.vlm.eval()
        device = next(self.vlm.parameters()).device
        if input_ids is None:
            # :626-632 -- prompt text from the backbone's builder, ids from the backbone's tokenizer (the Llama tokenizer files are not
            # in this image: attach one as vlm.llm_backbone.tokenizer, or pass input_ids)
            tokenizer = getattr(self.vlm.llm_backbone, "tokenizer", None)
            if instruction is None or tokenizer is None or not callable(tokenizer):
                raise ValueError("predict_action_diff needs `input_ids`, or `instruction` plus a callable vlm.llm_backbone.tokenizer")
            builder = self.vlm.llm_backbone.prompt_builder_fn("openvla")
            builder.add_turn(role="human", message=f"What action should the robot take to {instruction.lower()}?")
            input_ids = tokenizer(builder.get_prompt(), truncation=True, return_tensors="pt").input_ids
        if cfg_scale > 1.0:
            raise NotImplementedError("classifier-free guidance: the reference calls self.vlm.forward_with_cfg (model_mla.py:718-729), which "
                                      "PrismaticVLM does not define -- cfg_scale > 1 raises there too; the shipped evaluation uses cfg_scale=0")
        if not (torch.is_tensor(image) and image.is_floating_point()):
            # PIL image / uint8 HWC frame: the reference's CLIPImageProcessor step (:656-657), PIL-exact on the GPU
            image = self.vlm.get_vision_tower_2d().image_processor.preprocess(image, return_tensors="pt")["pixel_values"][0]
        input_ids = input_ids.to(device)
        if not bool(torch.all(input_ids[:, -1] == 29871)):
            tail = torch.tensor([[29871, 32001, 32002, 29871]], dtype=torch.long, device=device)
            input_ids = torch.cat((input_ids, tail), dim=1)[:, :-3]
        img = image.to(device)
        if img.dim() == 3:
            img = img.unsqueeze(0)
        if img.shape[1] == 3:
            img = torch.cat([img, torch.ones_like(img[:, :1])], dim=1)
        if isinstance(pointcloud, np.ndarray):
            pointcloud = torch.from_numpy(pointcloud)
        if pointcloud is not None:
            pointcloud = pointcloud.to(device).contiguous()
            if pointcloud.dim() == 2:
                pointcloud = pointcloud.unsqueeze(0)
        model_kwargs = {"input_ids": input_ids, "images": img, "point_cloud": pointcloud, "camera_name": camera_name}
        if cur_robot_state is not None:
            st = self.normalize_proprio(np.asarray(cur_robot_state), unnorm_key) if self.norm_stats is not None else np.asarray(cur_robot_state)
            model_kwargs["proprio"] = torch.tensor(st, dtype=torch.float32).reshape(1, 1, -1).to(device)
        else:
            raise ValueError("cur_robot_state is required: the proprio token is always spliced in (prismatic.py:985-990)")
        if noise is None:
            noise = torch.randn(1, self.future_action_window_size + 1, action_dim, device=device)
        _ = torch.randint(0, self.diffusion.num_timesteps, (self.future_action_window_size + 1,), device=device)  # drawn, unused (:708)
        eps_model = self.vlm.forward
        if reuse_prefix:
            from .infer import PrefixCachedEps
            eps_model = PrefixCachedEps.for_inputs(self.vlm, n_action_rows=self.future_action_window_size + 1, **model_kwargs)
        if use_ddim and num_ddim_steps is not None:
            if self.ddim_diffusion is None:
                self.create_ddim(ddim_step=num_ddim_steps)
            samples = self.ddim_diffusion.ddim_sample_loop(eps_model, noise.shape, noise.to(device).float(), clip_denoised=False,
                                                           model_kwargs=model_kwargs, progress=False, device=device, eta=0.0)
        else:
            samples = self.diffusion.p_sample_loop(eps_model, noise.shape, noise.to(device).float(), clip_denoised=False,
                                                   model_kwargs=model_kwargs, progress=False, device=device)
        normalized = samples[0].float().cpu().numpy()
        return self.unnormalize_actions(normalized, unnorm_key) if self.norm_stats is not None else normalized
