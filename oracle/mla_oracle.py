"""CPU oracle for one whole MLA diffusion training forward (fp32), written against a reference-named state dict.

TEST INFRASTRUCTURE ONLY (see oracle/torch_oracle.py header). Restates models/mla/model_mla.py:144-234 +
models/vlm/prismatic.py:840-1144 + transformers/models/llama/modeling_llama.py:1181-1317 with the component functions
of torch_oracle. Pinned by tests/golden/mla_tiny_e2e.npz (captured from the real reference by capture_golden.py)."""
import torch
import torch.nn.functional as F

from . import torch_oracle as O


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def vision_weights(sd, pfx="vlm.vision_tower_2d."):
    la = pfx + "local_attention."
    return dict(patch_w=sd[pfx + "patch_embedding.weight"], q_ln_w=sd[la + "q.0.weight"], q_ln_b=sd[la + "q.0.bias"],
                q_w=sd[la + "q.1.weight"], kv_ln_w=sd[la + "kv.0.weight"], kv_ln_b=sd[la + "kv.0.bias"], kv_w=sd[la + "kv.1.weight"],
                proj_w=sd[la + "proj.weight"], proj_b=sd[la + "proj.bias"])


def point_weights(sd, pfx="vlm.vision_tower_3d."):
    e = pfx + "patch_embed.EncP."
    w = {"raw.conv_w": sd[e + "raw_point_embed.net.0.weight"], "raw.bn_w": sd[e + "raw_point_embed.net.1.weight"],
         "raw.bn_b": sd[e + "raw_point_embed.net.1.bias"], "proj_w": sd[pfx + "proj.weight"], "proj_b": sd[pfx + "proj.bias"]}
    for i, nb in enumerate((2, 1)):
        for j in range(nb):
            s, d = f"{e}LGA_list.{i}.linear2.{j}.", f"s{i}.b{j}."
            w[d + "c1_w"] = sd[s + "net1.0.weight"].flatten(1)
            w[d + "c1_b"] = sd[s + "net1.0.bias"]
            w[d + "bn1_w"], w[d + "bn1_b"] = sd[s + "net1.1.weight"], sd[s + "net1.1.bias"]
            w[d + "c2_w"] = sd[s + "net2.0.weight"].flatten(1)
            w[d + "c2_b"] = sd[s + "net2.0.bias"]
            w[d + "bn2_w"], w[d + "bn2_b"] = sd[s + "net2.1.weight"], sd[s + "net2.1.bias"]
    return w


def mla_forward(sd: dict, batch: dict, draws: dict, n_layers: int, n_heads: int, eps: float, R: int, use_pointcloud=True,
                use_contrastive=True, tap: int = 8, zero_pad_rows: bool = True, gen_cfg: dict = None,
                use_tactile: bool = False, gen_tactile: bool = False):
    """zero_pad_rows=True: flash/varlen semantics (pad query rows give zero attention output -- what the GPU reference
    path and the HIP kernels do); False: eager semantics (what the CPU-imported reference does, used to pin this oracle
    against tests/golden). Valid rows and all losses/gradients are identical either way.
    Returns dict(total_loss, diff_mse, contrastive, ce, noise_pred, logits, hidden_states (list), patch_indices, valid)."""
    rep = lambda v: v.repeat(R, *([1] * (v.dim() - 1)))  # noqa: E731
    ids, am, labels = rep(batch["input_ids"]), rep(batch["attention_mask"]), rep(batch["labels"])
    images = rep(batch["images"]["front_image"])
    actions, proprio = rep(batch["actions"]), rep(batch["proprio"])
    noise, t = draws["noise"], draws["timestep"]
    x = O.q_sample(actions[:, -1:, :], t, noise)
    # prismatic.py:873-880 hard-casts to bf16; the fp32 oracle keeps the bf16-rounded VALUES in fp32
    proprio = proprio.to(torch.bfloat16).float()
    x = x.to(torch.bfloat16).float()
    P = "vlm."
    img_tok = O.vision_tokenizer(images, vision_weights(sd), dict(w0=sd[P + "projector_2d.mlp.0.weight"], b0=sd[P + "projector_2d.mlp.0.bias"],
                                                                  w2=sd[P + "projector_2d.mlp.2.weight"], b2=sd[P + "projector_2d.mlp.2.bias"]))
    B = ids.shape[0]
    H = img_tok.shape[-1]
    if use_pointcloud:
        pc = rep(batch["point_cloud"]).float()
        tokens, centers, _ = O.point_tokenizer(pc, point_weights(sd), [draws["fps_start0"], draws["fps_start1"]])
        pc_tok = O.mlp_projector(tokens, sd[P + "projector_3d.projector.0.weight"], sd[P + "projector_3d.projector.0.bias"],
                                 sd[P + "projector_3d.projector.2.weight"], sd[P + "projector_3d.projector.2.bias"])
        patch_idx, valid = O.project_points(centers, batch["camera_name"])
    else:
        pc_tok = torch.zeros(B, 256, H)
        patch_idx = torch.zeros(B, 256, 2, dtype=torch.long)
        valid = torch.zeros(B, 256, dtype=torch.bool)
    pos_pc = lin_img = None
    if use_tactile:
        # prismatic.py:706-750 (one arm): tactile token from tactile_embedder; positives = nearest point centre to the gripper and
        # the image patch that centre projects to
        tact = rep(batch["tactile"]).float()          # not hard-cast by the reference (only proprio / x / t are, prismatic.py:873-880)
        pt = "tactile_embedder"
        tac_tok = O.mlp_gelu_tanh(tact, sd[P + pt + ".mlp.fc1.weight"], sd[P + pt + ".mlp.fc1.bias"], sd[P + pt + ".mlp.fc2.weight"],
                                  sd[P + pt + ".mlp.fc2.bias"]).unsqueeze(1)
        grip = rep(batch["gripper_xyz"]).float().view(B, 1, 3)
        pos_pc = torch.cdist(grip, centers).argmin(dim=2, keepdim=True)                     # [B, 1, 1]
        idx2d = torch.gather(patch_idx.unsqueeze(1), 2, pos_pc.unsqueeze(-1).expand(-1, -1, -1, 2))
        lin_img = idx2d[..., 0] * 16 + idx2d[..., 1]
        fused = torch.cat([pc_tok, img_tok, tac_tok], dim=1)
    else:
        fused = torch.cat([pc_tok, img_tok, torch.zeros(B, 1, H)], dim=1)
    L_ = "vlm.llm_backbone.llm."
    emb = sd[L_ + "model.embed_tokens.weight"][ids]
    z = torch.cat([emb[:, :1], fused, emb[:, 1:]], dim=1)
    pe = lambda n, v: O.mlp_gelu_tanh(v, sd[P + n + ".mlp.fc1.weight"], sd[P + n + ".mlp.fc1.bias"], sd[P + n + ".mlp.fc2.weight"],  # noqa: E731
                                      sd[P + n + ".mlp.fc2.bias"])
    proprio_e, x_e = pe("proprio_embedder", proprio), pe("x_embedder", x)
    t_e = O.timestep_embedder(t.to(torch.bfloat16), sd[P + "t_embedder.mlp.0.weight"], sd[P + "t_embedder.mlp.0.bias"],
                              sd[P + "t_embedder.mlp.2.weight"], sd[P + "t_embedder.mlp.2.bias"]).unsqueeze(1)
    nf = fused.shape[1]
    seqs, masks, labs, ks = [], [], [], []
    for i in range(B):  # prismatic.py:981-1038
        k = torch.where(ids[i] == 2)[0][-1].item() + nf
        ks.append(k)
        seqs.append(torch.cat([z[i, :k], proprio_e[i], t_e[i], x_e[i], z[i, k:]], dim=0))
        m_ins = torch.ones(2 + x_e.shape[1], dtype=torch.bool)
        masks.append(torch.cat([am[i, :1], torch.ones(nf, dtype=torch.bool), am[i, 1:k - nf], m_ins, am[i, k - nf:]]))
        l_ins = torch.full((2 + x_e.shape[1],), -100)
        labs.append(torch.cat([labels[i, :1], torch.full((nf,), -100), labels[i, 1:k - nf], l_ins, labels[i, k - nf:]]))
    h = torch.stack(seqs)
    mask, flabels = torch.stack(masks), torch.stack(labs)
    S = h.shape[1]
    seqlens = mask.sum(-1)
    cos, sin = O.rope_tables(S, H // n_heads)
    hidden = [h]
    for li in range(n_layers):
        p = _sub(sd, f"{L_}model.layers.{li}.")
        h = O.decoder_layer(h, p, cos, sin, n_heads, eps, seqlens, zero_pad_rows)
        hidden.append(h)
    hn = O.rmsnorm(h, sd[L_ + "model.norm.weight"], eps)
    hidden[-1] = hn  # HF replaces the last entry with the normed state (modeling_llama.py:1033-1037)
    logits = F.linear(hn, sd[L_ + "lm_head.weight"]).float()
    ce = O.shifted_cross_entropy(logits, flabels)
    con = torch.tensor(0.0)
    if use_contrastive:
        c = L_ + "coordinate_aware_contrastive_loss_module."
        heads = {f"{a}_{n}_{wb[0]}": sd[f"{c}{m}_projection_head.{n}.{wb}"] for a, m in (("img", "image"), ("pc", "pointcloud"))
                 for n in (0, 2) for wb in ("weight", "bias")}
        tp = hidden[tap]
        con = O.coordinate_contrastive_loss(tp[:, 257:513], tp[:, 1:257], patch_idx, valid, heads)
    fl = O.final_layer(hn, sd[P + "final_layer.norm_final.weight"], sd[P + "final_layer.mlp.fc1.weight"], sd[P + "final_layer.mlp.fc1.bias"],
                       sd[P + "final_layer.mlp.fc2.weight"], sd[P + "final_layer.mlp.fc2.bias"])
    T = x_e.shape[1]
    noise_pred = torch.stack([fl[i, ks[i] + 2: ks[i] + 2 + T] for i in range(B)])
    diff = ((noise_pred - noise) ** 2).mean()
    total = diff
    extra = {}
    tac_con = torch.tensor(0.0)
    if use_tactile and use_contrastive:
        # TactileContrastiveLoss models/mla/fuser/contrastive.py:241-258 on hidden_states[tap]
        c = L_ + "tactile_contrastive_loss_module."
        head = lambda m, v: F.linear(F.relu(F.linear(v, sd[f"{c}{m}_projection_head.0.weight"], sd[f"{c}{m}_projection_head.0.bias"])),  # noqa: E731
                                     sd[f"{c}{m}_projection_head.2.weight"], sd[f"{c}{m}_projection_head.2.bias"])
        tp = hidden[tap]
        tacp = F.normalize(head("tactile", tp[:, 513:514]), p=2, dim=-1)
        pcp = F.normalize(head("pointcloud", tp[:, 1:257]), p=2, dim=-1)
        imp = F.normalize(head("image", tp[:, 257:513]), p=2, dim=-1)
        l_pc = F.cross_entropy((tacp @ pcp.transpose(1, 2) / 0.07).view(-1, 256), pos_pc.view(-1))
        l_im = F.cross_entropy((tacp @ imp.transpose(1, 2) / 0.07).view(-1, 256), lin_img.view(-1))
        tac_con = (l_pc + l_im) / 2
        extra["tactile_contrastive"] = tac_con
    if gen_tactile:
        # TactileGenerationModule models/mla/generation/models.py:418-430 + F.mse_loss (prismatic.py:827-835); joins the total before
        # the contrastive terms (model_mla.py:224-226)
        from oracle import gen_oracle as G
        g = P + "generation_manager.tactile_gen_module."
        mem = F.linear(hn, sd[g + "feature_projector.weight"], sd[g + "feature_projector.bias"])
        dec = G.transformer_decoder(sd[g + "tactile_query"].expand(B, -1, -1), mem, sd, g + "decoder.", 2, 4)
        pred = F.linear(dec.squeeze(1), sd[g + "output_head.weight"], sd[g + "output_head.bias"])
        tgl = F.mse_loss(pred, rep(batch["next_tactile"]).float())
        total = total + tgl
        extra["tactile_gen_loss"] = tgl
    if gen_cfg is not None:
        # post-training heads read the normed last hidden state of ALL positions, padding included (prismatic.py:1077, no
        # memory_key_padding_mask); next frames / clouds are tiled R times (model_mla.py:165-170); losses join before the
        # contrastive term (model_mla.py:218-229)
        from oracle import gen_oracle as G
        img_loss, pc_loss, gx = G.generation_losses(hn, images, rep(batch["next_images"]), rep(batch["next_point_cloud"]), sd, gen_cfg,
                                                    pfx=P + "generation_manager.")
        total = total + img_loss + pc_loss
        extra = dict(image_gen_loss=img_loss, point_cloud_gen_loss=pc_loss, gen=gx)
    return dict(total_loss=total + con + tac_con, diff_mse=diff, contrastive=con, ce=ce, noise_pred=noise_pred, logits=logits,
                hidden_states=hidden, patch_indices=patch_idx, valid=valid, mask=mask, labels=flabels, ks=ks, **extra)
