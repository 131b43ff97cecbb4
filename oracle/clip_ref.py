"""Oracle (test infrastructure): CLIP text encoder, torch-CPU fp32, over a flat transformers-style state dict.

Third-party arithmetic: the reference wraps transformers' ``CLIPTextModel`` / ``CLIPTextModelWithProjection``
(transformers==4.44.2, setup.py:20) for conversion (python_coreml_stable_diffusion/torch2coreml.py:379-441)
with the causal mask patched from -inf to -1e4 (torch2coreml.py:363-377), and calls the result from
pipeline.py:151-175.  The reference holds no golden vector for the encoder's numerics
(swift/StableDiffusionTests pins the tokenizer only) -> **PARITY UNPINNED** against the reference;
``tests/test_oracle.py`` pins this restatement against the transformers version installed here (5.x,
random-init weights, no checkpoint exists offline).

  embeddings   token_embedding[ids] + position_embedding[0..S)
  layer        x += out_proj(softmax(q k^T d^-1/2 + causal(-1e4)) v);  x += fc2(act(fc1(LN2(x))))   (pre-LN)
  outputs      last_hidden_state = final_layer_norm(x_L); hidden_states[-2] = x_{L-1} (SDXL, :416-428);
               pooler_output = last_hidden_state[eos]; text_embeds = text_projection(pooler_output)
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

CONFIGS = {
    # ViT-L-like (quick_gelu, no projection) and OpenCLIP-bigG-like (gelu, projection) miniatures, d_head = 64
    "mini-l": dict(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                   max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=2,
                   architectures=["CLIPTextModel"]),
    "mini-g": dict(vocab_size=1000, hidden_size=192, intermediate_size=768, num_hidden_layers=4, num_attention_heads=3,
                   max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5, eos_token_id=2, projection_dim=64,
                   architectures=["CLIPTextModelWithProjection"]),
    # full-size towers of the BASELINE models (random-init): SD2.1 OpenCLIP-H (23 layers used), SD1.5 CLIP ViT-L
    "openclip-h": dict(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                       num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5,
                       eos_token_id=2, architectures=["CLIPTextModel"]),
}


def param_shapes(cfg):
    sh = OrderedDict()
    d, i = cfg["hidden_size"], cfg["intermediate_size"]
    sh["text_model.embeddings.token_embedding.weight"] = (cfg["vocab_size"], d)
    sh["text_model.embeddings.position_embedding.weight"] = (cfg["max_position_embeddings"], d)
    for l in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{l}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sh[f"{p}.self_attn.{n}.weight"] = (d, d)
            sh[f"{p}.self_attn.{n}.bias"] = (d,)
        for n in ("layer_norm1", "layer_norm2"):
            sh[f"{p}.{n}.weight"] = (d,)
            sh[f"{p}.{n}.bias"] = (d,)
        sh[f"{p}.mlp.fc1.weight"], sh[f"{p}.mlp.fc1.bias"] = (i, d), (i,)
        sh[f"{p}.mlp.fc2.weight"], sh[f"{p}.mlp.fc2.bias"] = (d, i), (d,)
    sh["text_model.final_layer_norm.weight"] = (d,)
    sh["text_model.final_layer_norm.bias"] = (d,)
    if cfg.get("projection_dim"):
        sh["text_projection.weight"] = (cfg["projection_dim"], d)
    return sh


def eos_index(cfg, ids):
    ids = torch.as_tensor(ids).reshape(-1)
    if cfg.get("eos_token_id", 2) == 2:
        return int(ids.argmax())
    return int((ids == cfg["eos_token_id"]).int().argmax())


@torch.no_grad()
def text_encoder_forward(sd, cfg, input_ids):
    """input_ids (1, S) integer tensor -> dict(last_hidden_state, hidden_embeds, pooler_output[, text_embeds])."""
    ids = torch.as_tensor(input_ids).long().reshape(1, -1)
    s_len, d, h = ids.shape[1], cfg["hidden_size"], cfg["num_attention_heads"]
    dh = d // h
    eps = cfg.get("layer_norm_eps", 1e-5)
    x = sd["text_model.embeddings.token_embedding.weight"][ids[0]] + sd["text_model.embeddings.position_embedding.weight"][:s_len]
    mask = torch.full((s_len, s_len), -1e4).triu(1)                            # torch2coreml.py:363-377
    hidden = [x]
    for l in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{l}"
        n = F.layer_norm(x, (d,), sd[f"{p}.layer_norm1.weight"], sd[f"{p}.layer_norm1.bias"], eps)
        q = F.linear(n, sd[f"{p}.self_attn.q_proj.weight"], sd[f"{p}.self_attn.q_proj.bias"]) * dh ** -0.5
        k = F.linear(n, sd[f"{p}.self_attn.k_proj.weight"], sd[f"{p}.self_attn.k_proj.bias"])
        v = F.linear(n, sd[f"{p}.self_attn.v_proj.weight"], sd[f"{p}.self_attn.v_proj.bias"])
        q, k, v = (t.reshape(s_len, h, dh).transpose(0, 1) for t in (q, k, v))
        w = torch.softmax(q @ k.transpose(1, 2) + mask, dim=-1)
        a = (w @ v).transpose(0, 1).reshape(s_len, d)
        x = x + F.linear(a, sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"])
        n = F.layer_norm(x, (d,), sd[f"{p}.layer_norm2.weight"], sd[f"{p}.layer_norm2.bias"], eps)
        m = F.linear(n, sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"])
        m = m * torch.sigmoid(1.702 * m) if cfg["hidden_act"] == "quick_gelu" else F.gelu(m)
        x = x + F.linear(m, sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"])
        hidden.append(x)
    last = F.layer_norm(x, (d,), sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"], eps)
    pooled = last[eos_index(cfg, ids)][None]
    out = dict(last_hidden_state=last[None], hidden_embeds=hidden[-2][None], pooler_output=pooled)
    if "text_projection.weight" in sd:
        out["text_embeds"] = F.linear(pooled, sd["text_projection.weight"])
    return out
