"""Golden vectors that pin the head_dim-128 training kernels (fused flash-style attention backward, csrc/attn_bwd.hip) to the
REFERENCE's own backward -- RUN in the build container.  The reference's decoder class (`VideoReferQwen2ForCausalLM`,
videorefer_qwen2.py:80-127: HF Qwen2ForCausalLM underneath) is built with 2 layers, hidden 256, 2 query heads / 1 kv head of
128, d_ff 512, vocab 300; its weights are `ref_cpu.make_qwen2_weights(cfg, seed)` rounded to bf16 (so the test regenerates them
from the seed instead of storing them); the causal-LM loss of `forward(inference=False)`'s language-model call
(`super().forward(inputs_embeds=..., labels=...)`, videorefer_qwen2.py:200-215) is back-propagated with torch autograd.
Saved: inputs, loss, d(inputs_embeds), the gradients of every bias / norm weight and of layer 0's k_proj / v_proj (they
collect dK / dV of the attention backward), lm_head / embed omitted for size.  Test infrastructure only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_fixtures as GF  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402

LLM = dict(vocab_size=300, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
           rope_theta=10000.0, rms_norm_eps=1e-6)
SEED, S = 31, 200


def bfr(x):
    return x.to(torch.bfloat16).to(torch.float32)


def main():
    RQ = GF.RQ
    cfg = RQ.VideoReferQwen2Config(**{k: v for k, v in LLM.items() if k != "rope_theta"})
    cfg.rope_theta = LLM["rope_theta"]
    # no `mm_vision_tower` attribute: VideoReferMetaModel then builds neither tower nor projector (videorefer_arch.py:36-39)
    for k, v in dict(train_mask_decoder=False, sam_pretrained=None, sam_out_dim=256, num_frames=4, seg_token_id=299).items():
        setattr(cfg, k, v)
    with torch.no_grad():
        model = RQ.VideoReferQwen2ForCausalLM(cfg).eval()
    w = {k: bfr(v * (3.0 if v.ndim == 2 else 1.0)) for k, v in O.make_qwen2_weights(LLM, seed=SEED).items()}
    res = model.load_state_dict(w, strict=False)
    assert not res.unexpected_keys and all(not k.startswith(("model.layers", "model.norm", "model.embed", "lm_head")) for k in res.missing_keys), res
    assert model.config.hidden_size // model.config.num_attention_heads == 128
    g = torch.Generator().manual_seed(SEED + 1)
    emb = bfr(torch.randn(1, S, LLM["hidden_size"], generator=g) * 0.5).requires_grad_(True)
    labels = torch.randint(0, LLM["vocab_size"], (1, S), generator=g)
    labels[:, :7] = -100
    torch.set_grad_enabled(True)
    out = RQ.Qwen2ForCausalLM.forward(model, inputs_embeds=emb, attention_mask=torch.ones(1, S, dtype=torch.long), labels=labels)
    loss = out.loss
    loss.backward()
    params = dict(model.named_parameters())
    keep = [n for n, p in params.items() if n.startswith(("model.layers.", "model.norm")) and p.ndim == 1]
    keep += ["model.layers.0.self_attn.k_proj.weight", "model.layers.0.self_attn.v_proj.weight"]
    fx = {"llm": np.array([LLM[k] for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                                            "num_key_value_heads")]), "seed": np.array(SEED), "weight_scale": np.array(3.0),
          "inputs_embeds": emb.detach(), "labels": labels, "loss": np.array(float(loss)), "d_inputs_embeds": emb.grad.detach()}
    for n in keep:
        fx["g::" + n] = params[n].grad.detach().clone()
    print("loss", float(loss), "kept", len(keep), "tensors;", sum(v.numel() for k, v in fx.items() if hasattr(v, "numel")) * 4 / 1e6, "MB")
    GF.npz("train_grad_hd128", **fx)


if __name__ == "__main__":
    main()
