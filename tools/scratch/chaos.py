import sys, torch
sys.path.insert(0, '.')
from oracle import ref_cpu as O
cfg = dict(vocab_size=512, hidden_size=3584, intermediate_size=18944, num_hidden_layers=1, num_attention_heads=28,
           num_key_value_heads=4, rope_theta=1e6, rms_norm_eps=1e-6)
sd = O.make_qwen2_weights(cfg, seed=12)
x = torch.randn(1, 300, 3584, generator=torch.Generator().manual_seed(14)) * 0.5
def st(a, b): d=(a-b).abs(); return f"max {d.max()/b.abs().max():.2e} rms {d.norm()/b.norm():.2e}"
with O.bf16_mirror():
    a = O.qwen2_forward(sd, cfg, x)["hidden_states"][-1]
    b = O.qwen2_forward(sd, cfg, x * (1 + 1e-7 * torch.randn(x.shape)))["hidden_states"][-1]
f = O.qwen2_forward(sd, cfg, x)["hidden_states"][-1]
f2 = O.qwen2_forward(sd, cfg, x * (1 + 1e-7 * torch.randn(x.shape)))["hidden_states"][-1]
print("mirror vs perturbed mirror", st(b, a))
print("fp32 vs perturbed fp32", st(f2, f))
print("mirror vs fp32", st(a, f))
