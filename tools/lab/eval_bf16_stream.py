"""Lab (CPU, oracle only): what a bf16 residual stream in the ViT tower would cost in parity.  The 26-layer SigLIP-so400m-dims tower on one 336 x 336 frame:
the fp32 oracle, the bf16 mirror as built (fp32 stream) and the mirror with the stream rounded to bf16 after both residual adds of every layer (what a
bf16 tower keeps, and what would halve the in-place update traffic of out_proj / fc2).  Result (round 4): rel-L2 vs fp32 6.9e-3 -> 1.3e-2, max norm 8.1e-3 -> 2.5e-2."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import ref_cpu as O
torch.manual_seed(0)
vit = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, image_size=336, patch_size=14)
sd = O.make_siglip_weights(vit, seed=61)
x = torch.randn(1, 3, 336, 336, generator=torch.Generator().manual_seed(62))
y32 = O.siglip_tower(sd, vit, x)
with O.bf16_mirror():
    ym = O.siglip_tower(sd, vit, x)
# bf16 stream variant: same as the mirror, but the residual stream is rounded to bf16 after every residual add (what a bf16 tower keeps)
orig = O.vit_encoder_layer
def layer_bf16_stream(sd_, p, x_, heads, eps, act):
    B, N, D = x_.shape; hd = D // heads
    rb = lambda t: t.to(torch.bfloat16).float()
    g = lambda n: rb(sd_[p + n].float())
    h = rb(F.layer_norm(x_, (D,), g("layer_norm1.weight"), g("layer_norm1.bias"), eps))
    q = rb(F.linear(h, g("self_attn.q_proj.weight"), g("self_attn.q_proj.bias"))).view(B, N, heads, hd).transpose(1, 2)
    k = rb(F.linear(h, g("self_attn.k_proj.weight"), g("self_attn.k_proj.bias"))).view(B, N, heads, hd).transpose(1, 2)
    v = rb(F.linear(h, g("self_attn.v_proj.weight"), g("self_attn.v_proj.bias"))).view(B, N, heads, hd).transpose(1, 2)
    with O.bf16_mirror():
        o = rb(O.attention_noncausal(q, k, v, hd ** -0.5)).transpose(1, 2).reshape(B, N, D)
    x_ = rb(x_ + F.linear(o, g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias")))
    h = rb(F.layer_norm(x_, (D,), g("layer_norm2.weight"), g("layer_norm2.bias"), eps))
    h = rb(O._ACT[act](F.linear(h, g("mlp.fc1.weight"), g("mlp.fc1.bias"))))
    return rb(x_ + F.linear(h, g("mlp.fc2.weight"), g("mlp.fc2.bias")))
with O.bf16_mirror():
    hb = O.siglip_embeddings(sd, "", x, 14).to(torch.bfloat16).float()
for i in range(26):
    hb = layer_bf16_stream(sd, f"encoder.layers.{i}.", hb, 16, 1e-6, "gelu_pytorch_tanh")
def l2(a,b): return float((a.double()-b.double()).norm()/b.double().norm())
def mx(a,b): return float((a-b).abs().max()/b.abs().max())
print("fp32-stream mirror vs fp32:  max-norm %.2e rel-L2 %.2e" % (mx(ym,y32), l2(ym,y32)))
print("bf16-stream        vs fp32:  max-norm %.2e rel-L2 %.2e" % (mx(hb,y32), l2(hb,y32)))
