import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch.nn.functional as F
from oracle import ref_cpu as O
from ufvideo_amd import ops
from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM
from ufvideo_amd.model.videorefer_qwen2 import KVCache
cfg = dict(vocab_size=512, hidden_size=3584, intermediate_size=18944, num_hidden_layers=1, num_attention_heads=28,
           num_key_value_heads=4, rope_theta=1e6, rms_norm_eps=1e-6)
sd = O.make_qwen2_weights(cfg, seed=12)
m = VideoReferQwen2ForCausalLM(VideoReferQwen2Config(**cfg, train_mask_decoder=True))
m.load_state_dict(sd, strict=True); m = m.cuda()
S=300; D=3584; H=28; KV=4; hd=128
x = (torch.randn(1, S, D, generator=torch.Generator().manual_seed(14)) * 0.5)[0]
pk = m.model.packed(); L = pk["layers"][0]
def st(name, got, ref):
    g = got.float().cpu(); ref = ref.float()
    d = (g-ref).abs()
    print(f"{name:14s} max/max {d.max()/ref.abs().max():.2e}  rms {d.norm()/ref.norm():.2e}  differing {(d>0).float().mean():.2e}")
xd = x.cuda().clone()
p = "model.layers.0."
with O.bf16_mirror():
    g = lambda n: O._g(sd, p, n)
    h = ops.rmsnorm(xd, L["ln1"], 1e-6)
    hm = O._rb(O.rmsnorm(x, g("input_layernorm.weight"), 1e-6)); st("rmsnorm", h, hm)
    qkv = ops.gemm(h, L["wqkv"], bias=L["bqkv"])
    hc = h.float().cpu()
    qkvm = O._rb(F.linear(hc, torch.cat([g("self_attn.q_proj.weight"), g("self_attn.k_proj.weight"), g("self_attn.v_proj.weight")]),
                 torch.cat([g("self_attn.q_proj.bias"), g("self_attn.k_proj.bias"), g("self_attn.v_proj.bias")]))); st("qkv gemm", qkv, qkvm)
    cache = KVCache(1, 512, 2*KV*hd, "cuda")
    qkv_in = qkv.float().cpu()
    tab = ops.rope_table(pk["inv_freq"], 0, S, hd)
    ops.rope_kv(qkv, S, H, KV, hd, pk["inv_freq"], 0, cache.buf[0], table=tab)
    cos, sin = O.rope_cos_sin(torch.arange(S), hd, 1e6)
    st("rope table cos", tab[:, :hd//2], cos[:, :hd//2]); st("rope table sin", tab[:, hd//2:], sin[:, :hd//2])
    q = qkv_in[:, :H*hd].view(S, H, hd); k = qkv_in[:, H*hd:(H+KV)*hd].view(S, KV, hd); v = qkv_in[:, (H+KV)*hd:].view(S, KV, hd)
    qr = O._rb(q*cos[:,None]+O.rotate_half(q)*sin[:,None]); kr = O._rb(k*cos[:,None]+O.rotate_half(k)*sin[:,None])
    st("rope q", qkv[:, :H*hd], qr.reshape(S,-1)); st("rope k", cache.buf[0][:S, :KV*hd], kr.reshape(S,-1)); st("v copy", cache.buf[0][:S, KV*hd:], v.reshape(S,-1))
    o = torch.empty((S, H*hd), device="cuda", dtype=torch.bfloat16)
    kvb = cache.buf[0]
    ops.attention(qkv, kvb, kvb[:, KV*hd:], 1, H, KV, S, S, hd, (0, qkv.stride(0)), (0, kvb.stride(0)), (0, kvb.stride(0)), causal=True, q_pos0=0, out=o)
    qh = qkv[:, :H*hd].float().cpu().view(S,H,hd).transpose(0,1)
    kh = kvb[:S, :KV*hd].float().cpu().view(S,KV,hd).transpose(0,1).repeat_interleave(H//KV, 0)
    vh = kvb[:S, KV*hd:].float().cpu().view(S,KV,hd).transpose(0,1).repeat_interleave(H//KV, 0)
    att = qh @ kh.transpose(1,2) * hd**-0.5
    att = att.masked_fill(torch.arange(S)[None,:] > torch.arange(S)[:,None], torch.finfo(torch.float32).min)
    om = O._rb(O._softmax_pv(att, vh)).transpose(0,1).reshape(S, H*hd); st("attention", o, om)
    x1 = ops.gemm(o, L["wo"], resid=xd, out_dtype=torch.float32)
    x1m = x + F.linear(o.float().cpu(), g("self_attn.o_proj.weight")); st("o_proj+res", x1, x1m)
    h2 = ops.rmsnorm(x1, L["ln2"], 1e-6)
    h2m = O._rb(O.rmsnorm(x1.cpu(), g("post_attention_layernorm.weight"), 1e-6)); st("rmsnorm2", h2, h2m)
    act = ops.gemm(h2, L["wgu"], swiglu=True)
    h2c = h2.float().cpu()
    gg = F.linear(h2c, g("mlp.gate_proj.weight")); uu = F.linear(h2c, g("mlp.up_proj.weight"))
    actm = O._rb(F.silu(gg)*uu); st("gate/up swiglu", act, actm)
    x2 = ops.gemm(act, L["wd"], resid=x1, out_dtype=torch.float32)
    x2m = x1.cpu() + F.linear(act.float().cpu(), g("mlp.down_proj.weight")); st("down+res", x2, x2m)
    n = m.model.final_norm(x2)
    nm = O.rmsnorm(x2.cpu(), O._g(sd, "model.", "norm.weight"), 1e-6); st("final norm", n, nm)
    full = O.qwen2_forward(sd, cfg, x[None]); st("layer chain", n, full["hidden_states"][-1][0])
