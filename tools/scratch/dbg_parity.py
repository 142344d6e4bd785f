import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden, t, rel_err
from oracle import ref_cpu as O
from test_model_gpu import tiny_model, TINY_VIT
from ufvideo_amd import ops
m, a, w = tiny_model()
vt = "model.vision_tower.vision_tower."
if any(k.startswith(vt + "vision_model.") for k in w): vt += "vision_model."
video = t(a["video"])
tw = m.get_vision_tower().encode(video.cuda())
with O.bf16_mirror():
    tm = O.siglip_tower(w, TINY_VIT, video, prefix=vt)
print("tower vs mirror", rel_err(tw.cpu(), tm))
proj = m.get_model().mm_projector
y = proj(tw[None])
with O.bf16_mirror():
    ym = O.stc_connector(w, tw.cpu()[None], prefix="model.mm_projector.", downsample=(1,2,2), padding=1, depth=0)
    ym2 = O.stc_connector(w, tm[None], prefix="model.mm_projector.", downsample=(1,2,2), padding=1, depth=0)
print("proj teacher-forced vs mirror", rel_err(y.cpu(), ym), "chain", rel_err(y.cpu(), ym2))
# stage by stage
pk = proj.packed()
h = ops.convert(tw.reshape(-1, 64).contiguous(), torch.bfloat16)
A, (To, Ho, Wo) = ops.conv3d_gather(h, 4, 4, 4, 64, proj.downsample, proj.PADDING)
h1 = ops.gemm(A, pk["samp_w"], bias=pk["samp_b"], act="silu")
import torch.nn.functional as F
with O.bf16_mirror():
    x = O._rb(tw.cpu()[None].float()).view(1,4,4,4,64).permute(0,1,4,2,3).reshape(4,64,4,4)
    x = x.view(1,4,64,4,4).permute(0,2,1,3,4)
    c = F.conv3d(x, O._g(w,"model.mm_projector.","sampler.0.weight"), O._g(w,"model.mm_projector.","sampler.0.bias"), stride=(1,2,2), padding=1)
    c1 = O._rb(F.silu(c))
    nt,nh,nw = c1.shape[2:]
    c1t = c1.permute(0,2,3,4,1).reshape(nt*nh*nw, 64)
print("sampler out", h1.shape, c1t.shape, rel_err(h1.float().cpu(), c1t), (h1.float().cpu()!=c1t).float().mean().item())
d = (h1.float().cpu()-c1t).abs()
i = d.argmax(); print(i//64, i%64, h1.float().cpu().flatten()[i], c1t.flatten()[i], F.silu(c).permute(0,2,3,4,1).reshape(-1,64).flatten()[i])
h2 = ops.gemm(h1, pk["readout"][0][0], bias=pk["readout"][0][1], act="gelu")
with O.bf16_mirror():
    r1 = O._rb(F.gelu(F.linear(h1.float().cpu(), O._g(w,"model.mm_projector.","readout.0.weight"), O._g(w,"model.mm_projector.","readout.0.bias"))))
print("readout0 TF", rel_err(h2.float().cpu(), r1), (h2.float().cpu()!=r1).float().mean().item())
h3 = ops.gemm(h2, pk["readout"][1][0], bias=pk["readout"][1][1], out_dtype=torch.float32)
with O.bf16_mirror():
    r2 = F.linear(h2.float().cpu(), O._g(w,"model.mm_projector.","readout.2.weight"), O._g(w,"model.mm_projector.","readout.2.bias"))
print("readout2 TF", rel_err(h3.cpu(), r2))
