"""timm's RegStage (ufvideo/model/projector.py:153-161,176-184) is absent offline, so `oracle.regstage_block` is PARITY UNPINNED.  The one
independent implementation of the RegNet-Y bottleneck in this image is HF transformers' `RegNetYLayer`
(transformers/models/regnet/modeling_regnet.py): same topology -- 1x1 conv+norm+act -> grouped 3x3 conv+norm+act -> squeeze-excite with
round(in_channels / 4) hidden units -> 1x1 conv+norm -> + shortcut (1x1 conv+norm when the width changes) -> act.  This test runs THAT
module's forward with the oracle's weights, with the two things timm's configuration changes swapped in (norm layer BatchNorm2d ->
LayerNorm2d over channels, activations ReLU -> SiLU; groups_width 1 = timm's group_size 1 = depthwise) and requires the same output as
the oracle's restatement.  A corroboration of the graph, the SE width rule and the placement of every norm / activation; NOT a pin
(the module is not timm's, and LayerNorm2d's eps 1e-5 vs 1e-6 stays the parameter it is in the oracle)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import ref_cpu as O

regnet = pytest.importorskip("transformers.models.regnet.modeling_regnet")
from transformers import RegNetConfig  # noqa: E402


class LayerNorm2d(nn.Module):
    """LayerNorm over the channel dimension of NCHW (what timm.layers.LayerNorm2d computes)."""

    def __init__(self, w, b, eps):
        super().__init__()
        self.w, self.b, self.eps = w, b, eps

    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), self.w, self.b, self.eps).permute(0, 3, 1, 2)


def hf_block(sd, p, cin, cout, eps):
    cfg = RegNetConfig(groups_width=1, hidden_act="silu")
    blk = regnet.RegNetYLayer(cfg, cin, cout, stride=1).eval()
    convs = [blk.layer[0], blk.layer[1], blk.layer[3]]
    for name, cl in zip(("conv1", "conv2", "conv3"), convs):
        assert cl.convolution.weight.shape == sd[p + name + ".conv.weight"].shape, name      # 1x1, depthwise 3x3 (groups = cout), 1x1
        cl.convolution.weight.data.copy_(sd[p + name + ".conv.weight"])
        cl.normalization = LayerNorm2d(sd[p + name + ".bn.weight"], sd[p + name + ".bn.bias"], eps)
    assert blk.layer[1].convolution.groups == cout and blk.layer[1].convolution.padding == (1, 1)
    se = blk.layer[2].attention
    assert se[0].weight.shape == sd[p + "se.fc1.weight"].shape                                # hidden width round(in / 4): the rule the oracle uses
    se[0].weight.data.copy_(sd[p + "se.fc1.weight"]); se[0].bias.data.copy_(sd[p + "se.fc1.bias"])
    se[2].weight.data.copy_(sd[p + "se.fc2.weight"]); se[2].bias.data.copy_(sd[p + "se.fc2.bias"])
    se[1] = nn.SiLU()
    if cin != cout:
        assert isinstance(blk.shortcut, regnet.RegNetShortCut)
        blk.shortcut.convolution.weight.data.copy_(sd[p + "downsample.conv.weight"])
        blk.shortcut.normalization = LayerNorm2d(sd[p + "downsample.bn.weight"], sd[p + "downsample.bn.bias"], eps)
    else:
        assert isinstance(blk.shortcut, nn.Identity) and (p + "downsample.conv.weight") not in sd
    return blk


@pytest.mark.parametrize("cin,cout,eps", [(24, 40, 1e-5), (40, 40, 1e-5), (30, 30, 1e-6), (18, 50, 1e-6)])
def test_regstage_block_equals_hf_regnet_y_layer(cin, cout, eps):
    g = torch.Generator().manual_seed(cin * 100 + cout)
    sd = {}
    O.make_regstage_weights(sd, g, "s.", 1, cin, cout, 0.1)
    x = torch.randn(3, cin, 6, 6, generator=g)
    with torch.no_grad():
        y_hf = hf_block(sd, "s.b1.", cin, cout, eps)(x.clone())
        y_or = O.regstage_block(sd, "s.b1.", x, eps)
    assert y_hf.shape == y_or.shape == (3, cout, 6, 6)
    assert torch.allclose(y_hf, y_or, rtol=1e-5, atol=1e-6), (y_hf - y_or).abs().max()


def test_regstage_chain_equals_hf_stage_of_four():
    """depth 4 as STC-v35 builds it (the first block widens 1152 -> 3584 with a projected shortcut, the other three keep the width)."""
    g = torch.Generator().manual_seed(5)
    sd = {}
    O.make_regstage_weights(sd, g, "s1.", 4, 16, 48, 0.1)
    x = torch.randn(2, 16, 4, 4, generator=g)
    with torch.no_grad():
        y = x.clone()
        for i in range(4):
            y = hf_block(sd, f"s1.b{i + 1}.", 16 if i == 0 else 48, 48, 1e-5)(y)
        y_or = O.regstage(sd, "s1.", x, 4, 1e-5)
    assert torch.allclose(y, y_or, rtol=1e-5, atol=1e-6)
