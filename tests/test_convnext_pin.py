"""CLIP-ConvNeXt trunk (BASELINE configs[3]; clip_convnext_encoder.py:150-174 drives timm's ConvNeXt.stem / .stages / .norm_pre).
timm / open_clip are NOT installed here and there is no network.  Two pins:
 (1) tests/golden/convnext.npz (oracle/gen_golden.py run_convnext): the REFERENCE's own `CLIPConvNextVisionTower._forward` run on a stand-in
     trunk made of transformers.ConvNextModel's sub-modules — a third-party implementation of the published block, independent of timm and of
     this repo — with the closed-form weights loaded through a timm-name -> HF-name map.  The oracle must reproduce its output
     (test_oracle_matches_reference_forward_on_hf_convnext).  This is the row's pin.
 (2) the oracle against an assembly of stock torch.nn modules wired per the public block definition and loaded through timm's state-dict
     names, piece by piece — kept from round 2.
What stays unverifiable here: that timm's own modules compute what HF's and the paper's do (same published architecture)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import visper_oracle as O, weights as WT

P = "model.vision_tower.vision_tower."


class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel axis of an NCHW map (timm.layers.LayerNorm2d: permute -> F.layer_norm -> permute)."""

    def forward(self, x):
        return super().forward(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


class Block(nn.Module):
    """timm ConvNeXtBlock (conv_mlp=False): dwconv7x7 -> NHWC -> LayerNorm -> fc1 -> GELU -> fc2 -> * gamma -> NCHW -> + shortcut."""

    def __init__(self, C, eps):
        super().__init__()
        self.conv_dw = nn.Conv2d(C, C, 7, padding=3, groups=C)
        self.norm = nn.LayerNorm(C, eps=eps)
        self.mlp = nn.ModuleDict(dict(fc1=nn.Linear(C, 4 * C), fc2=nn.Linear(4 * C, C)))
        self.act = nn.GELU()
        self.gamma = nn.Parameter(torch.ones(C))

    def forward(self, x):
        y = self.conv_dw(x).permute(0, 2, 3, 1)
        y = self.mlp["fc2"](self.act(self.mlp["fc1"](self.norm(y))))
        return x + (y * self.gamma).permute(0, 3, 1, 2)


class Stage(nn.Module):
    def __init__(self, Cin, C, depth, eps, first):
        super().__init__()
        self.downsample = nn.Identity() if first else nn.Sequential(LayerNorm2d(Cin, eps=eps), nn.Conv2d(Cin, C, 2, stride=2))
        self.blocks = nn.Sequential(*[Block(C, eps) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class Trunk(nn.Module):
    def __init__(self, dims, depths, eps):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, dims[0], 4, stride=4), LayerNorm2d(dims[0], eps=eps))
        self.stages = nn.Sequential(*[Stage(dims[max(i - 1, 0)], dims[i], depths[i], eps, i == 0) for i in range(len(dims))])
        self.norm_pre = nn.Identity()

    def forward(self, images):                       # clip_convnext_encoder.py:161-173
        x = self.norm_pre(self.stages(self.stem(images)))
        return x.flatten(2, 3).permute(0, 2, 1).contiguous()


def _case(dims=(16, 32, 48, 64), depths=(1, 2, 2, 1), px=64):
    cfg = O.make_config(cnx_dims=dims, cnx_depths=depths)
    m = Trunk(dims, depths, cfg.cnx_eps)
    W = {}
    for k, v in m.state_dict().items():
        amp = 0.5 if k.endswith("gamma") else None
        W[P + k] = WT.tensor("cnx." + k, v.shape, amp) if amp else WT.param(P + k, v.shape)
    m.load_state_dict({k[len(P):]: v for k, v in W.items()})
    images = WT.tensor("cnx_pin_images", (2, 3, px, px))
    return cfg, m.eval(), W, images


def test_oracle_matches_reference_forward_on_hf_convnext():
    """Pin (1): oracle.convnext_features == the reference's _forward over transformers.ConvNextModel's modules (golden)."""
    import json
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "convnext.npz"))
    dims, depths, px = tuple(int(x) for x in g["dims"]), tuple(int(x) for x in g["depths"]), int(g["px"])
    cfg = O.make_config(cnx_dims=dims, cnx_depths=depths, cnx_eps=float(g["eps"]))
    name_map = json.loads(str(g["name_map"]))
    W = {}
    for k in name_map:                                            # timm names: the keys of this repo's checkpoints
        shp = _timm_shape(k, dims)
        W[P + k] = WT.tensor(P + k, shp, 0.5) if k.endswith("gamma") else WT.param(P + k, shp)
    images = WT.tensor("cnx_pin_images", (2, 3, px, px))
    got = O.convnext_features(images, W, cfg)
    ref = torch.from_numpy(g["features"])
    assert got.shape == ref.shape == (2, (px // 32) ** 2, dims[-1])
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err
    # and the name map covers exactly the parameter set this repo declares for the tower
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.params import param_shapes
    vc = VisperConfig(mm_vision_tower="CLIP-convnext-pin", cnx_dims=dims, cnx_depths=depths)
    assert {k[len(P):] for k in param_shapes(vc) if k.startswith(P)} == set(name_map)


def _timm_shape(k, dims):
    parts = k.split(".")
    if parts[0] == "stem":
        return (dims[0], 3, 4, 4) if parts[1] == "0" and parts[2] == "weight" else (dims[0],)
    i = int(parts[1])
    C = dims[i]
    if parts[2] == "downsample":
        if parts[3] == "0":
            return (dims[i - 1],)
        return (C, dims[i - 1], 2, 2) if parts[4] == "weight" else (C,)
    rest = ".".join(parts[4:])
    return {"gamma": (C,), "conv_dw.weight": (C, 1, 7, 7), "conv_dw.bias": (C,), "norm.weight": (C,), "norm.bias": (C,),
            "mlp.fc1.weight": (4 * C, C), "mlp.fc1.bias": (4 * C,), "mlp.fc2.weight": (C, 4 * C), "mlp.fc2.bias": (C,)}[rest]


def test_state_dict_names_are_timm_convnext_names():
    from visper_lm_amd.config import VisperConfig
    from visper_lm_amd.params import param_shapes
    cfg, m, W, _ = _case()
    vc = VisperConfig(mm_vision_tower="CLIP-convnext-test", cnx_dims=(16, 32, 48, 64), cnx_depths=(1, 2, 2, 1))
    mine = {k: tuple(s) for k, s in param_shapes(vc).items() if k.startswith(P)}
    assert mine == {k: tuple(v.shape) for k, v in W.items()}


def test_oracle_trunk_equals_torch_nn_assembly_end_to_end():
    cfg, m, W, images = _case()
    with torch.no_grad():
        ref = m(images)
        got = O.convnext_features(images, W, cfg)
    assert got.shape == ref.shape == (2, 4, 64)
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_pieces_layernorm2d_dwconv_layerscale_downsample():
    """The four operations the HIP path re-implements (vp_layernorm on NHWC rows, vp_dwconv7x7_nhwc, gamma folded into fc2, the 2x2/s2
    patch gather + GEMM) in the oracle's arithmetic vs the stock modules, on one feature map."""
    torch.manual_seed(0)
    C, H = 24, 10
    x = WT.tensor("cnx_piece_x", (2, C, H, H))
    # LayerNorm2d
    ln = LayerNorm2d(C, eps=1e-5)
    ln.weight.data, ln.bias.data = WT.tensor("p.lnw", (C,), 0.1, 1.0), WT.tensor("p.lnb", (C,), 0.02)
    mine = F.layer_norm(x.permute(0, 2, 3, 1).reshape(-1, C), (C,), ln.weight, ln.bias, 1e-5).view(2, H, H, C).permute(0, 3, 1, 2)
    assert torch.allclose(mine, ln(x), atol=1e-6)
    # depthwise 7x7 as 49 shifted multiply-adds over an NHWC map with tap-major weights (the layout vp_dwconv7x7_nhwc takes)
    dw = nn.Conv2d(C, C, 7, padding=3, groups=C)
    dw.weight.data, dw.bias.data = WT.tensor("p.dww", (C, 1, 7, 7), 0.1), WT.tensor("p.dwb", (C,), 0.02)
    taps = dw.weight.data.reshape(C, 49).t()                          # [49, C]
    xp = F.pad(x.permute(0, 2, 3, 1), (0, 0, 3, 3, 3, 3))             # NHWC, zero pad 3
    acc = dw.bias.data.expand(2, H, H, C).clone()
    for ky in range(7):
        for kx in range(7):
            acc = acc + xp[:, ky:ky + H, kx:kx + H, :] * taps[ky * 7 + kx]
    assert torch.allclose(acc.permute(0, 3, 1, 2), dw(x), atol=1e-5)
    # layer scale folded into fc2 (what the frozen tower's weights are loaded as): (fc2(h) * gamma) == F.linear(h, W*gamma[:,None], b*gamma)
    fc2 = nn.Linear(4 * C, C)
    gamma = WT.tensor("p.gamma", (C,), 0.5)
    h = WT.tensor("p.h", (7, 4 * C))
    assert torch.allclose(fc2(h) * gamma, F.linear(h, fc2.weight * gamma[:, None], fc2.bias * gamma), atol=1e-6)
    # 2x2 / stride-2 conv as a patch gather (dy, dx, cin) + GEMM with the weight permuted to [Cout, (dy, dx, cin)]
    ds = nn.Conv2d(C, 2 * C, 2, stride=2)
    xn = x.permute(0, 2, 3, 1)                                        # NHWC
    patches = xn.reshape(2, H // 2, 2, H // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 4 * C)
    wmat = ds.weight.data.permute(0, 2, 3, 1).reshape(2 * C, 4 * C)
    mine = F.linear(patches, wmat, ds.bias.data).view(2, H // 2, H // 2, 2 * C).permute(0, 3, 1, 2)
    assert torch.allclose(mine, ds(x), atol=1e-5)
