"""Oracle (CPU, test infrastructure only): the Mix Transformer encoder, restated functionally on a state dict.

Follows /root/reference/domain_adaptation/GTA5/model/networks/MixTransfomer.py:
  OverlapPatchEmbed.forward :221-228, Block.forward :175-179, Attention.forward :120-139, Mlp.forward :71-79,
  DWConv.forward :415-421, MixVisionTransformer.forward_features :370-407, mit_b1..b5 :435-475.
Plain torch ops in the dtype of the state dict (float32, or float64 for tight kernel checks).  DropPath (timm, stochastic
depth) is the identity here: the goldens are captured in eval() mode, where it is; dropout rates are 0 in every mit_b*.
Pinned by tests/golden/mit.npz, a capture of the REFERENCE module (tools/gen_golden.py::gen_mit imports it with stand-ins
for timm's / mmcv's non-arithmetic helpers only).  LayerNorm epsilons: 1e-6 for Block.norm1/norm2 and the stage norms
(`norm_layer=partial(nn.LayerNorm, eps=1e-6)`, :438), torch's default 1e-5 for OverlapPatchEmbed.norm (:202) and
Attention.norm (:106).
"""
import dataclasses
import zlib

import torch
import torch.nn.functional as F


@dataclasses.dataclass(frozen=True)
class MitArch:
    embed_dims: tuple = (64, 128, 320, 512)
    num_heads: tuple = (1, 2, 5, 8)
    depths: tuple = (3, 6, 40, 3)
    sr_ratios: tuple = (8, 4, 2, 1)
    mlp_ratio: int = 4
    in_chans: int = 3


MIT_B1 = MitArch(depths=(2, 2, 2, 2))
MIT_B2 = MitArch(depths=(3, 4, 6, 3))
MIT_B5 = MitArch()
MIT_TINY = MitArch(embed_dims=(64, 128), num_heads=(1, 2), depths=(1, 1), sr_ratios=(2, 1))     # two stages: kernel-level tests


def state_shapes(arch=MIT_B5):
    """Key -> (shape, kind) in the reference's state-dict order (MixTransfomer.py:245-287)."""
    out = {}
    cin = arch.in_chans
    nst = len(arch.embed_dims)
    for s in range(nst):
        c = arch.embed_dims[s]
        k = 7 if s == 0 else 3
        p = f"patch_embed{s + 1}"
        out[p + ".proj.weight"] = ((c, cin, k, k), "conv")
        out[p + ".proj.bias"] = ((c,), "bias")
        out[p + ".norm.weight"] = ((c,), "ln_w")
        out[p + ".norm.bias"] = ((c,), "bias")
        cin = c
    for s in range(nst):
        c, sr, hid = arch.embed_dims[s], arch.sr_ratios[s], arch.embed_dims[s] * arch.mlp_ratio
        for i in range(arch.depths[s]):
            b = f"block{s + 1}.{i}"
            out[b + ".norm1.weight"] = ((c,), "ln_w")
            out[b + ".norm1.bias"] = ((c,), "bias")
            out[b + ".attn.q.weight"] = ((c, c), "lin")
            out[b + ".attn.q.bias"] = ((c,), "bias")
            out[b + ".attn.kv.weight"] = ((2 * c, c), "lin")
            out[b + ".attn.kv.bias"] = ((2 * c,), "bias")
            out[b + ".attn.proj.weight"] = ((c, c), "lin_out")
            out[b + ".attn.proj.bias"] = ((c,), "bias")
            if sr > 1:
                out[b + ".attn.sr.weight"] = ((c, c, sr, sr), "conv")
                out[b + ".attn.sr.bias"] = ((c,), "bias")
                out[b + ".attn.norm.weight"] = ((c,), "ln_w")
                out[b + ".attn.norm.bias"] = ((c,), "bias")
            out[b + ".norm2.weight"] = ((c,), "ln_w")
            out[b + ".norm2.bias"] = ((c,), "bias")
            out[b + ".mlp.fc1.weight"] = ((hid, c), "lin")
            out[b + ".mlp.fc1.bias"] = ((hid,), "bias")
            out[b + ".mlp.dwconv.dwconv.weight"] = ((hid, 1, 3, 3), "dw")
            out[b + ".mlp.dwconv.dwconv.bias"] = ((hid,), "bias")
            out[b + ".mlp.fc2.weight"] = ((c, hid), "lin_out")
            out[b + ".mlp.fc2.bias"] = ((c,), "bias")
        out[f"norm{s + 1}.weight"] = ((c,), "ln_w")
        out[f"norm{s + 1}.bias"] = ((c,), "bias")
    # the reference registers block1, norm1, block2, norm2, ... in this order: rebuild it
    order = [k for k in out if k.startswith("patch_embed")]
    for s in range(nst):
        order += [k for k in out if k.startswith(f"block{s + 1}.")] + [f"norm{s + 1}.weight", f"norm{s + 1}.bias"]
    return {k: out[k] for k in order}


def fill(name, shape, kind):
    """Deterministic, name-hashed test weights (values depend on (name, shape, kind) only; cf. oracle/detweights.py).
    Branch-output layers (proj, fc2) are damped like in a trained network so that 52 residual blocks stay well conditioned."""
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(("mit." + name).encode()) & 0x7FFFFFFF)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if kind == "conv":
        return r * (1.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
    if kind == "lin":
        return r * (1.0 / shape[1]) ** 0.5
    if kind == "lin_out":
        return r * 0.3 * (1.0 / shape[1]) ** 0.5
    if kind == "dw":
        return r * 0.3 + (torch.tensor([0, 0, 0, 0, 1.0, 0, 0, 0, 0]).reshape(1, 1, 3, 3))
    if kind == "ln_w":
        return 1.0 + 0.1 * r
    if kind == "bias":
        return 0.05 * r
    raise ValueError(kind)


def state_dict(arch=MIT_B5):
    return {k: fill(k, shp, kind) for k, (shp, kind) in state_shapes(arch).items()}


def attention(sd, p, x, H, W, heads, sr):
    """Attention.forward, MixTransfomer.py:120-139."""
    B, N, C = x.shape
    d = C // heads
    q = F.linear(x, sd[p + ".q.weight"], sd.get(p + ".q.bias")).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = x.permute(0, 2, 1).reshape(B, C, H, W)
        x_ = F.conv2d(x_, sd[p + ".sr.weight"], sd[p + ".sr.bias"], stride=sr).reshape(B, C, -1).permute(0, 2, 1)
        x_ = F.layer_norm(x_, (C,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
    else:
        x_ = x
    kv = F.linear(x_, sd[p + ".kv.weight"], sd.get(p + ".kv.bias")).reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def mlp(sd, p, x, H, W):
    """Mlp.forward :71-79 with DWConv.forward :415-421."""
    B, N, _ = x.shape
    x = F.linear(x, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])
    hid = x.shape[-1]
    x = x.transpose(1, 2).reshape(B, hid, H, W)
    x = F.conv2d(x, sd[p + ".dwconv.dwconv.weight"], sd[p + ".dwconv.dwconv.bias"], 1, 1, 1, hid)
    x = x.flatten(2).transpose(1, 2)
    x = F.gelu(x)
    return F.linear(x, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])


def block(sd, p, x, H, W, heads, sr):
    """Block.forward :175-179 (drop_path = identity)."""
    C = x.shape[-1]
    x = x + attention(sd, p + ".attn", F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6), H, W, heads, sr)
    x = x + mlp(sd, p + ".mlp", F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6), H, W)
    return x


def forward(sd, x, arch=MIT_B5):
    """MixVisionTransformer.forward_features :370-407 -> [c1, .., c4] (NCHW)."""
    outs = []
    B = x.shape[0]
    for s in range(len(arch.embed_dims)):
        p = f"patch_embed{s + 1}"
        k = 7 if s == 0 else 3
        x = F.conv2d(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"], stride=4 if s == 0 else 2, padding=k // 2)   # :221-222
        _, C, H, W = x.shape
        x = x.flatten(2).transpose(1, 2)
        x = F.layer_norm(x, (C,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
        for i in range(arch.depths[s]):
            x = block(sd, f"block{s + 1}.{i}", x, H, W, arch.num_heads[s], arch.sr_ratios[s])
        x = F.layer_norm(x, (C,), sd[f"norm{s + 1}.weight"], sd[f"norm{s + 1}.bias"], 1e-6)
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs
