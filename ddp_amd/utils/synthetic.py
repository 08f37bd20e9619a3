"""Deterministic synthetic weights / inputs for the DDP hot path.

No checkpoints or datasets are reachable offline, so tests, ``bench.py`` and the golden-vector
generator all draw the *same* seeded tensors from here.  Key names and shapes follow the
reference ``state_dict`` layout (probe of
``segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py``) so that a dict produced here
loads into the reference model with ``load_state_dict`` and into ``ddp_amd``'s drop-in classes
unchanged.

The reference's own initialisation is degenerate for testing
(``MultiScaleDeformableAttention.init_weights`` zeroes ``sampling_offsets.weight`` and
``attention_weights.*``; controlnet/annotator/uniformer/mmcv/ops/multi_scale_deform_attn.py:230-247),
so every term is given a non-trivial seeded value instead: offsets move a few pixels, attention
logits are O(1), LayerNorm affine is perturbed.
"""
import math

import torch

EMBED = 256
HEADS = 8
POINTS = 4
FFN = 1024
TIME_DIM = 1024
SINU_DIM = 16


def _uniform(gen, shape, bound):
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * bound


def _normal(gen, shape, mean=0.0, std=1.0):
    return torch.randn(shape, generator=gen, dtype=torch.float32) * std + mean


def _xavier(gen, out_f, in_f, *extra):
    fan_in = in_f
    fan_out = out_f
    for e in extra:
        fan_in *= e
        fan_out *= e
    bound = math.sqrt(6.0 / (fan_in + fan_out))
    return _uniform(gen, (out_f, in_f) + tuple(extra), bound)


def _offset_ring_bias():
    """sampling_offsets.bias of the reference init: per-head direction ring of radius 1..4 px
    (multi_scale_deform_attn.py:233-244)."""
    thetas = torch.arange(HEADS, dtype=torch.float32) * (2.0 * math.pi / HEADS)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(HEADS, 1, 1, 2).repeat(1, 1, POINTS, 1)
    for i in range(POINTS):
        grid[:, :, i, :] *= i + 1
    return grid.reshape(-1)


# Weight profiles.  'init': the reference's initialisation plus small perturbations (offsets = the per-head ring +- 0.3 px,
# 0.3 px of content-dependent spread: what every number in DESIGN.md is quoted on).  'trained_like': what training does to
# these parameters, exaggerated - content-dependent offsets of +- 2.4 px on top of a ring perturbed by 1 px (taps far from
# the tile's mean offset: the gather's global-memory path, windows clamped at the map border), peaked attention weights
# and 8x larger class scores (few near-ties).  No checkpoint exists in this image; this profile is a stress test, not data.
PROFILES = {
    'init': dict(off_w=0.02, off_b=0.3, attn_w=0.05, attn_b=1.0, seg_gain=1.0),
    'trained_like': dict(off_w=0.15, off_b=1.0, attn_w=0.3, attn_b=2.0, seg_gain=8.0),
    # every (head, point) sits 4 px (1 sigma) off the reference's ring: the points of a head spread +- 6 px around their mean, so
    # every 8-token group of the LDS-staged gather has taps outside its +- 3 px window and many taps leave the map.  The spread is
    # in the BIAS (content-independent): with +- 6 px of CONTENT-dependent offsets (off_w = 0.375) the loop is chaotic - the fp32
    # reference and its own fp64 evaluation take 1004 of 2880 different argmax decisions on a 24 x 40 map - and pins nothing
    'wide_offsets': dict(off_w=0.02, off_b=4.0, attn_w=0.05, attn_b=1.0, seg_gain=1.0),
}


def encoder_layer_state(gen, prefix, sd, profile='init'):
    pr = PROFILES[profile]
    a = prefix + 'attentions.0.'
    sd[a + 'sampling_offsets.weight'] = _normal(gen, (HEADS * POINTS * 2, EMBED), std=pr['off_w'])
    sd[a + 'sampling_offsets.bias'] = _offset_ring_bias() + _normal(gen, (HEADS * POINTS * 2,), std=pr['off_b'])
    sd[a + 'attention_weights.weight'] = _normal(gen, (HEADS * POINTS, EMBED), std=pr['attn_w'])
    sd[a + 'attention_weights.bias'] = _normal(gen, (HEADS * POINTS,), std=pr['attn_b'])
    sd[a + 'value_proj.weight'] = _xavier(gen, EMBED, EMBED)
    sd[a + 'value_proj.bias'] = _normal(gen, (EMBED,), std=0.02)
    sd[a + 'output_proj.weight'] = _xavier(gen, EMBED, EMBED)
    sd[a + 'output_proj.bias'] = _normal(gen, (EMBED,), std=0.02)
    sd[prefix + 'time_mlp.1.weight'] = _uniform(gen, (2 * EMBED, TIME_DIM), 1.0 / math.sqrt(TIME_DIM))
    sd[prefix + 'time_mlp.1.bias'] = _uniform(gen, (2 * EMBED,), 1.0 / math.sqrt(TIME_DIM))
    sd[prefix + 'ffns.0.layers.0.0.weight'] = _xavier(gen, FFN, EMBED)
    sd[prefix + 'ffns.0.layers.0.0.bias'] = _normal(gen, (FFN,), std=0.02)
    sd[prefix + 'ffns.0.layers.1.weight'] = _xavier(gen, EMBED, FFN)
    sd[prefix + 'ffns.0.layers.1.bias'] = _normal(gen, (EMBED,), std=0.02)
    for n in (0, 1):
        sd[prefix + f'norms.{n}.weight'] = _normal(gen, (EMBED,), mean=1.0, std=0.1)
        sd[prefix + f'norms.{n}.bias'] = _normal(gen, (EMBED,), std=0.1)


def time_mlp_state(gen, sd):
    sd['time_mlp.0.weights'] = _normal(gen, (SINU_DIM // 2,))
    sd['time_mlp.1.weight'] = _uniform(gen, (TIME_DIM, SINU_DIM + 1), 1.0 / math.sqrt(SINU_DIM + 1))
    sd['time_mlp.1.bias'] = _uniform(gen, (TIME_DIM,), 1.0 / math.sqrt(SINU_DIM + 1))
    sd['time_mlp.3.weight'] = _uniform(gen, (TIME_DIM, TIME_DIM), 1.0 / math.sqrt(TIME_DIM))
    sd['time_mlp.3.bias'] = _uniform(gen, (TIME_DIM,), 1.0 / math.sqrt(TIME_DIM))


def make_state_dict(task='seg', num_classes=150, num_layers=6, feat_channels=256, seed=2, profile='init'):
    """Hot-path ``state_dict`` (CPU fp32) for ``task`` in {'seg', 'depth', 'bev'}.

    seg  : segmentation/mmseg/models/segmentors/ddp.py:78,92-112 + decode head
    depth: depth/depth/models/depther/ddp.py:70-91 (``down`` conv over 256+1 channels, 3x3 ``conv_depth``)
    bev  : bev/mmdet3d/models/fusion_models/ddp.py:91,104-114 (``transform`` over 256+feat_channels,
           embedding (7,256)); head keys are given the ``decode_head.`` prefix here as well.
    """
    gen = torch.Generator(device='cpu')
    gen.manual_seed(int(seed))
    sd = {}
    if task == 'depth':
        cin = feat_channels + 1
        sd['down.conv.weight'] = _xavier(gen, EMBED, cin, 1, 1)
        sd['down.conv.bias'] = _normal(gen, (EMBED,), std=0.02)
    else:
        cin = feat_channels + EMBED
        sd['transform.conv.weight'] = _xavier(gen, EMBED, cin, 1, 1)
        sd['transform.conv.bias'] = _normal(gen, (EMBED,), std=0.02)
    time_mlp_state(gen, sd)
    if task != 'depth':
        sd['embedding_table.weight'] = _normal(gen, (num_classes + 1, EMBED))
    for l in range(num_layers):
        encoder_layer_state(gen, f'decode_head.encoder.layers.{l}.', sd, profile)
    if task == 'depth':
        sd['decode_head.conv_depth.weight'] = _xavier(gen, 1, EMBED, 3, 3) * 4.0
        sd['decode_head.conv_depth.bias'] = _normal(gen, (1,), mean=2.0, std=0.1)
    else:
        sd['decode_head.conv_seg.weight'] = _xavier(gen, num_classes, EMBED, 1, 1) * PROFILES[profile]['seg_gain']
        sd['decode_head.conv_seg.bias'] = _normal(gen, (num_classes,), std=0.02)
    return sd


def make_inputs(batch, h, w, randsteps=1, feat_channels=256, noise_channels=256, seed=0):
    """Synthetic frozen feature ``x`` (B,Cx,h,w) ~ N(0,1) and start noise (B,r,Cm,h,w) ~ N(0,1)
    (FPN+GN output is roughly unit scale)."""
    gx = torch.Generator(device='cpu')
    gx.manual_seed(int(seed))
    x = torch.randn((batch, feat_channels, h, w), generator=gx, dtype=torch.float32)
    gn = torch.Generator(device='cpu')
    gn.manual_seed(int(seed) + 1)
    noise = torch.randn((batch, randsteps, noise_channels, h, w), generator=gn, dtype=torch.float32)
    return x, noise


def checksum(sd):
    """Order-independent fingerprint of a state dict (stored in golden fixtures to detect drift of
    the seeded generator across torch versions)."""
    tot = 0.0
    for k in sorted(sd):
        v = sd[k].double()
        tot += float(v.sum()) + 0.5 * float(v.abs().sum())
    return tot


def make_scores(batch, num_classes, h, w, seed):
    """Seeded low-resolution class scores (B,K,h,w) for the post-loop epilogue tests."""
    g = torch.Generator().manual_seed(10_000 + seed)
    return torch.randn((batch, num_classes, h, w), generator=g) * 3.0


def make_depth_map(batch, h, w, seed):
    """Seeded low-resolution metric depth (B,1,h,w) for the depth epilogue tests: smooth + noise, with a share of the values
    outside [1e-3, 80] on both sides so the clamp of encode_decode (depth/depth/models/depther/ddp.py:101) does something."""
    g = torch.Generator().manual_seed(15_000 + seed)
    yy = torch.linspace(0, 1, h).view(1, 1, h, 1)
    xx = torch.linspace(0, 1, w).view(1, 1, 1, w)
    base = -10.0 + 105.0 * (0.5 * xx + 0.5 * yy)            # -10 .. 95: ~10 % below min_depth, ~14 % above 80
    return base + 6.0 * torch.randn((batch, 1, h, w), generator=g)


def make_neck_state_dict(seed):
    """MultiStageMerging parameters with the reference's key names (necks/multi_stage_merging.py:28-37)."""
    g = torch.Generator().manual_seed(20_000 + seed)
    return {
        'down.conv.weight': torch.randn((256, 1024, 1, 1), generator=g) * 0.03,
        'down.gn.weight': 1.0 + 0.1 * torch.randn((256,), generator=g),
        'down.gn.bias': 0.1 * torch.randn((256,), generator=g),
    }


def make_levels(batch, h, w, seed):
    """Four seeded FPN-like levels (B,256,ceil(h/2^l),ceil(w/2^l)), l = 0..3."""
    g = torch.Generator().manual_seed(30_000 + seed)
    out = []
    for l in range(4):
        out.append(torch.randn((batch, 256, -(-h // (1 << l)), -(-w // (1 << l))), generator=g))
    return out


def make_fcn_state_dict(num_convs, num_classes, with_norm, concat_input, seed):
    """FCNHeadWithTime parameters / buffers with the reference's key names (decode_heads/fcn_head_with_time.py)."""
    g = torch.Generator().manual_seed(40_000 + seed)
    sd = {}

    def bn(prefix):
        sd[prefix + 'bn.weight'] = 1.0 + 0.2 * torch.randn((256,), generator=g)
        sd[prefix + 'bn.bias'] = 0.1 * torch.randn((256,), generator=g)
        sd[prefix + 'bn.running_mean'] = 0.2 * torch.randn((256,), generator=g)
        sd[prefix + 'bn.running_var'] = 0.5 + torch.rand((256,), generator=g)
        sd[prefix + 'bn.num_batches_tracked'] = torch.tensor(7)
    for i in range(num_convs):
        p = f'convs.{i}.'
        sd[p + 'conv.weight'] = torch.randn((256, 256, 3, 3), generator=g) * 0.03
        if with_norm:
            bn(p)
        else:
            sd[p + 'conv.bias'] = 0.1 * torch.randn((256,), generator=g)
        sd[p + 'time_mlp.1.weight'] = torch.randn((512, 1024), generator=g) * 0.02
        sd[p + 'time_mlp.1.bias'] = 0.05 * torch.randn((512,), generator=g)
    if concat_input:
        sd['conv_cat.conv.weight'] = torch.randn((256, 512, 3, 3), generator=g) * 0.02
        if with_norm:
            bn('conv_cat.')
        else:
            sd['conv_cat.conv.bias'] = 0.1 * torch.randn((256,), generator=g)
    sd['conv_seg.weight'] = torch.randn((num_classes, 256, 1, 1), generator=g) * 0.05
    sd['conv_seg.bias'] = 0.1 * torch.randn((num_classes,), generator=g)
    return sd


def make_fcn_segmentor_state_dict(num_convs, num_classes, with_norm, concat_input, seed):
    """Hot-path state_dict of ``DDP(decode_head=FCNHeadWithTime)``: the segmentor's own parameters
    (transform, time_mlp, embedding_table) + the FCN head's under ``decode_head.``."""
    sd = {k: v for k, v in make_state_dict('seg', num_classes, 0, 256, seed=seed).items() if not k.startswith('decode_head.')}
    sd.update({'decode_head.' + k: v for k, v in make_fcn_state_dict(num_convs, num_classes, with_norm, concat_input, seed).items()})
    return sd


def make_fcn_inputs(maps, h, w, seed):
    g = torch.Generator().manual_seed(50_000 + seed)
    return torch.randn((maps, 256, h, w), generator=g), torch.randn((1, 1024), generator=g)


def make_fpn_state_dict(in_channels, seed):
    """FPN parameters with the reference's key names (necks/fpn.py:119-134)."""
    g = torch.Generator().manual_seed(60_000 + seed)
    sd = {}
    for l, c in enumerate(in_channels):
        sd[f'lateral_convs.{l}.conv.weight'] = torch.randn((256, c, 1, 1), generator=g) * (1.0 / c ** 0.5)
        sd[f'lateral_convs.{l}.gn.weight'] = 1.0 + 0.1 * torch.randn((256,), generator=g)
        sd[f'lateral_convs.{l}.gn.bias'] = 0.1 * torch.randn((256,), generator=g)
        sd[f'fpn_convs.{l}.conv.weight'] = torch.randn((256, 256, 3, 3), generator=g) * 0.03
        sd[f'fpn_convs.{l}.gn.weight'] = 1.0 + 0.1 * torch.randn((256,), generator=g)
        sd[f'fpn_convs.{l}.gn.bias'] = 0.1 * torch.randn((256,), generator=g)
    return sd


def make_backbone_levels(batch, in_channels, h, w, seed):
    """Four seeded backbone-like levels (B,C_l,ceil(h/2^l),ceil(w/2^l))."""
    g = torch.Generator().manual_seed(70_000 + seed)
    return [torch.randn((batch, c, -(-h // (1 << l)), -(-w // (1 << l))), generator=g) for l, c in enumerate(in_channels)]
