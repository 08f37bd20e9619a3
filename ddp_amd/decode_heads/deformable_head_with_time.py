"""Drop-in ``DeformableHeadWithTime`` (plugin surface #2): same constructor kwargs, parameter names
(hence ``state_dict`` keys) and ``forward(inputs, times)`` contract as
segmentation/mmseg/models/decode_heads/deformable_head_with_time.py:21-189, with the compute done by
libddp_mi355x.so (ddp_head_forward).  The nn.Modules below only HOLD parameters in the reference's
layout; none of their ``forward`` methods is on the product path.
"""
import math

import torch
import torch.nn as nn

from ..registry import HEADS

EMBED, HEADS_N, POINTS, FFN_DIM = 256, 8, 4, 1024


class _MSDAParams(nn.Module):
    """attentions.0.* of one layer (mmcv MultiScaleDeformableAttention parameters and init:
    controlnet/annotator/uniformer/mmcv/ops/multi_scale_deform_attn.py:223-247)."""

    def __init__(self, embed_dims=EMBED, num_heads=HEADS_N, num_levels=1, num_points=POINTS, **_):
        super().__init__()
        if (embed_dims, num_heads, num_levels, num_points) != (EMBED, HEADS_N, 1, POINTS):
            raise ValueError('libddp_mi355x is built for embed_dims=256, num_heads=8, num_levels=1, num_points=4 '
                             '(every shipped DDP config); got '
                             f'{(embed_dims, num_heads, num_levels, num_points)}')
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn.init.zeros_(self.sampling_offsets.weight)
        thetas = torch.arange(HEADS_N, dtype=torch.float32) * (2.0 * math.pi / HEADS_N)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(HEADS_N, 1, 1, 2).repeat(1, 1, POINTS, 1)
        for i in range(POINTS):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        for lin in (self.value_proj, self.output_proj):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)


class _FFNParams(nn.Module):
    """ffns.0.* : layers = Sequential(Sequential(Linear, act, drop), Linear, drop)
    (mmcv cnn/bricks/transformer.py:253-268) -> keys layers.0.0.* and layers.1.*"""

    def __init__(self, embed_dims=EMBED, feedforward_channels=FFN_DIM, act_cfg=None, **_):
        super().__init__()
        if (embed_dims, feedforward_channels) != (EMBED, FFN_DIM):
            raise ValueError('libddp_mi355x is built for FFN 256->1024->256')
        if act_cfg is not None and act_cfg.get('type', 'GELU') != 'GELU':
            raise ValueError('only the GELU FFN of the DDP configs is implemented')
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.GELU(), nn.Identity()),
                                    nn.Linear(feedforward_channels, embed_dims), nn.Identity())


class _TimeAwareLayerParams(nn.Module):
    """One time-aware BaseTransformerLayer (segmentation/mmseg/models/utils/transformer.py:182-315):
    attentions.0, ffns.0, norms.{0,1}, time_mlp = Sequential(SiLU, Linear(1024, 512))."""

    def __init__(self, attn_cfgs=None, ffn_cfgs=None, use_time_mlp=False,
                 operation_order=('self_attn', 'norm', 'ffn', 'norm'), **_):
        super().__init__()
        if tuple(operation_order) != ('self_attn', 'norm', 'ffn', 'norm'):
            raise ValueError("only operation_order ('self_attn','norm','ffn','norm') is implemented")
        attn_cfgs = dict(attn_cfgs or {})
        attn_cfgs.pop('type', None)
        ffn_cfgs = dict(ffn_cfgs or {})
        ffn_cfgs.pop('type', None)
        self.attentions = nn.ModuleList([_MSDAParams(**attn_cfgs)])
        self.ffns = nn.ModuleList([_FFNParams(**ffn_cfgs)])
        self.norms = nn.ModuleList([nn.LayerNorm(EMBED), nn.LayerNorm(EMBED)])
        self.use_time_mlp = use_time_mlp
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(EMBED * 4, EMBED * 2)) if use_time_mlp else None


class _EncoderParams(nn.Module):
    """DetrTransformerEncoder: ``layers`` ModuleList, no final norm (utils/transformer.py:1300-1329)."""

    def __init__(self, num_layers=6, transformerlayers=None, **_):
        super().__init__()
        tl = dict(transformerlayers or {})
        tl.pop('type', None)
        self.layers = nn.ModuleList([_TimeAwareLayerParams(**tl) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.embed_dims = EMBED


@HEADS.register_module()
class DeformableHeadWithTime(nn.Module):
    """forward(inputs: list[Tensor (R,256,h,w)], times: Tensor (1|R,1024)) -> logits (R,K,h,w)."""

    task = 'seg'
    head_conv = 'conv_seg'

    def __init__(self, num_feature_levels=1, encoder=None, positional_encoding=None, in_channels=(256,),
                 channels=256, in_index=(0,), num_classes=150, dropout_ratio=0., norm_cfg=None,
                 align_corners=False, loss_decode=None, **kwargs):
        super().__init__()
        if num_feature_levels != 1:
            raise ValueError('DDP configs use num_feature_levels=1')
        pe = dict(positional_encoding or dict(num_feats=128, normalize=True, offset=-0.5))
        if pe.get('num_feats', 128) * 2 != EMBED or not pe.get('normalize', False) or pe.get('offset', 0.) != -0.5:
            raise ValueError('positional_encoding must be SinePositionalEncoding(num_feats=128, normalize=True, '
                             'offset=-0.5) as in every DDP config')
        enc = dict(encoder or {})
        enc.pop('type', None)
        self.num_feature_levels = num_feature_levels
        self.in_channels = list(in_channels) if isinstance(in_channels, (list, tuple)) else [in_channels]
        self.channels = channels
        self.in_index = in_index
        self.num_classes = num_classes
        self.dropout_ratio = dropout_ratio
        self.align_corners = align_corners
        self.encoder = _EncoderParams(**enc)
        self.embed_dims = EMBED
        self._make_head_conv()
        self.init_weights()
        self._engines = {}
        self._owner = None      # set by the enclosing DDP segmentor: shares its packed weights

    def _make_head_conv(self):
        self.conv_seg = nn.Conv2d(self.channels, self.num_classes, kernel_size=1)

    def init_weights(self):
        """xavier on every matrix, then the MSDA-specific init (reference :53-60)."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for layer in self.encoder.layers:
            layer.attentions[0].init_weights()

    # ------------------------------------------------------------------------------------------
    def _state_for_engine(self):
        sd = {'decode_head.' + k: v for k, v in self.state_dict().items()}
        # the head alone has no transform/time_mlp/embedding: supply inert placeholders
        dev = self.conv_seg.weight.device if hasattr(self, 'conv_seg') else next(self.parameters()).device
        z = lambda *s: torch.zeros(*s, device=dev)
        if self.task == 'depth':
            sd.update({'down.conv.weight': z(256, 257, 1, 1), 'down.conv.bias': z(256)})
        else:
            sd.update({'transform.conv.weight': z(256, 512, 1, 1), 'transform.conv.bias': z(256),
                       'embedding_table.weight': z(self.num_classes + 1, 256)})
        sd.update({'time_mlp.0.weights': z(8), 'time_mlp.1.weight': z(1024, 17), 'time_mlp.1.bias': z(1024),
                   'time_mlp.3.weight': z(1024, 1024), 'time_mlp.3.bias': z(1024)})
        return sd

    def _engine_kwargs(self):
        return dict(num_classes=self.num_classes)

    def _engine(self, R, h, w, device):
        from ..engine import DDPEngine
        key = (R, h, w, str(device), sum(p._version for p in self.parameters()))
        eng = self._engines.get(key)
        if eng is None:
            self._engines.clear()
            eng = DDPEngine(self._state_for_engine(), self.task, h=h, w=w, batch=R, randsteps=1, timesteps=1,
                            device=device, **self._engine_kwargs())
            self._engines[key] = eng
        return eng

    @torch.no_grad()
    def forward(self, inputs, times):
        feat = inputs[-self.num_feature_levels:][0]
        R, c, h, w = feat.shape
        if c != EMBED:
            raise RuntimeError(f'expected {EMBED} input channels, got {c}')
        if not feat.is_cuda:
            raise RuntimeError('ddp_amd has no CPU path: inputs must live on an MI355X (HIP) device')
        if times is not None and times.shape[0] not in (1, R):
            raise RuntimeError(f'times batch {times.shape[0]} does not broadcast to {R}')
        if times is not None and times.shape[0] == R and R > 1 and not bool((times == times[:1]).all()):
            raise RuntimeError('per-sample time embeddings are not supported: the sampler feeds one time per step')
        eng = self._engine(R, h, w, feat.device)
        return eng.head_forward(feat.contiguous().float(), None if times is None else times.float())

    def forward_test(self, inputs, times, img_metas=None, test_cfg=None):
        return self.forward(inputs, times)

    def forward_train(self, *a, **k):
        raise NotImplementedError('ddp_amd implements the inference loop only (SURVEY.md §8: training is out of scope)')
