"""Drop-in depth ``DDP`` + ``DeformableHeadWithTime`` (depth/depth/models/depther/ddp.py:34-247;
depth/depth/models/decode_heads/deformable_head_with_time.py:20-169): ``down`` concat-conv over
256+1 channels, raw-t time embedding, 3x3 ``conv_depth`` regression head, cosine-gamma DDIM step."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..decode_heads.deformable_head_with_time import DeformableHeadWithTime as _SegHead
from ..registry import DEPTHER, HEADS, build_backbone, build_head
from ..segmentors.ddp import LearnedSinusoidalPosEmb, _Conv1x1, _SamplerMixin, _build_neck
from .. import schedule


@HEADS.register_module(name='DepthDeformableHeadWithTime')
class DepthDeformableHeadWithTime(_SegHead):
    task = 'depth'
    head_conv = 'conv_depth'

    def __init__(self, min_depth=1e-3, max_depth=None, scale_up=False, classify=False, use_eps=True, n_bins=None,
                 init_inputs=False, **kwargs):
        if scale_up or classify or not use_eps:
            raise ValueError('only the regression branch used by the DDP configs is implemented '
                             '(scale_up=False, classify=False, use_eps=True; decode_head.py:264-269)')
        self.min_depth, self.max_depth = min_depth, max_depth
        self.scale_up, self.classify, self.use_eps = scale_up, classify, use_eps
        kwargs.setdefault('num_classes', 1)
        super().__init__(**kwargs)

    def _make_head_conv(self):
        self.conv_depth = nn.Conv2d(self.channels, 1, kernel_size=3, padding=1, stride=1)

    def _engine_kwargs(self):
        return dict(min_depth=self.min_depth, max_depth=self.max_depth if self.max_depth is not None else 80.0)


@DEPTHER.register_module(name='DepthDDP')
class DDP(nn.Module, _SamplerMixin):
    task = 'depth'

    def __init__(self, bit_scale=1, bits=8, timesteps=1, randsteps=1, time_difference=1, learned_sinusoidal_dim=16,
                 sample_range=(0, 0.999), ddim=True, rule=None, min_depth=1e-3, max_depth=80, backbone=None,
                 neck=None, decode_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        if not ddim:
            raise NotImplementedError('the reference references ddpm_step but never defines it (depther/ddp.py:244)')
        self.backbone = build_backbone(backbone) if backbone is not None else None
        self.neck = _build_neck(neck)
        if isinstance(decode_head, dict) and decode_head.get('type') == 'DeformableHeadWithTime':
            decode_head = dict(decode_head, type='DepthDeformableHeadWithTime')
        self.decode_head = build_head(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.bit_scale, self.BITS, self.timesteps, self.randsteps = bit_scale, bits, timesteps, randsteps
        self.time_difference, self.sample_range, self.ddim = time_difference, sample_range, ddim
        self.min_depth, self.max_depth = min_depth, max_depth
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        c = self.decode_head.in_channels[0]
        self.down = _Conv1x1(c + 1, c)
        self.time_mlp = nn.Sequential(LearnedSinusoidalPosEmb(learned_sinusoidal_dim),
                                      nn.Linear(learned_sinusoidal_dim + 1, c * 4), nn.GELU(), nn.Linear(c * 4, c * 4))

    def extract_feat(self, img):
        x = self.backbone(img)
        if self.neck is not None:
            x = self.neck(x)
        return x

    def hot_path_state_dict(self):
        return {k: v for k, v in self.state_dict().items() if not k.startswith(('backbone.', 'neck.'))}

    def _get_sampling_timesteps(self, batch, *, device):
        return [torch.tensor([a, b], device=device)[:, None].repeat(1, batch)
                for a, b in schedule.get_sampling_timesteps(self.timesteps, self.time_difference, 0.0)]

    @torch.no_grad()
    def sample(self, x, img_metas=None, noise=None):
        if not x.is_cuda:
            raise RuntimeError('ddp_amd has no CPU path: features must live on an MI355X (HIP) device')
        b, c, h, w = x.shape
        if noise is None:
            noise = torch.randn((b, self.randsteps, 1, h, w), device=x.device)

        def factory():
            from ..engine import DDPEngine
            return DDPEngine(self.hot_path_state_dict(), 'depth', h=h, w=w, batch=b, randsteps=self.randsteps,
                             timesteps=self.timesteps, bit_scale=self.bit_scale, time_difference=self.time_difference,
                             min_depth=self.min_depth, max_depth=self.max_depth, device=x.device)
        # keyed without the geometry: a new (b, h, w) re-uses the engine through set_geometry (no weight repacking)
        eng = self._get_engine(('depth', str(x.device), self.timesteps, self.randsteps, self.bit_scale, self.time_difference,
                                self.min_depth, self.max_depth), factory, geometry=(b, h, w))
        return eng.sample(x.contiguous().float(), noise.contiguous().float())

    def _decode_head_forward_test(self, x, t, img_metas=None):
        return self.decode_head.forward_test(x, t, img_metas, self.test_cfg)

    def encode_decode(self, img, img_metas=None, rescale=False):
        """depther/ddp.py:95-109."""
        x = self.extract_feat(img)[0]
        out = self.sample(x, img_metas)
        out = torch.clamp(out, min=self.decode_head.min_depth, max=self.decode_head.max_depth)
        if rescale:
            out = F.interpolate(out, size=img.shape[2:], mode='bilinear', align_corners=self.align_corners)
        return out

    def forward_train(self, *a, **k):
        raise NotImplementedError('training is out of scope of ddp_amd (SURVEY.md §8)')
