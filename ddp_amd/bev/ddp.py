"""Drop-in BEV map-segmentation sampler (bev/mmdet3d/models/fusion_models/ddp.py:65-114,268-301) and
head (bev/mmdet3d/models/heads/segm/deformable_head_with_time.py:57-235).  The reference class
derives from BEVFusion (sensor encoders, out of scope); this one keeps only the diffusion members and
``ddim_sample(x: list[Tensor], head)``."""
import torch
import torch.nn as nn

from ..decode_heads.deformable_head_with_time import DeformableHeadWithTime as _SegHead
from ..registry import FUSIONMODELS, HEADS
from ..segmentors.ddp import LearnedSinusoidalPosEmb, _Conv1x1, _SamplerMixin
from .. import schedule


@HEADS.register_module(name='BEVDeformableHeadWithTime')
class BEVDeformableHeadWithTime(_SegHead):
    """runs BEVGridTransform before the encoder and returns sigmoid maps (reference :179-235)."""
    task = 'bev'

    def __init__(self, num_feature_levels=1, encoder=None, positional_encoding=None, classes=(), loss='focal',
                 grid_transform=None, in_channels=256, seg_conv_kernel=1, **kwargs):
        if seg_conv_kernel != 1:
            raise ValueError('only the 1x1 conv_seg of the DDP configs is implemented')
        self.classes = list(classes)
        self.loss = loss
        self.grid_transform = dict(grid_transform or {})
        if self.grid_transform.get('prescale_factor', 1) != 1:
            raise ValueError('prescale_factor != 1 is not implemented')
        super().__init__(num_feature_levels=num_feature_levels, encoder=encoder,
                         positional_encoding=positional_encoding, in_channels=[in_channels], channels=in_channels,
                         num_classes=len(self.classes), **kwargs)

    def _engine_kwargs(self):
        return dict(num_classes=self.num_classes, bev_input_scope=self.grid_transform['input_scope'],
                    bev_output_scope=self.grid_transform['output_scope'])

    def forward(self, inputs, times, target=None):
        return super().forward(inputs, times)


@FUSIONMODELS.register_module(name='BEVDDP')
class DDP(nn.Module, _SamplerMixin):
    task = 'bev'

    def __init__(self, bit_scale=1, timesteps=1, randsteps=1, time_difference=1, learned_sinusoidal_dim=16,
                 sample_range=(0, 0.999), noise_schedule='cosine', diffusion='ddim', threshold=0.5,
                 feat_channels=512, tmp_channels=256, **kwargs):
        super().__init__()
        if noise_schedule not in schedule.NOISE_SCHEDULES:
            raise ValueError(f'invalid noise schedule {noise_schedule}')
        if tmp_channels != 256:
            raise ValueError('libddp_mi355x is built for tmp_channels=256')
        if learned_sinusoidal_dim != 16:
            raise ValueError('libddp_mi355x is built for learned_sinusoidal_dim=16')
        self.bit_scale, self.timesteps, self.randsteps = bit_scale, timesteps, randsteps
        self.diffusion, self.time_difference, self.sample_range = diffusion, time_difference, sample_range
        self.noise_schedule = noise_schedule
        self.num_classes = 6
        self.threshold = threshold
        self.feat_channels = feat_channels
        self.embedding_table = nn.Embedding(self.num_classes + 1, tmp_channels)
        self.transform = _Conv1x1(tmp_channels + feat_channels, tmp_channels)
        self.time_mlp = nn.Sequential(LearnedSinusoidalPosEmb(learned_sinusoidal_dim),
                                      nn.Linear(learned_sinusoidal_dim + 1, tmp_channels * 4), nn.GELU(),
                                      nn.Linear(tmp_channels * 4, tmp_channels * 4))

    @torch.no_grad()
    def ddim_sample(self, x, head, noise=None):
        x0 = x[0]
        if not x0.is_cuda:
            raise RuntimeError('ddp_amd has no CPU path: features must live on an MI355X (HIP) device')
        b, c, h, w = x0.shape
        if noise is None:
            noise = torch.randn((b, self.randsteps, 256, h, w), device=x0.device)
        sd = dict(self.state_dict())
        sd.update({'decode_head.' + k: v for k, v in head.state_dict().items()})

        def factory():
            from ..engine import DDPEngine
            return DDPEngine(sd, 'bev', h=h, w=w, batch=b, randsteps=self.randsteps, timesteps=self.timesteps,
                             num_classes=self.num_classes, feat_channels=c, bit_scale=self.bit_scale,
                             time_difference=self.time_difference, noise_schedule=self.noise_schedule,
                             threshold=self.threshold, bev_input_scope=head.grid_transform['input_scope'],
                             bev_output_scope=head.grid_transform['output_scope'], device=x0.device)
        ver = sum(p._version for p in head.parameters())
        eng = self._get_engine((b, c, h, w, str(x0.device), self.timesteps, self.randsteps, ver), factory)
        return eng.sample(x0.contiguous().float(), noise.contiguous().float())
