"""Drop-in depth ``DDP`` + ``DeformableHeadWithTime`` (depth/depth/models/depther/ddp.py:34-247;
depth/depth/models/decode_heads/deformable_head_with_time.py:20-169): ``down`` concat-conv over
256+1 channels, raw-t time embedding, 3x3 ``conv_depth`` regression head, cosine-gamma DDIM step - and the toolbox's test
entry around it (depther/base.py:50-115 ``forward`` / ``forward_test``; encoder_decoder.py:130-235 ``whole_inference`` /
``inference`` / ``simple_test`` / ``aug_test``), so that ``depth/tools/test.py`` runs unchanged: ``model(return_loss=False,
**data)`` (depth/depth/apis/test.py:88,204).  Everything after the loop is ONE kernel (``ddp_depth_postprocess``)."""
import torch
import torch.nn as nn

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
        if classify:
            raise ValueError('the regression branches of depth_pred are implemented (relu + eps, and scale_up: sigmoid * eps; '
                             'decode_head.py:252-262); classify=True (binned depth, :236-250) is not - no DDP config uses it')
        self.min_depth, self.max_depth = min_depth, max_depth
        self.scale_up, self.classify, self.use_eps = scale_up, classify, use_eps
        kwargs.setdefault('num_classes', 1)
        super().__init__(**kwargs)

    def _make_head_conv(self):
        self.conv_depth = nn.Conv2d(self.channels, 1, kernel_size=3, padding=1, stride=1)

    def _engine_kwargs(self):
        return dict(min_depth=self.min_depth, max_depth=self.max_depth if self.max_depth is not None else 80.0,
                    depth_scale_up=bool(self.scale_up), depth_use_eps=bool(self.use_eps))


@DEPTHER.register_module(name='DepthDDP')
class DDP(nn.Module, _SamplerMixin):
    task = 'depth'

    def __init__(self, bit_scale=1, bits=8, timesteps=1, randsteps=1, time_difference=1, learned_sinusoidal_dim=16,
                 sample_range=(0, 0.999), ddim=True, rule=None, min_depth=1e-3, max_depth=80, backbone=None,
                 neck=None, decode_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        if not ddim:
            raise NotImplementedError('the reference references ddpm_step but never defines it (depther/ddp.py:244)')
        if learned_sinusoidal_dim != 16:
            raise ValueError('libddp_mi355x is built for learned_sinusoidal_dim=16')
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

        head = self.decode_head
        su, ue = bool(getattr(head, 'scale_up', False)), bool(getattr(head, 'use_eps', True))
        # the library takes ONE (min_depth, max_depth): the head's eps of depth_pred (decode_head.py:252-262) and the depther's x0
        # normalisation (depther/ddp.py:239) read the same pair in every shipped config
        hmin, hmax = getattr(head, 'min_depth', self.min_depth), getattr(head, 'max_depth', None)
        if ue and ((not su and hmin != self.min_depth) or (su and hmax is not None and hmax != self.max_depth)):
            raise ValueError(f'decode_head min / max depth ({hmin}, {hmax}) differ from the depther\'s ({self.min_depth}, {self.max_depth})')

        def factory():
            from ..engine import DDPEngine
            return DDPEngine(self.hot_path_state_dict(), 'depth', h=h, w=w, batch=b, randsteps=self.randsteps,
                             timesteps=self.timesteps, bit_scale=self.bit_scale, time_difference=self.time_difference,
                             min_depth=self.min_depth, max_depth=self.max_depth, depth_scale_up=su, depth_use_eps=ue, device=x.device)
        # keyed without the geometry: a new (b, h, w) re-uses the engine through set_geometry (no weight repacking)
        eng = self._get_engine(('depth', str(x.device), self.timesteps, self.randsteps, self.bit_scale, self.time_difference,
                                self.min_depth, self.max_depth, su, ue), factory, geometry=(b, h, w))
        return eng.sample(x.contiguous().float(), noise.contiguous().float())

    def _decode_head_forward_test(self, x, t, img_metas=None):
        return self.decode_head.forward_test(x, t, img_metas, self.test_cfg)

    # -- the toolbox's test entry (depth/depth/models/depther/base.py:50-115, encoder_decoder.py:130-235) ----------------
    @property
    def with_neck(self):
        return self.neck is not None

    @property
    def with_decode_head(self):
        return self.decode_head is not None

    @property
    def with_auxiliary_head(self):
        return False                     # training-only deep supervision: out of scope

    def _check_mode(self):
        """encoder_decoder.py:183-187: ``test_cfg.mode`` in ('slide', 'whole'); 'slide' is NotImplementedError there too."""
        cfg = self.test_cfg
        mode = (cfg.get('mode') if isinstance(cfg, dict) else getattr(cfg, 'mode', None)) if cfg is not None else None
        if mode not in (None, 'whole', 'slide'):
            raise AssertionError(f"test_cfg.mode must be 'slide' or 'whole', got {mode!r}")
        if mode == 'slide':
            raise NotImplementedError("test_cfg.mode='slide' (the reference raises here as well: encoder_decoder.py:186-187)")

    def _low_res(self, img, img_metas):
        """backbone + neck + the K-step loop: the (b,1,h/4,w/4) map every epilogue below starts from."""
        return self.sample(self.extract_feat(img)[0], img_metas)

    def _post(self, maps, flips, size):
        from ..engine import depth_postprocess
        # torch.clamp(out, min=head.min_depth, max=head.max_depth) (depther/ddp.py:101): a head built without max_depth does not
        # clamp from above
        lo = self.decode_head.min_depth if self.decode_head.min_depth is not None else float('-inf')
        hi = self.decode_head.max_depth if self.decode_head.max_depth is not None else float('inf')
        return depth_postprocess(maps, flips, size, lo, hi, self.align_corners)

    def encode_decode(self, img, img_metas=None, rescale=False):
        """depther/ddp.py:95-109: clamp to the head's depth range, resize to the network input when ``rescale`` - one fused
        kernel (``ddp_depth_postprocess``), no intermediate (b,1,h,w) clamp result."""
        d = self._low_res(img, img_metas)
        return self._post([d], [None], img.shape[2:] if rescale else d.shape[2:])

    def whole_inference(self, img, img_meta, rescale):
        """encoder_decoder.py:160-166."""
        return self.encode_decode(img, img_meta, rescale)

    @staticmethod
    def _flip_of(img_meta):
        if img_meta and img_meta[0].get('flip', False):
            direction = img_meta[0].get('flip_direction', 'horizontal')
            assert direction in ('horizontal', 'vertical')
            return direction
        return None

    def inference(self, img, img_meta, rescale):
        """encoder_decoder.py:168-196 (mode 'whole'): ``whole_inference`` with the test-time flip undone, the flip folded into
        the same kernel (it reads the mirrored source pixel)."""
        self._check_mode()
        if img_meta:
            ori_shape = img_meta[0]['ori_shape']
            assert all(m['ori_shape'] == ori_shape for m in img_meta)
        d = self._low_res(img, img_meta)
        return self._post([d], [self._flip_of(img_meta)], img.shape[2:] if rescale else d.shape[2:])

    def simple_test(self, img, img_meta, rescale=True):
        """encoder_decoder.py:198-209: list (batch) of (1,H,W) float32 arrays.  ``img`` may hold b >= 1 images (independent
        noise per image; the reference's sampler is b = 1 only, depther/ddp.py:232)."""
        return list(self.inference(img, img_meta, rescale).cpu().numpy())

    def aug_test(self, imgs, img_metas, rescale=True):
        """encoder_decoder.py:210-229: mean over the augmentations (KITTI / NYU test pipelines: plain + horizontal flip,
        depth/configs/_base_/datasets/kitti.py:30-33).  Every augmentation runs the full sampling loop with its own noise; only the
        LOW-RESOLUTION maps are kept and ONE kernel does clamp -> resize -> flip-undo -> running sum -> / n per output pixel."""
        assert rescale, 'aug_test rescales every augmentation back to the network input'
        self._check_mode()
        sizes = {tuple(img.shape[2:]) for img in imgs}
        if len(sizes) != 1:
            # the reference adds the per-augmentation maps in place (:222-224): they must have one size
            raise RuntimeError(f'aug_test: augmentations of different input sizes {sorted(sizes)} cannot be averaged')
        maps, flips = [], []
        for img, meta in zip(imgs, img_metas):
            if meta:
                ori_shape = meta[0]['ori_shape']
                assert all(m['ori_shape'] == ori_shape for m in meta)
            maps.append(self._low_res(img, meta))
            flips.append(self._flip_of(meta))
        return list(self._post(maps, flips, imgs[0].shape[2:]).cpu().numpy())

    def forward_test(self, imgs, img_metas, **kwargs):
        """base.py:62-92: the outer lists are the test-time augmentations."""
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        for img_meta in img_metas:
            for key in ('ori_shape', 'img_shape', 'pad_shape'):
                vals = [m[key] for m in img_meta if key in m]
                assert all(v == vals[0] for v in vals), f'{key} differs inside one augmentation batch'
        if len(imgs) == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """base.py:94-108; what the harness calls: ``model(return_loss=False, **data)`` (depth/depth/apis/test.py:88,204)."""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_dummy(self, img):
        """encoder_decoder.py:124-128."""
        return self.encode_decode(img, None)

    def val_step(self, data_batch, **kwargs):
        """base.py:150-158."""
        return self(**data_batch, **kwargs)

    def forward_train(self, *a, **k):
        raise NotImplementedError('training is out of scope of ddp_amd (SURVEY.md §8)')

    def train_step(self, *a, **k):
        raise NotImplementedError('training is out of scope of ddp_amd (SURVEY.md §8)')
