"""Drop-in ``DDP`` segmentor (plugin surface #1) for MMSegmentation-style configs.

Keeps the constructor kwargs, attribute names, ``state_dict`` keys and the inference methods of
segmentation/mmseg/models/segmentors/ddp.py:49-290 (``extract_feat``, ``encode_decode``, ``whole_inference``,
``inference``, ``simple_test``, ``aug_test``, ``ddim_sample``, ``ddpm_sample``, ``_decode_head_forward_test``,
``_get_sampling_timesteps``) while
the K-step loop itself runs in libddp_mi355x.so through ``DDPEngine``.  Unlike the reference, whose
sampler only works for one image per call (ddp.py:219-223 allocates the noisy map with batch
``randsteps``), ``ddim_sample`` accepts b >= 1 images and draws independent noise for each.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import schedule
from ..registry import SEGMENTORS, build_backbone, build_head, build_neck


class LearnedSinusoidalPosEmb(nn.Module):
    """parameter holder for time_mlp.0 (ddp.py:31-46); evaluated on device by k_sinusoid."""

    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))


class _Conv1x1(nn.Module):
    """``ConvModule(cin, cout, 1, norm_cfg=None, act_cfg=None)`` parameter holder: key ``conv.*``."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1)


def _build_neck(neck):
    if neck is None:
        return None
    if isinstance(neck, (list, tuple)):      # the DDP configs chain FPN -> MultiStageMerging: one fused C entry (necks/chain.py)
        from ..necks import NeckChain
        return NeckChain(*[build_neck(n) for n in neck])
    return build_neck(neck)


class _SamplerMixin:
    """engine cache shared by the task variants.

    One ``PackedWeights`` blob per (device, weights version) and a small LRU of engines keyed by everything EXCEPT the
    geometry: a new (batch, h, w) - the normal case under the reference's test protocol, one keep-ratio-resized image per
    call (segmentation/tools/test.py:214-219, mmseg/apis/test.py:87-89) - re-uses the engine through ``set_geometry`` (positional
    tables + workspace carve only; no weight repacking, no ``ddp_prepare``)."""
    ENGINE_CACHE_SIZE = 4

    def _weights_version(self):
        return sum(p._version for p in self.parameters())

    def _packed_weights(self, device, task, num_layers):
        from ..engine import PackedWeights
        ver = self._weights_version()
        cache = self.__dict__.setdefault('_weights_cache', {})
        key = (str(device), task, num_layers)
        hit = cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, PackedWeights(self.hot_path_state_dict(), task, num_layers, device))
            cache[key] = hit
        return hit[1]

    def _get_engine(self, key, factory, geometry=None):
        cache = self.__dict__.setdefault('_engine_cache', {})
        ver = self._weights_version()
        for k in [k for k in cache if k[-1] != ver]:      # weights changed: every cached engine is stale
            del cache[k]
        key = key + (ver,)
        eng = cache.pop(key, None)
        if eng is None:
            eng = factory()
            while len(cache) >= self.ENGINE_CACHE_SIZE:
                cache.pop(next(iter(cache)))               # least recently used
        elif geometry is not None:
            eng.set_geometry(*geometry)
        cache[key] = eng                                   # most recently used last
        return eng


@SEGMENTORS.register_module()
class DDP(nn.Module, _SamplerMixin):
    task = 'seg'

    def __init__(self, bit_scale=0.1, timesteps=1, randsteps=1, time_difference=1, learned_sinusoidal_dim=16,
                 sample_range=(0, 0.999), noise_schedule='cosine', diffusion='ddim', accumulation=False,
                 backbone=None, neck=None, decode_head=None, auxiliary_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__()
        if noise_schedule not in schedule.NOISE_SCHEDULES:
            raise ValueError(f'invalid noise schedule {noise_schedule}')            # ddp.py:90
        if learned_sinusoidal_dim != 16:
            raise ValueError('libddp_mi355x is built for learned_sinusoidal_dim=16')
        self.backbone = build_backbone(backbone) if backbone is not None else None
        self.neck = _build_neck(neck)
        self.decode_head = build_head(decode_head)
        self.auxiliary_head = None           # training-only deep supervision: out of scope
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bit_scale, self.timesteps, self.randsteps = bit_scale, timesteps, randsteps
        self.diffusion, self.time_difference = diffusion, time_difference
        self.sample_range, self.noise_schedule = sample_range, noise_schedule
        self.use_gt = False
        self.accumulation = accumulation
        c = self.decode_head.in_channels[0]
        self.embedding_table = nn.Embedding(self.num_classes + 1, c)
        self.transform = _Conv1x1(c * 2, c)
        time_dim = c * 4
        self.time_mlp = nn.Sequential(LearnedSinusoidalPosEmb(learned_sinusoidal_dim),
                                      nn.Linear(learned_sinusoidal_dim + 1, time_dim), nn.GELU(),
                                      nn.Linear(time_dim, time_dim))

    # -- plugin surface ------------------------------------------------------------------------
    @property
    def with_neck(self):
        return self.neck is not None

    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    def _get_sampling_timesteps(self, batch, *, device):
        times = []
        for t_now, t_next in schedule.get_sampling_timesteps(self.timesteps, self.time_difference,
                                                             self.sample_range[0]):
            t = torch.tensor([t_now, t_next], device=device)
            times.append(t[:, None].repeat(1, batch))
        return times

    def _engine_for(self, b, h, w, device, sampler, timesteps=None, randsteps=None, accumulation=None, kind='sample'):
        from ..decode_heads.fcn_head_with_time import FCNHeadWithTime
        K = self.timesteps if timesteps is None else timesteps
        r = self.randsteps if randsteps is None else randsteps
        acc = self.accumulation if accumulation is None else accumulation
        common = dict(batch=b, randsteps=r, timesteps=K, num_classes=self.num_classes, bit_scale=self.bit_scale,
                      time_difference=self.time_difference, sample_range0=self.sample_range[0],
                      noise_schedule=self.noise_schedule, sampler=sampler, accumulation=acc, device=device)
        key = (kind, str(device), sampler, K, r, acc, self.bit_scale, self.time_difference, self.sample_range[0],
               self.noise_schedule)
        if isinstance(self.decode_head, FCNHeadWithTime):
            # any registered head goes through _decode_head_forward_test in the reference (ddp.py:192-196); here the loop
            # around FCNHeadWithTime is its own C entry (ddp_sample_fcn), one engine per geometry
            def factory():
                from ..engine import FcnSamplerEngine
                return FcnSamplerEngine(self.hot_path_state_dict(), self.decode_head, h=h, w=w, **common)
            return self._get_engine(key + ('fcn', b, h, w), factory)

        def factory():
            from ..engine import DDPEngine, count_layers
            nl = count_layers(self.hot_path_state_dict())
            return DDPEngine(None, 'seg', h=h, w=w, feat_channels=256, weights=self._packed_weights(device, 'seg', nl), **common)
        return self._get_engine(key, factory, geometry=(b, h, w))

    def hot_path_state_dict(self):
        return {k: v for k, v in self.state_dict().items()
                if not k.startswith(('backbone.', 'neck.', 'auxiliary_head.'))}

    def _check_feature(self, x):
        if not x.is_cuda:
            raise RuntimeError('ddp_amd has no CPU path: features must live on an MI355X (HIP) device')
        if x.shape[1] != self.decode_head.in_channels[0]:
            raise RuntimeError(f'expected {self.decode_head.in_channels[0]} feature channels, got {x.shape[1]}')

    @torch.no_grad()
    def ddim_sample(self, x, img_metas=None, noise=None):
        """x (b,256,h,w) -> (b,K,h,w).  ``noise`` (b,r,256,h,w) may be injected (parity tests);
        by default it is drawn with torch.randn like the reference (ddp.py:220)."""
        self._check_feature(x)
        b, c, h, w = x.shape
        if noise is None:
            noise = torch.randn((b, self.randsteps, c, h, w), device=x.device)
        eng = self._engine_for(b, h, w, x.device, 'ddim')
        return eng.sample(x.contiguous().float(), noise.contiguous().float())

    @torch.no_grad()
    def ddpm_sample(self, x, img_metas=None, noise=None, step_noise=None):
        self._check_feature(x)
        b, c, h, w = x.shape
        if noise is None:
            noise = torch.randn((b, self.randsteps, c, h, w), device=x.device)
        if step_noise is None:
            step_noise = torch.randn((self.timesteps, b, self.randsteps, c, h, w), device=x.device)
        eng = self._engine_for(b, h, w, x.device, 'ddpm')
        return eng.sample(x.contiguous().float(), noise.contiguous().float(), step_noise.contiguous().float())

    def _decode_head_forward_test(self, x, t, img_metas=None):
        return self.decode_head.forward_test(x, t, img_metas, self.test_cfg)

    def encode_decode(self, img, img_metas=None):
        """ddp.py:114-129."""
        x = self.extract_feat(img)[0]
        if self.diffusion == 'ddim':
            out = self.ddim_sample(x, img_metas)
        elif self.diffusion == 'ddpm':
            out = self.ddpm_sample(x, img_metas)
        else:
            raise NotImplementedError
        return F.interpolate(out, size=img.shape[2:], mode='bilinear', align_corners=self.align_corners)

    def whole_inference(self, img, img_meta=None, rescale=False):
        """encoder_decoder.py:229-248: crop to img_shape and resize to ori_shape when rescale."""
        seg_logit = self.encode_decode(img, img_meta)
        if rescale and img_meta:
            rs = img_meta[0]['img_shape'][:2]
            seg_logit = seg_logit[:, :, :rs[0], :rs[1]]
            seg_logit = F.interpolate(seg_logit, size=tuple(img_meta[0]['ori_shape'][:2]), mode='bilinear',
                                      align_corners=self.align_corners)
        return seg_logit

    def _mode(self):
        """encoder_decoder.py:266-271: ``test_cfg.mode`` in ('slide', 'whole'); every shipped DDP config is 'whole'
        (configs/ade/*:111, configs/cityscapes/*:96-98)."""
        cfg = self.test_cfg
        mode = (cfg.get('mode') if isinstance(cfg, dict) else getattr(cfg, 'mode', None)) if cfg is not None else None
        if mode not in (None, 'whole', 'slide'):
            raise AssertionError(f"test_cfg.mode must be 'slide' or 'whole', got {mode!r}")
        return mode or 'whole'

    def _sample(self, x, img_meta=None):
        if self.diffusion == 'ddim':
            return self.ddim_sample(x, img_meta)
        if self.diffusion == 'ddpm':
            return self.ddpm_sample(x, img_meta)
        raise NotImplementedError

    SLIDE_WINDOWS_PER_CALL = 16          # windows x images sampled in one call of the loop (memory bound of the workspace)

    def _slide(self, img, img_meta, rescale, want, flip=None):
        """Sliding-window inference (encoder_decoder.py:180-227).  The reference runs ``encode_decode`` - backbone, neck and a
        K-step loop - window by window; the windows all have one size, so here they go through backbone + loop as ONE batch
        (chunks of SLIDE_WINDOWS_PER_CALL maps, independent noise per window as in the reference) and only their LOW-RESOLUTION
        scores are kept.  ``ddp_seg_slide_postprocess`` then does, per output pixel, what the reference does with image-size
        tensors: resize per window, ``preds += pad(...)`` in window order, ``/ count_mat``, crop to img_shape, resize to
        ori_shape, [softmax, flip-undo, argmax]."""
        from ..engine import seg_slide_postprocess, slide_windows
        cfg = self.test_cfg
        get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
        ys, xs, (ch, cw) = slide_windows(img.shape[2:], get('crop_size'), get('stride'))
        b = img.shape[0]
        crops = [img[:, :, y1:y1 + ch, x1:x1 + cw] for y1 in ys for x1 in xs]
        per = max(1, self.SLIDE_WINDOWS_PER_CALL // b)
        lows = []
        for i in range(0, len(crops), per):
            chunk = crops[i:i + per]
            x = self.extract_feat(torch.cat(chunk, dim=0))[0]
            lows.append(self._sample(x, img_meta).reshape(len(chunk), b, self.num_classes, x.shape[2], x.shape[3]))
        scores = torch.cat(lows, dim=0)
        keep = out = None
        if rescale and img_meta:
            keep = tuple(img_meta[0]['img_shape'][:2])
            out = tuple(img_meta[0]['ori_shape'][:2])
        return seg_slide_postprocess(scores, ys, xs, (ch, cw), img.shape[2:], keep, out, self.align_corners, flip, want)

    def slide_inference(self, img, img_meta, rescale):
        """encoder_decoder.py:180-227: the window-averaged scores (b,K,H,W) (at ori_shape when ``rescale``)."""
        return self._slide(img, img_meta, rescale, 'scores')

    def inference(self, img, img_meta, rescale):
        """encoder_decoder.py:251-287, mode 'whole' (what every DDP config sets): class probabilities at ``ori_shape`` with
        the test-time flip undone - the building block of ``aug_test``.  ``simple_test`` does not go through here: its
        fused epilogue never materialises these (B,K,H,W) tensors."""
        mode = self._mode()
        if img_meta:
            ori_shape = img_meta[0]['ori_shape']
            assert all(m['ori_shape'] == ori_shape for m in img_meta)
        if mode == 'slide':
            fl = None
            if img_meta and img_meta[0].get('flip', False):
                fl = img_meta[0].get('flip_direction', 'horizontal')
                assert fl in ('horizontal', 'vertical')
            return self._slide(img, img_meta, rescale, 'prob', fl)
        output = F.softmax(self.whole_inference(img, img_meta, rescale), dim=1)
        if img_meta and img_meta[0].get('flip', False):
            direction = img_meta[0].get('flip_direction', 'horizontal')
            assert direction in ('horizontal', 'vertical')
            output = output.flip(dims=(3,) if direction == 'horizontal' else (2,))
        return output

    def aug_test(self, imgs, img_metas, rescale=True):
        """encoder_decoder.py:306-331: mean of the per-augmentation probabilities (multi-scale / flip), then argmax.
        Every augmentation runs the full sampling loop with its own noise, as in the reference; only the LOW-RESOLUTION
        scores of each are kept, and one fused epilogue (``ddp_seg_aug_postprocess``) does resize -> crop -> resize ->
        softmax -> flip-undo for all of them, the mean and the argmax per output pixel: the (1,K,H,W) tensors ``inference``
        materialises per augmentation and the running sum at ori_shape never exist."""
        from ..engine import seg_aug_postprocess
        assert rescale, 'aug_test rescales every augmentation back to ori_shape'
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        if self._mode() == 'slide':
            # no shipped config combines sliding windows with test-time augmentation: the reference's own composition
            # (running mean of ``inference``, encoder_decoder.py:306-331) over the fused per-augmentation slide epilogue
            prob = self.inference(imgs[0], img_metas[0], rescale)
            for img, meta in zip(imgs[1:], img_metas[1:]):
                prob += self.inference(img, meta, rescale)
            prob /= len(imgs)
            return list(prob.argmax(dim=1).cpu().numpy().astype('int64'))
        ori_shape = tuple(img_metas[0][0]['ori_shape'][:2])
        scores, metas = [], []
        for img, meta in zip(imgs, img_metas):
            assert all(tuple(m['ori_shape'][:2]) == ori_shape for m in meta)
            x = self.extract_feat(img)[0]
            if self.diffusion == 'ddim':
                scores.append(self.ddim_sample(x, meta))
            elif self.diffusion == 'ddpm':
                scores.append(self.ddpm_sample(x, meta))
            else:
                raise NotImplementedError
            metas.append(dict(img_size=tuple(img.shape[2:]), crop_size=tuple(meta[0]['img_shape'][:2]),
                              flip=meta[0].get('flip_direction', 'horizontal') if meta[0].get('flip', False) else None))
        seg = seg_aug_postprocess(scores, metas, ori_shape, self.align_corners)
        return list(seg.cpu().numpy().astype('int64'))

    @staticmethod
    def _epilogue_args(meta, rescale):
        crop = out_size = flip = None
        if meta:
            if rescale:
                crop = tuple(meta['img_shape'][:2])
                out_size = tuple(meta['ori_shape'][:2])
            if meta.get('flip', False):
                flip = meta.get('flip_direction', 'horizontal')
        return crop, out_size, flip

    def simple_test(self, img, img_meta=None, rescale=True):
        """encoder_decoder.py:250-304 (mode='whole'), post-loop epilogue fused into one kernel:
        the (1,K,H,W) resized scores / probabilities of the reference are never materialised.  ``img`` may hold
        b >= 1 images (§8 f4): the loop runs once on the whole batch with independent noise per image, and every
        image gets the crop / rescale / flip of its OWN ``img_meta`` entry (one epilogue launch per distinct geometry)."""
        from ..engine import seg_postprocess
        if self._mode() == 'slide':
            if img_meta and any(self._epilogue_args(m, rescale) != self._epilogue_args(img_meta[0], rescale) for m in img_meta):
                raise ValueError('slide inference: the images of a batch must share img_shape / ori_shape / flip')
            fl = self._epilogue_args(img_meta[0], rescale)[2] if img_meta else None
            # ('seg' = argmax of the window-averaged SCORES; the reference takes argmax of their softmax, encoder_decoder.py:277,296.
            # softmax is monotone, so the two agree except where fp32 rounding of exp / the division makes two probabilities EQUAL
            # that came from different scores - the reference then returns the lower class index, this path the class with the
            # larger score.  test_slide_epilogue_golden bounds it: identical wherever the reference's top-2 margin exceeds 1e-5.)
            return list(self._slide(img, img_meta, rescale, 'seg', fl).cpu().numpy().astype('int64'))
        x = self.extract_feat(img)[0]
        if self.diffusion == 'ddim':
            out = self.ddim_sample(x, img_meta)
        elif self.diffusion == 'ddpm':
            out = self.ddpm_sample(x, img_meta)
        else:
            raise NotImplementedError
        b = out.shape[0]
        if img_meta and len(img_meta) not in (1, b):
            raise ValueError(f'{len(img_meta)} img_metas for a batch of {b} images')
        args = [self._epilogue_args(img_meta[i if len(img_meta) == b else 0] if img_meta else None, rescale)
                for i in range(b)]
        if all(a == args[0] for a in args):
            seg = seg_postprocess(out, img.shape[2:], args[0][0], args[0][1], self.align_corners, args[0][2])
            return list(seg.cpu().numpy().astype('int64'))
        res = []
        for i, (crop, out_size, flip) in enumerate(args):
            seg = seg_postprocess(out[i:i + 1].contiguous(), img.shape[2:], crop, out_size, self.align_corners, flip)
            res.append(seg[0].cpu().numpy().astype('int64'))
        return res

    def forward(self, img, img_metas=None, return_loss=False, rescale=True, **kwargs):
        """base.py:62-110 ``forward`` / ``forward_test``: ``img`` / ``img_metas`` may be the one-element
        augmentation lists the test pipeline produces."""
        if return_loss:
            raise NotImplementedError('training is out of scope of ddp_amd (SURVEY.md §8)')
        if isinstance(img, (list, tuple)):
            if len(img) != 1:
                if img_metas is None or len(img_metas) != len(img):
                    raise ValueError(f'num of augmentations ({len(img)}) != num of image meta ({0 if img_metas is None else len(img_metas)})')
                return self.aug_test(list(img), list(img_metas), rescale)
            img = img[0]
            if img_metas is not None:
                img_metas = img_metas[0]
        return self.simple_test(img, img_metas, rescale)

    def forward_train(self, *a, **k):
        raise NotImplementedError('training is out of scope of ddp_amd (SURVEY.md §8)')


@SEGMENTORS.register_module()
class SelfAlignedDDP(DDP):
    """segmentation/mmseg/models/segmentors/self_aligned_ddp.py:48-129: the self-aligned variant differs from DDP only
    in ``forward_train`` (:131-186); its inference surface - constructor kwargs, state_dict keys, ``encode_decode`` /
    ``ddim_sample`` / ``ddpm_sample`` - is DDP's, so the two Cityscapes ``*_aligned`` configs resolve to the same MI355X
    path.  The one piece of ``forward_train`` that is an inference pass - the no-grad "self-aligned denoising" pre-pass
    (:150-164) - is exposed as ``self_aligned_predict``."""

    @torch.no_grad()
    def self_aligned_predict(self, x, noise=None, return_logits=False):
        """self_aligned_ddp.py:150-164: ONE decoder pass at t = 1 on pure noise, then the x0 projection:
            feat = transform(cat[x, noise]); logits = decode_head(feat, time_mlp(log_snr(1)))
            preds = (sigmoid(embedding_table(argmax(logits))) * 2 - 1) * bit_scale
        x (b,256,h,w) -> preds (b,256,h,w) [, logits (b,K,h,w)].  ``noise`` (b,256,h,w) replaces ``torch.randn_like(x)``.
        Runs as a 1-step sampler call (the step's t_now is 1 for every K; no update, no accumulation) followed by
        ``ddp_seg_x0_project``."""
        import ctypes as C

        from .. import _lib
        self._check_feature(x)
        b, c, h, w = x.shape
        if noise is None:
            noise = torch.randn_like(x)

        # the head dispatch of _engine_for (the reference's pre-pass goes through the generic _decode_head_forward_test)
        eng = self._engine_for(b, h, w, x.device, 'ddim', timesteps=1, randsteps=1, accumulation=False, kind='self_aligned')
        logits = eng.sample(x.contiguous().float(), noise.reshape(b, 1, c, h, w).contiguous().float())
        preds = torch.empty((b, 256, h, w), dtype=torch.float32, device=x.device)
        emb = self.embedding_table.weight.detach().float().contiguous()
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().ddp_seg_x0_project(logits.data_ptr(), b, self.num_classes, h * w, emb.data_ptr(),
                                                      C.c_float(self.bit_scale), preds.data_ptr(),
                                                      torch.cuda.current_stream(x.device).cuda_stream))
        return (preds, logits) if return_logits else preds
