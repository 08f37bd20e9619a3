#!/usr/bin/env python
"""Generate golden vectors for the DDP hot path FROM THE REFERENCE ITSELF.

Run by hand in the build container (needs /root/reference):

    python tests/golden/gen_golden.py            # all tasks, one subprocess each (~1 min)
    python tests/golden/gen_golden.py --task seg
    python tests/golden/gen_golden.py --task fullsize   # NOT part of 'all' (~3 min): the reference at the sizes of BASELINE.json's
                                                        # configurations (full_c1 .. full_c5_r4), see FULLSIZE_* below

The reference packages are imported through ``ref_shim`` (mmcv 1.3.17 python bricks vendored in
the reference tree, torch CPU fp32), the seeded synthetic hot-path weights of
``ddp_amd.utils.synthetic`` are loaded with ``load_state_dict`` and the in-method ``torch.randn``
is replaced by the seeded start noise, so that the oracle / HIP path can be fed bit-identical
inputs.  Only *data* is written: per case one ``tests/golden/<case>.npz`` holding the config, the
input fingerprints and the reference outputs (final output, per-step logits / noisy map, and a few
per-layer activations for the smallest case).  Inputs and weights are NOT stored - they are
regenerated from the seeds (the fingerprints guard against generator drift).
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ddp_amd.utils import synthetic  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
from golden_util import class_projection_weights  # noqa: E402

SEG_CASES = [
    # ADE-like: 150 classes, accumulation (configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py:13-15)
    dict(name='seg_ade_k3', h=16, w=24, num_classes=150, timesteps=3, randsteps=1, bit_scale=0.01,
         accumulation=True, noise_schedule='cosine', diffusion='ddim', seed=0, trace=False),
    # Cityscapes-like: 19 classes, no accumulation, odd map size, two noise replicas, full trace
    dict(name='seg_city_r2', h=13, w=17, num_classes=19, timesteps=3, randsteps=2, bit_scale=0.01,
         accumulation=False, noise_schedule='cosine', diffusion='ddim', seed=1, trace=True),
    # 1-step plumbing case (BASELINE.json configs[0] shape class, scaled down)
    dict(name='seg_ade_k1', h=32, w=32, num_classes=150, timesteps=1, randsteps=1, bit_scale=0.01,
         accumulation=True, noise_schedule='cosine', diffusion='ddim', seed=2, trace=False),
    # 10-step override (configs[2]) + accumulation + r=2
    dict(name='seg_city_k10', h=12, w=20, num_classes=19, timesteps=10, randsteps=2, bit_scale=0.01,
         accumulation=True, noise_schedule='cosine', diffusion='ddim', seed=3, trace=False),
    # linear schedule, larger bit scale, non-zero sample_range[0]
    dict(name='seg_linear', h=9, w=33, num_classes=150, timesteps=4, randsteps=1, bit_scale=0.1,
         accumulation=False, noise_schedule='linear', diffusion='ddim', seed=4, trace=False,
         sample_range=(0.1, 0.999)),
    # ancestral sampler (segmentors/ddp.py:248-290)
    dict(name='seg_ddpm', h=10, w=14, num_classes=19, timesteps=4, randsteps=1, bit_scale=0.01,
         accumulation=True, noise_schedule='cosine', diffusion='ddpm', seed=5, trace=False),
    # time_difference != 1 (segmentors/ddp.py:204-213: t_next = max(1 - (step + 1 + td) / K, sample_range[0])): 0 = the plain
    # DDIM grid, 2 = the last two steps both jump to t = 0
    dict(name='seg_td0', h=9, w=14, num_classes=19, timesteps=3, randsteps=1, bit_scale=0.01,
         accumulation=True, noise_schedule='cosine', diffusion='ddim', seed=6, trace=False, time_difference=0),
    dict(name='seg_td2', h=8, w=11, num_classes=150, timesteps=4, randsteps=2, bit_scale=0.01,
         accumulation=False, noise_schedule='cosine', diffusion='ddim', seed=7, trace=False, time_difference=2),
    # content-dependent weight profiles (ddp_amd/utils/synthetic.py PROFILES; VERDICT r05 "next" #4): sampling offsets that react
    # to the query by +- 2.4 px, peaked attention, 8x class scores ('trained_like': the network amplifies rounding - the reference is
    # 1.1e-3 from its own fp64 evaluation on this map with every decision equal), and points spread +- 6 px around their head's mean
    # ('wide_offsets': every tap group leaves the gather's staged window, many taps leave the map) - on a small map
    dict(name='seg_trained_small', h=24, w=40, num_classes=19, timesteps=3, randsteps=1, bit_scale=0.01,
         accumulation=True, noise_schedule='cosine', diffusion='ddim', seed=8, trace=False, profile='trained_like'),
    dict(name='seg_wide_offsets', h=24, w=40, num_classes=19, timesteps=3, randsteps=1, bit_scale=0.01,
         accumulation=True, noise_schedule='cosine', diffusion='ddim', seed=9, trace=False, profile='wide_offsets'),
]

DEPTH_CASES = [
    dict(name='depth_k3_r2', h=11, w=19, timesteps=3, randsteps=2, bit_scale=0.1, seed=10,
         min_depth=1e-3, max_depth=80.0),
    dict(name='depth_k20', h=8, w=12, timesteps=20, randsteps=1, bit_scale=0.1, seed=11,
         min_depth=1e-3, max_depth=80.0),
    dict(name='depth_td2', h=7, w=13, timesteps=4, randsteps=1, bit_scale=0.1, seed=12,
         min_depth=1e-3, max_depth=80.0, time_difference=2),
    # the other regression branches of depth_pred (depth/depth/models/decode_heads/decode_head.py:252-262): sigmoid * max_depth, eps = 0
    dict(name='depth_scale_up', h=9, w=10, timesteps=3, randsteps=1, bit_scale=0.1, seed=13,
         min_depth=1e-3, max_depth=10.0, scale_up=True),
    dict(name='depth_no_eps', h=6, w=15, timesteps=3, randsteps=2, bit_scale=0.1, seed=14,
         min_depth=1e-3, max_depth=80.0, use_eps=False),
]

BEV_CASES = [
    # fusion variant: 512 feature channels; scopes scaled so that x is 16x16 and the head runs at 25x25
    dict(name='bev_fusion', h=16, w=16, feat_channels=512, timesteps=3, randsteps=2, bit_scale=0.01,
         num_layers=5, seed=20, input_scope=[[-51.2, 51.2, 6.4], [-51.2, 51.2, 6.4]],
         output_scope=[[-50, 50, 4.0], [-50, 50, 4.0]]),
    # camera variant: 256 feature channels, rectangular
    dict(name='bev_camera', h=12, w=20, feat_channels=256, timesteps=2, randsteps=1, bit_scale=0.01,
         num_layers=5, seed=21, input_scope=[[-51.2, 51.2, 8.533333333333333], [-51.2, 51.2, 5.12]],
         output_scope=[[-50, 50, 5.0], [-50, 50, 3.125]]),
    # the SHIPPED sampler settings (bev/configs/nuscenes/seg/ddp-fusion-bev256d2-lss-scale001-d5-lr5e-5.yaml:5 randsteps: 4;
    # ddp-camera-bev256d2-lss-scale001-d5-lr5e-5.yaml:6 randsteps: 5) on small maps
    dict(name='bev_fusion_r4', h=16, w=16, feat_channels=512, timesteps=3, randsteps=4, bit_scale=0.01,
         num_layers=5, seed=22, input_scope=[[-51.2, 51.2, 6.4], [-51.2, 51.2, 6.4]],
         output_scope=[[-50, 50, 5.0], [-50, 50, 5.0]]),
    dict(name='bev_camera_r5', h=14, w=10, feat_channels=256, timesteps=3, randsteps=5, bit_scale=0.01,
         num_layers=5, seed=23, input_scope=[[-51.2, 51.2, 7.314285714285714], [-51.2, 51.2, 10.24]],
         output_scope=[[-50, 50, 4.0], [-50, 50, 6.25]]),
]


ONLY = set()


def fingerprint(t):
    t = t.double()
    return [float(t.sum()), float(t.abs().sum())]


class RandnPatch:
    """Replace torch.randn / torch.randn_like inside the reference samplers by queued tensors."""

    def __init__(self, start, per_step=None):
        self.queue = [start]
        self.per_step = list(per_step) if per_step is not None else []

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like

        def randn(*a, **k):
            t = self.queue.pop(0)
            return t.clone()

        def randn_like(x, *a, **k):
            t = self.per_step.pop(0)
            assert t.shape == x.shape
            return t.clone()

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def wrap_forward(module, sink):
    """Record every result of ``module.forward`` (the reference calls head.forward_test ->
    self.forward directly, which bypasses forward hooks)."""
    orig = module.forward

    def wrapped(*a, **k):
        o = orig(*a, **k)
        sink.append(o.clone())
        return o

    module.forward = wrapped


def save(name, cfg, arrays):
    path = os.path.join(HERE, name + '.npz')
    arrays = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(path, config=np.array(json.dumps(cfg)), **arrays)
    print(f'wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)')


def load_hot_path(model, sd):
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    missing_hot = [k for k in res.missing_keys
                   if not k.startswith(('backbone.', 'neck.', 'auxiliary_head.'))]
    assert not missing_hot, missing_hot


def gen_seg():
    import ref_shim
    build_segmentor, Config, revert = ref_shim.import_seg()
    cfg_path = os.path.join(ref_shim.REF, 'segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py')
    for case in SEG_CASES:
        if ONLY and case['name'] not in ONLY:
            continue
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.timesteps = case['timesteps']
        m.randsteps = case['randsteps']
        m.bit_scale = case['bit_scale']
        m.accumulation = case['accumulation']
        m.noise_schedule = case['noise_schedule']
        m.diffusion = case['diffusion']
        if 'sample_range' in case:
            m.sample_range = tuple(case['sample_range'])
        if 'time_difference' in case:
            m.time_difference = case['time_difference']
        m.decode_head.num_classes = case['num_classes']
        m.auxiliary_head.num_classes = case['num_classes']
        model = revert(build_segmentor(m)).eval()
        sd = synthetic.make_state_dict('seg', case['num_classes'], 6, 256, seed=case['seed'] + 100, profile=case.get('profile', 'init'))
        load_hot_path(model, sd)
        x, noise = synthetic.make_inputs(1, case['h'], case['w'], case['randsteps'], 256, 256, seed=case['seed'])
        step_noise = None
        if case['diffusion'] == 'ddpm':
            g = torch.Generator().manual_seed(case['seed'] + 7)
            step_noise = torch.randn((case['timesteps'], case['randsteps'], 256, case['h'], case['w']), generator=g)

        rec = dict(feat=[], logits=[], layer0=[], layer_last=[], mask_in=[])
        def grab(key, sel=lambda i, o: o):
            def hook(mod, i, o):
                rec[key].append(sel(i, o).clone())      # returns None: never replaces the output
            return hook

        def grab_transform(mod, i, o):
            rec['feat'].append(o.clone())
            rec['mask_in'].append(i[0][:, 256:].clone())

        wrap_forward(model.decode_head, rec['logits'])   # forward_test calls .forward directly
        hooks = [
            model.transform.register_forward_hook(grab_transform),
            model.decode_head.encoder.layers[0].register_forward_hook(grab('layer0')),
            model.decode_head.encoder.layers[-1].register_forward_hook(grab('layer_last')),
        ]
        with RandnPatch(noise[0], None if step_noise is None else list(step_noise)):
            out = model.ddim_sample(x, None) if case['diffusion'] == 'ddim' else model.ddpm_sample(x, None)
        for hk in hooks:
            hk.remove()
        arrays = dict(out=out, logits_last=rec['logits'][-1],
                      x_fp=fingerprint(x), noise_fp=fingerprint(noise), weights_fp=synthetic.checksum(sd))
        if case['trace']:
            arrays['feat_step0'] = rec['feat'][0]
            arrays['layer0_step0'] = rec['layer0'][0]            # (N, r, 256) seq-first as in the reference
            arrays['layer_last_step0'] = rec['layer_last'][0]
            arrays['logits_steps'] = torch.stack(rec['logits'])
            arrays['mask_t_steps'] = torch.stack(rec['mask_in'][1:])   # noisy map entering steps 1..K-1
        save(case['name'], dict(task='seg', **case), arrays)


def gen_depth():
    import ref_shim
    build_depther, Config = ref_shim.import_depth()
    from mmcv.cnn.utils import revert_sync_batchnorm
    cfg_path = os.path.join(ref_shim.REF, 'depth/configs/ddp_kitti/ddp_swint_1k_w7_kitti_bs2x8_scale01.py')
    for case in DEPTH_CASES:
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.timesteps = case['timesteps']
        m.randsteps = case['randsteps']
        m.bit_scale = case['bit_scale']
        m.min_depth = case['min_depth']
        m.max_depth = case['max_depth']
        if 'time_difference' in case:
            m.time_difference = case['time_difference']
        m.decode_head.min_depth, m.decode_head.max_depth = case['min_depth'], case['max_depth']
        if 'scale_up' in case:
            m.decode_head.scale_up = case['scale_up']
        if 'use_eps' in case:
            m.decode_head.use_eps = case['use_eps']
        model = revert_sync_batchnorm(build_depther(m)).eval()
        assert hasattr(model.decode_head.encoder.layers[0], 'time_mlp'), 'time-aware layer not registered'
        sd = synthetic.make_state_dict('depth', 1, 6, 256, seed=case['seed'] + 100)
        load_hot_path(model, sd)
        x, noise = synthetic.make_inputs(1, case['h'], case['w'], case['randsteps'], 256, 1, seed=case['seed'])
        rec = dict(pred=[], feat=[])
        def grab(key):
            def hook(mod, i, o):
                rec[key].append(o.clone())
            return hook

        wrap_forward(model.decode_head, rec['pred'])
        hooks = [model.down.register_forward_hook(grab('feat'))]
        with RandnPatch(noise[0]):
            out = model.sample(x, None)
        for hk in hooks:
            hk.remove()
        save(case['name'], dict(task='depth', **case),
             dict(out=out, depth_pred_steps=torch.stack(rec['pred']), feat_step0=rec['feat'][0],
                  x_fp=fingerprint(x), noise_fp=fingerprint(noise), weights_fp=synthetic.checksum(sd)))


def gen_bev():
    import ref_shim
    ddp_mod, head_mod = ref_shim.import_bev()
    from mmcv.utils import ConfigDict
    for case in BEV_CASES:
        nl = case['num_layers']
        encoder = dict(
            type='DetrTransformerEncoder', num_layers=nl,
            transformerlayers=dict(
                type='BaseTransformerLayer', use_time_mlp=True,
                attn_cfgs=dict(type='MultiScaleDeformableAttention', embed_dims=256, num_levels=1,
                               num_heads=8, dropout=0.0),
                ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024, ffn_drop=0.,
                              act_cfg=dict(type='GELU')),
                operation_order=('self_attn', 'norm', 'ffn', 'norm')))
        head = head_mod.DeformableHeadWithTime(
            num_feature_levels=1, encoder=ConfigDict(encoder),
            positional_encoding=ConfigDict(dict(type='SinePositionalEncoding', num_feats=128, normalize=True, offset=-0.5)),
            classes=['a', 'b', 'c', 'd', 'e', 'f'], loss='focal',
            grid_transform=dict(input_scope=case['input_scope'], output_scope=case['output_scope']),
            in_channels=256).eval()
        model = ddp_mod.DDP(bit_scale=case['bit_scale'], timesteps=case['timesteps'], randsteps=case['randsteps'],
                            feat_channels=case['feat_channels']).eval()
        sd = synthetic.make_state_dict('bev', 6, nl, case['feat_channels'], seed=case['seed'] + 100)
        res = model.load_state_dict({k: v for k, v in sd.items() if not k.startswith('decode_head.')}, strict=False)
        assert not res.unexpected_keys and not res.missing_keys, res
        res = head.load_state_dict({k[len('decode_head.'):]: v for k, v in sd.items() if k.startswith('decode_head.')},
                                   strict=False)
        assert not res.unexpected_keys and not res.missing_keys, res
        assert hasattr(head.encoder.layers[0], 'time_mlp') and head.encoder.layers[0].time_mlp is not None
        x, noise = synthetic.make_inputs(1, case['h'], case['w'], case['randsteps'], case['feat_channels'], 256,
                                         seed=case['seed'])
        rec = dict(prob=[])
        def grab_prob(mod, i, o):
            rec['prob'].append(o.clone())

        hk = head.register_forward_hook(grab_prob)
        with RandnPatch(noise[0]):
            out = model.ddim_sample([x], head)
        hk.remove()
        save(case['name'], dict(task='bev', **case),
             dict(out=out, prob_steps=torch.stack(rec['prob']),
                  x_fp=fingerprint(x), noise_fp=fingerprint(noise), weights_fp=synthetic.checksum(sd)))


POST_CASES = [
    # network input == img_shape == ori_shape: one resize (x4), no flip
    dict(name='post_same', num_classes=19, h=16, w=24, img=(64, 96), img_shape=(64, 96), ori_shape=(64, 96), flip=None,
         align_corners=False, seed=0),
    # ADE-like: padded input, crop to img_shape, shrink to ori_shape
    dict(name='post_ade', num_classes=150, h=16, w=20, img=(64, 80), img_shape=(61, 77), ori_shape=(47, 59), flip=None,
         align_corners=False, seed=1),
    # odd map, enlarge to ori_shape, horizontal flip
    dict(name='post_hflip', num_classes=19, h=13, w=17, img=(52, 68), img_shape=(52, 68), ori_shape=(75, 101),
         flip='horizontal', align_corners=False, seed=2),
    # vertical flip + align_corners=True
    dict(name='post_vflip_ac', num_classes=19, h=9, w=11, img=(36, 44), img_shape=(33, 41), ori_shape=(40, 50),
         flip='vertical', align_corners=True, seed=3),
]


def gen_post():
    """Post-loop epilogue through the reference's own simple_test / inference / whole_inference /
    encode_decode; the backbone and the sampler are replaced by seeded scores."""
    import ref_shim
    build_segmentor, Config, revert = ref_shim.import_seg()
    cfg_path = os.path.join(ref_shim.REF, 'segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py')
    for case in POST_CASES:
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.test_cfg.mode = 'whole'
        m.decode_head.num_classes = case['num_classes']
        m.auxiliary_head.num_classes = case['num_classes']
        model = revert(build_segmentor(m)).eval()
        model.align_corners = case['align_corners']
        scores = synthetic.make_scores(1, case['num_classes'], case['h'], case['w'], case['seed'])
        model.extract_feat = lambda img: [None]
        model.ddim_sample = lambda x, img_metas: scores.clone()
        img = torch.zeros((1, 3) + tuple(case['img']))
        meta = [dict(img_shape=tuple(case['img_shape']) + (3,), ori_shape=tuple(case['ori_shape']) + (3,),
                     pad_shape=tuple(case['img']) + (3,), flip=case['flip'] is not None,
                     flip_direction=case['flip'] or 'horizontal')]
        seg = model.simple_test(img, meta, rescale=True)[0]
        prob = model.inference(img, meta, True)
        top2 = prob.topk(2, dim=1).values
        save(case['name'], dict(task='post', **case),
             dict(seg=seg.astype('uint8'), margin=(top2[:, 0] - top2[:, 1])[0], scores_fp=fingerprint(scores)))


AUG_CASES = [
    # multi-scale x horizontal flip on a Cityscapes-like head: 3 scales x {plain, flipped}; img_shape < padded input
    dict(name='aug_ms3_hflip', num_classes=19, ori_shape=(40, 50), align_corners=False, seed=0,
         augs=[dict(h=8, w=10, img=(32, 40), img_shape=(30, 38), flip=None), dict(h=8, w=10, img=(32, 40), img_shape=(30, 38), flip='horizontal'),
               dict(h=10, w=13, img=(40, 52), img_shape=(40, 50), flip=None), dict(h=10, w=13, img=(40, 52), img_shape=(40, 50), flip='horizontal'),
               dict(h=15, w=19, img=(60, 76), img_shape=(60, 75), flip=None), dict(h=15, w=19, img=(60, 76), img_shape=(60, 75), flip='horizontal')]),
    # ADE-like: 150 classes, flip only, no second resize for the plain augmentation (img_shape == ori_shape == input)
    dict(name='aug_ade_flip', num_classes=150, ori_shape=(48, 64), align_corners=False, seed=1,
         augs=[dict(h=12, w=16, img=(48, 64), img_shape=(48, 64), flip=None), dict(h=12, w=16, img=(48, 64), img_shape=(48, 64), flip='horizontal')]),
    # vertical flip + align_corners=True + odd sizes
    dict(name='aug_vflip_ac', num_classes=19, ori_shape=(37, 45), align_corners=True, seed=2,
         augs=[dict(h=9, w=11, img=(36, 44), img_shape=(33, 41), flip=None), dict(h=9, w=11, img=(36, 44), img_shape=(33, 41), flip='vertical'),
               dict(h=13, w=17, img=(52, 68), img_shape=(50, 66), flip='vertical')]),
]


def gen_aug():
    """Multi-scale / flip test-time augmentation (encoder_decoder.py:306-331) through the reference's own aug_test /
    inference / whole_inference / encode_decode; backbone and sampler replaced by seeded per-augmentation scores."""
    import ref_shim
    build_segmentor, Config, revert = ref_shim.import_seg()
    cfg_path = os.path.join(ref_shim.REF, 'segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py')
    for case in AUG_CASES:
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.test_cfg.mode = 'whole'
        m.decode_head.num_classes = case['num_classes']
        m.auxiliary_head.num_classes = case['num_classes']
        model = revert(build_segmentor(m)).eval()
        model.align_corners = case['align_corners']
        scores = [synthetic.make_scores(1, case['num_classes'], a['h'], a['w'], case['seed'] * 100 + i) for i, a in enumerate(case['augs'])]
        calls = []
        model.extract_feat = lambda img: [None]

        def sample(x, img_metas, _calls=calls, _scores=scores):
            _calls.append(1)
            return _scores[(len(_calls) - 1) % len(_scores)].clone()
        model.ddim_sample = sample
        imgs = [torch.zeros((1, 3) + tuple(a['img'])) for a in case['augs']]
        metas = [[dict(img_shape=tuple(a['img_shape']) + (3,), ori_shape=tuple(case['ori_shape']) + (3,),
                       pad_shape=tuple(a['img']) + (3,), flip=a['flip'] is not None,
                       flip_direction=a['flip'] or 'horizontal')] for a in case['augs']]
        seg = model.aug_test(imgs, metas, rescale=True)[0]
        # the mean probabilities the argmax was taken of (same calls again: the stub cycles through the scores)
        prob = model.inference(imgs[0], metas[0], True)
        for i in range(1, len(imgs)):
            prob += model.inference(imgs[i], metas[i], True)
        prob /= len(imgs)
        assert (prob.argmax(1)[0].numpy() == seg).all()
        top2 = prob.topk(2, dim=1).values
        save(case['name'], dict(task='aug', **case),
             dict(seg=seg.astype('uint8'), prob=prob[0], margin=(top2[:, 0] - top2[:, 1])[0],
                  scores_fp=np.stack([fingerprint(t) for t in scores])))


SLIDE_CASES = [
    # Cityscapes-like: 3 x 3 overlapping windows, image == img_shape == ori_shape (one resize per window, no second stage)
    dict(name='slide_city_3x3', num_classes=19, img=(64, 128), crop_size=(32, 64), stride=(21, 43), h=8, w=16, img_shape=(64, 128),
         ori_shape=(64, 128), flip=None, align_corners=False, seed=0),
    # padded input cropped to img_shape, enlarged to ori_shape, horizontal flip; last windows clamped back into the image
    dict(name='slide_crop_resize_hflip', num_classes=19, img=(48, 80), crop_size=(32, 32), stride=(20, 20), h=8, w=8, img_shape=(45, 77),
         ori_shape=(61, 99), flip='horizontal', align_corners=False, seed=1),
    # crop larger than the image in one dimension ("the small patch will be used"), 150 classes, align_corners, vertical flip
    dict(name='slide_small_image_ac', num_classes=150, img=(24, 72), crop_size=(32, 32), stride=(16, 24), h=6, w=8, img_shape=(24, 70),
         ori_shape=(30, 64), flip='vertical', align_corners=True, seed=2),
]


def gen_slide():
    """Sliding-window inference (encoder_decoder.py:180-227) through the reference's own simple_test / inference /
    slide_inference / DDP.encode_decode; backbone and sampler replaced by seeded low-resolution scores, one per window."""
    import ref_shim
    build_segmentor, Config, revert = ref_shim.import_seg()
    cfg_path = os.path.join(ref_shim.REF, 'segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py')
    for case in SLIDE_CASES:
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.test_cfg.mode = 'slide'
        m.test_cfg.crop_size = tuple(case['crop_size'])
        m.test_cfg.stride = tuple(case['stride'])
        m.decode_head.num_classes = case['num_classes']
        m.auxiliary_head.num_classes = case['num_classes']
        model = revert(build_segmentor(m)).eval()
        model.align_corners = case['align_corners']
        # the window grid is RECORDED from the reference (nothing of the product is imported here): the image carries its own
        # pixel coordinates in channels 0 / 1, so every crop slide_inference hands to encode_decode -> extract_feat tells where
        # it was cut (y1, x1) and how large it is; window i gets the seeded scores i (one sampler call per window)
        H, W = case['img']
        img = torch.zeros((1, 3, H, W))
        img[0, 0] = torch.arange(H, dtype=torch.float32).view(H, 1)
        img[0, 1] = torch.arange(W, dtype=torch.float32).view(1, W)
        windows, calls, state = [], [], dict(n=None)

        def extract(im, _w=windows):
            y1, x1 = int(im[0, 0, 0, 0]), int(im[0, 1, 0, 0])
            _w.append((y1, x1, y1 + im.shape[2], x1 + im.shape[3]))
            return [None]
        model.extract_feat = extract

        def sample(x, img_metas, _calls=calls):
            i = len(_calls) if state['n'] is None else len(_calls) % state['n']
            _calls.append(1)
            return synthetic.make_scores(1, case['num_classes'], case['h'], case['w'], case['seed'] * 100 + i)
        model.ddim_sample = sample
        meta = [dict(img_shape=tuple(case['img_shape']) + (3,), ori_shape=tuple(case['ori_shape']) + (3,),
                     pad_shape=tuple(case['img']) + (3,), flip=case['flip'] is not None, flip_direction=case['flip'] or 'horizontal')]
        seg = model.simple_test(img, meta, rescale=True)[0]
        n_win = state['n'] = len(calls)
        win = list(windows)
        assert len(win) == n_win
        prob = model.inference(img, meta, True)
        assert windows[n_win:] == win and (prob.argmax(1)[0].numpy() == seg).all()
        scores = [synthetic.make_scores(1, case['num_classes'], case['h'], case['w'], case['seed'] * 100 + i) for i in range(n_win)]
        top2 = prob.topk(2, dim=1).values
        save(case['name'], dict(task='slide', n_windows=n_win, **case),
             dict(seg=seg.astype('uint8'), prob=prob[0], margin=(top2[:, 0] - top2[:, 1])[0], scores_fp=np.stack([fingerprint(t) for t in scores]),
                  windows=np.asarray(win, dtype=np.int32)))


DPOST_CASES = [
    # KITTI test pipeline (depth/configs/_base_/datasets/kitti.py:26-41): 352 x 1216 after KBCrop, plain + horizontal flip
    dict(name='dpost_kitti_flip', batch=1, img=(352, 1216), align_corners=False, rescale=True, min_depth=1e-3, max_depth=80.0, seed=0,
         augs=[dict(h=88, w=304, flip=None), dict(h=88, w=304, flip='horizontal')]),
    # one augmentation = simple_test; odd map, input not 4x the map (Swin pads), horizontal flip
    dict(name='dpost_simple_hflip', batch=1, img=(53, 70), align_corners=False, rescale=True, min_depth=1e-3, max_depth=80.0, seed=1,
         augs=[dict(h=13, w=17, flip='horizontal')]),
    # vertical flip, align_corners=True, three augmentations, maps of different sizes, NYU depth range
    dict(name='dpost_vflip_ac3', batch=1, img=(41, 47), align_corners=True, rescale=True, min_depth=1e-3, max_depth=10.0, seed=2,
         augs=[dict(h=11, w=12, flip=None), dict(h=11, w=12, flip='vertical'), dict(h=9, w=15, flip='horizontal')]),
    # rescale=False: clamp + flip at the map's own size
    dict(name='dpost_norescale', batch=1, img=(36, 44), align_corners=False, rescale=False, min_depth=1e-3, max_depth=80.0, seed=3,
         augs=[dict(h=9, w=11, flip='horizontal')]),
]


def gen_dpost():
    """Depth toolbox test entry (SURVEY.md §8 b, depth line) through the reference's OWN call chain: ``model(return_loss=False,
    img=[...], img_metas=[[...]])`` (depth/depth/apis/test.py:88) -> base.py ``forward`` -> ``forward_test`` -> ``simple_test`` /
    ``aug_test`` -> ``inference`` -> ``whole_inference`` -> ``DDP.encode_decode`` (clamp + resize); the backbone and the sampler
    are replaced by seeded low-resolution maps."""
    import ref_shim
    build_depther, Config = ref_shim.import_depth()
    from mmcv.cnn.utils import revert_sync_batchnorm
    cfg_path = os.path.join(ref_shim.REF, 'depth/configs/ddp_kitti/ddp_swint_1k_w7_kitti_bs2x8_scale01.py')
    for case in DPOST_CASES:
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.min_depth = m.decode_head.min_depth = case['min_depth']
        m.max_depth = m.decode_head.max_depth = case['max_depth']
        model = revert_sync_batchnorm(build_depther(m)).eval()
        model.align_corners = case['align_corners']
        maps = [synthetic.make_depth_map(case['batch'], a['h'], a['w'], case['seed'] * 100 + i) for i, a in enumerate(case['augs'])]
        calls = []
        model.extract_feat = lambda img: [None]

        def sample(x, img_metas, _calls=calls, _maps=maps):
            _calls.append(1)
            return _maps[(len(_calls) - 1) % len(_maps)].clone()
        model.sample = sample
        H, W = case['img']
        imgs = [torch.zeros((case['batch'], 3, H, W)) for _ in case['augs']]
        metas = [[dict(img_shape=(H, W, 3), ori_shape=(H, W, 3), pad_shape=(H, W, 3), flip=a['flip'] is not None,
                       flip_direction=a['flip'] or 'horizontal')] * case['batch'] for a in case['augs']]
        if case['rescale']:
            res = model(return_loss=False, img=imgs, img_metas=metas)               # what apis/test.py calls
        else:
            res = model(return_loss=False, img=imgs, img_metas=metas, rescale=False)
        assert isinstance(res, list) and len(res) == case['batch'] and len(calls) == len(case['augs'])
        out = torch.from_numpy(np.stack(res))                                       # (B,1,H,W)
        save(case['name'], dict(task='dpost', **case),
             dict(out=out, maps_fp=np.stack([fingerprint(t) for t in maps])))


NECK_CASES = [
    dict(name='neck_msm_even', batch=2, h=16, w=24, align_corners=False, seed=0),
    dict(name='neck_msm_odd', batch=1, h=13, w=19, align_corners=False, seed=1),
    dict(name='neck_msm_ac', batch=1, h=12, w=20, align_corners=True, seed=2),
]


def gen_neck():
    """MultiStageMerging from the reference class itself."""
    import ref_shim
    ref_shim.import_seg()
    from mmseg.models.necks import MultiStageMerging
    for case in NECK_CASES:
        neck = MultiStageMerging([256] * 4, 256, kernel_size=1, norm_cfg=dict(type='GN', num_groups=32), act_cfg=None,
                                 align_corners=case['align_corners']).eval()
        sd = synthetic.make_neck_state_dict(case['seed'])
        neck.load_state_dict(sd, strict=True)
        levels = synthetic.make_levels(case['batch'], case['h'], case['w'], case['seed'])
        out = neck(levels)[0]
        save(case['name'], dict(task='neck', **case),
             dict(out=out, levels_fp=np.array([fingerprint(t) for t in levels]), weights_fp=synthetic.checksum(sd)))


FCN_CASES = [
    dict(name='fcn_bn_2conv', num_convs=2, num_classes=19, with_norm=True, concat_input=True, dilation=1, maps=2, h=12, w=20,
         seed=0, with_time=True),
    dict(name='fcn_nonorm_1conv', num_convs=1, num_classes=150, with_norm=False, concat_input=False, dilation=1, maps=1, h=13,
         w=17, seed=1, with_time=True),
    dict(name='fcn_bn_dil2_notime', num_convs=2, num_classes=19, with_norm=True, concat_input=False, dilation=2, maps=1, h=10,
         w=14, seed=2, with_time=False),
]


def gen_fcn():
    """FCNHeadWithTime from the reference class itself, eval mode."""
    import ref_shim
    ref_shim.import_seg()
    from mmseg.models.decode_heads import FCNHeadWithTime
    for case in FCN_CASES:
        head = FCNHeadWithTime(num_convs=case['num_convs'], kernel_size=3, concat_input=case['concat_input'],
                               dilation=case['dilation'], in_channels=256, channels=256, num_classes=case['num_classes'],
                               in_index=0, dropout_ratio=0.1, norm_cfg=dict(type='BN') if case['with_norm'] else None,
                               align_corners=False).eval()
        sd = synthetic.make_fcn_state_dict(case['num_convs'], case['num_classes'], case['with_norm'], case['concat_input'],
                                           case['seed'])
        head.load_state_dict(sd, strict=True)
        feat, temb = synthetic.make_fcn_inputs(case['maps'], case['h'], case['w'], case['seed'])
        times = temb.expand(case['maps'], 1024) if case['with_time'] else None
        out = head([feat], times)
        save(case['name'], dict(task='fcn', **case),
             dict(out=out, feat_fp=fingerprint(feat), weights_fp=synthetic.checksum(sd)))


FPN_CASES = [
    dict(name='fpn_swin_t', in_channels=[96, 192, 384, 768], batch=2, h=16, w=24, seed=0),      # configs/ade/ddp_swin_t...:41-46
    dict(name='fpn_odd', in_channels=[128, 256, 512, 1024], batch=1, h=13, w=19, seed=1),       # odd sizes: 13,7,4,2 rows
]


def gen_fpn():
    """FPN from the reference class itself."""
    import ref_shim
    ref_shim.import_seg()
    from mmseg.models.necks import FPN
    for case in FPN_CASES:
        neck = FPN(in_channels=case['in_channels'], out_channels=256, act_cfg=None, norm_cfg=dict(type='GN', num_groups=32),
                   num_outs=4).eval()
        sd = synthetic.make_fpn_state_dict(case['in_channels'], case['seed'])
        neck.load_state_dict(sd, strict=True)
        levels = synthetic.make_backbone_levels(case['batch'], case['in_channels'], case['h'], case['w'], case['seed'])
        outs = neck(levels)
        arrays = {f'out{l}': o for l, o in enumerate(outs)}
        arrays.update(levels_fp=np.array([fingerprint(t) for t in levels]), weights_fp=synthetic.checksum(sd))
        save(case['name'], dict(task='fpn', **case), arrays)


ALIGNED_CASES = [
    dict(name='aligned_city', h=12, w=18, num_classes=19, batch=2, bit_scale=0.01, noise_schedule='cosine', seed=30),
    dict(name='aligned_ade_linear', h=9, w=13, num_classes=150, batch=1, bit_scale=0.1, noise_schedule='linear', seed=31),
]


class _Stop(Exception):
    pass


def gen_aligned():
    """The self-aligned pre-pass of SelfAlignedDDP.forward_train (segmentors/self_aligned_ddp.py:150-164), recorded from the
    reference class itself: forward_train is entered with the backbone replaced by the seeded feature, ``torch.randn_like``
    by the seeded noise, and is left (sentinel exception) right after ``embedding_table(preds)`` - the rest of the method is
    the training loss."""
    import ref_shim
    build_segmentor, Config, revert = ref_shim.import_seg()
    cfg_path = os.path.join(ref_shim.REF, 'segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py')
    for case in ALIGNED_CASES:
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.type = 'SelfAlignedDDP'             # same kwargs (the *_aligned Cityscapes configs need mmcls for their backbone)
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.bit_scale = case['bit_scale']
        m.noise_schedule = case['noise_schedule']
        m.decode_head.num_classes = case['num_classes']
        m.auxiliary_head.num_classes = case['num_classes']
        model = revert(build_segmentor(m)).eval()
        assert type(model).__name__ == 'SelfAlignedDDP'
        sd = synthetic.make_state_dict('seg', case['num_classes'], 6, 256, seed=case['seed'] + 100)
        load_hot_path(model, sd)
        x, noise = synthetic.make_inputs(case['batch'], case['h'], case['w'], 1, 256, 256, seed=case['seed'])
        noise = noise[:, 0]                                   # (b,256,h,w) = randn_like(x)
        rec = dict(logits=[], emb=[])
        wrap_forward(model.decode_head, rec['logits'])

        def grab_emb(mod, i, o):
            rec['emb'].append(o.clone())
            raise _Stop()
        hk = model.embedding_table.register_forward_hook(grab_emb)
        model.extract_feat = lambda img: [x]
        img = torch.zeros(case['batch'], 3, 4 * case['h'], 4 * case['w'])
        try:
            with RandnPatch(None, [noise]):
                model.forward_train(img, [dict()] * case['batch'], torch.zeros(case['batch'], 1, 4 * case['h'], 4 * case['w'],
                                                                               dtype=torch.long))
        except _Stop:
            pass
        hk.remove()
        e = rec['emb'][0]                                     # (b,h,w,256): embedding_table(argmax(logits))
        preds = (torch.sigmoid(e.squeeze(1).permute(0, 3, 1, 2)) * 2 - 1) * case['bit_scale']     # :163-164
        save(case['name'], dict(task='aligned', **case),
             dict(preds=preds, logits=rec['logits'][0], x_fp=fingerprint(x), noise_fp=fingerprint(noise),
                  weights_fp=synthetic.checksum(sd)))


LOOPFCN_CASES = [
    dict(name='loopfcn_bn_k3', h=11, w=15, num_classes=19, timesteps=3, randsteps=1, bit_scale=0.01, accumulation=True,
         diffusion='ddim', num_convs=2, with_norm=True, concat_input=False, dilation=1, seed=40),
    dict(name='loopfcn_nonorm_r2', h=8, w=13, num_classes=150, timesteps=2, randsteps=2, bit_scale=0.01, accumulation=False,
         diffusion='ddim', num_convs=1, with_norm=False, concat_input=True, dilation=1, seed=41),
    dict(name='loopfcn_ddpm', h=9, w=10, num_classes=19, timesteps=3, randsteps=1, bit_scale=0.01, accumulation=True,
         diffusion='ddpm', num_convs=2, with_norm=True, concat_input=False, dilation=2, seed=42),
]


def gen_loopfcn():
    """The reference sampler loop (segmentors/ddp.py:215-290) driving the reference's FCNHeadWithTime
    (decode_heads/fcn_head_with_time.py:228-343) as decode head.  No shipped config pairs them and the segmentor's
    constructor reads ``decode_head.in_channels[0]`` (ddp.py:78), which FCNHeadWithTime's int ``in_channels`` does not
    support - so the segmentor is built from the ADE config and its decode head is then REPLACED by the reference FCN
    head; ``_decode_head_forward_test`` (:192-196) only needs ``forward_test(inputs, times, img_metas, test_cfg)``."""
    import ref_shim
    build_segmentor, Config, revert = ref_shim.import_seg()
    from mmseg.models.decode_heads import FCNHeadWithTime
    cfg_path = os.path.join(ref_shim.REF, 'segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py')
    for case in LOOPFCN_CASES:
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.timesteps, m.randsteps, m.bit_scale = case['timesteps'], case['randsteps'], case['bit_scale']
        m.accumulation, m.diffusion = case['accumulation'], case['diffusion']
        m.decode_head.num_classes = case['num_classes']
        m.auxiliary_head.num_classes = case['num_classes']
        model = revert(build_segmentor(m)).eval()
        model.decode_head = FCNHeadWithTime(num_convs=case['num_convs'], kernel_size=3, concat_input=case['concat_input'],
                                            dilation=case['dilation'], in_channels=256, channels=256,
                                            num_classes=case['num_classes'], in_index=0, dropout_ratio=0.1,
                                            norm_cfg=dict(type='BN') if case['with_norm'] else None, align_corners=False).eval()
        # the sampler reads ``decode_head.in_channels[0]`` for the shape of the start noise (ddp.py:220,252); the FCN head
        # stores an int and no longer reads it after construction
        model.decode_head.in_channels = [256]
        sd = synthetic.make_fcn_segmentor_state_dict(case['num_convs'], case['num_classes'], case['with_norm'],
                                                     case['concat_input'], case['seed'] + 100)
        load_hot_path(model, sd)
        x, noise = synthetic.make_inputs(1, case['h'], case['w'], case['randsteps'], 256, 256, seed=case['seed'])
        step_noise = None
        if case['diffusion'] == 'ddpm':
            g = torch.Generator().manual_seed(case['seed'] + 7)
            step_noise = torch.randn((case['timesteps'], case['randsteps'], 256, case['h'], case['w']), generator=g)
        logits = []
        wrap_forward(model.decode_head, logits)
        with RandnPatch(noise[0], None if step_noise is None else list(step_noise)):
            out = model.ddim_sample(x, None) if case['diffusion'] == 'ddim' else model.ddpm_sample(x, None)
        save(case['name'], dict(task='loopfcn', **case),
             dict(out=out, logits_steps=torch.stack(logits), x_fp=fingerprint(x), noise_fp=fingerprint(noise),
                  weights_fp=synthetic.checksum(sd)))


# ---- full-size fixtures (VERDICT r04 "next" #1): the REFERENCE itself at the sizes of BASELINE.json's configurations, one image each.
# Inputs / weights = the seeds tests/test_full_size_parity.py uses for the batch of that configuration (image ``b`` of a
# ``B``-image draw), so the GPU test runs the engine once on the whole batch and compares image b.  Stored compactly:
#   seg  : per-step argmax decisions (uint8), per-step top-2 gap (fp16, clipped at 1e-2 of the score scale - only small gaps
#          matter), per-step score scale, the final class map, the final scores on every ``stride``-th pixel (all classes,
#          fp32) and a seeded class-weighted projection of the final scores at EVERY pixel (one fp32 per pixel);
#   depth: the final map and the per-step predictions on every 16th pixel;   bev: the final map (r = 4: every 2nd pixel).
FULLSIZE_SEG = [
    dict(name='full_c1', B=1, b=0, h=128, w=128, num_classes=150, timesteps=1, accumulation=True, sd_seed=2, in_seed=10, stride=11),
    dict(name='full_c2', B=8, b=0, h=128, w=256, num_classes=150, timesteps=3, accumulation=True, sd_seed=2, in_seed=0, stride=11),
    dict(name='full_c3', B=4, b=2, h=256, w=512, num_classes=19, timesteps=10, accumulation=False, sd_seed=3, in_seed=30, stride=7),
    # round 6: a SECOND image of the C2 and C3 batches, and the content-dependent 'trained_like' weight
    # profile at C2 size (the seeds of tests/test_full_size_parity.py::test_c2_size_trained_like_weights, image 1 of its 2-image draw)
    dict(name='full_c2_b5', B=8, b=5, h=128, w=256, num_classes=150, timesteps=3, accumulation=True, sd_seed=2, in_seed=0, stride=11),
    dict(name='full_c3_b0', B=4, b=0, h=256, w=512, num_classes=19, timesteps=10, accumulation=False, sd_seed=3, in_seed=30, stride=7),
    dict(name='full_c2_trained', B=2, b=1, h=128, w=256, num_classes=150, timesteps=3, accumulation=True, sd_seed=7, in_seed=4, stride=11,
         profile='trained_like'),
]
FULLSIZE_DEPTH = [
    dict(name='full_c4', B=16, b=0, h=88, w=304, timesteps=20, sd_seed=4, in_seed=40, bit_scale=0.1, min_depth=1e-3, max_depth=80.0),
]
_BEV_SCOPES = dict(input_scope=[[-51.2, 51.2, 0.8], [-51.2, 51.2, 0.8]], output_scope=[[-50, 50, 0.5], [-50, 50, 0.5]])
FULLSIZE_BEV = [
    # the shape of BASELINE.json configs[4] (fusion features 512 ch at 128^2 -> 200^2 decoder grid, 5 layers, 6 classes, K = 3)
    dict(name='full_c5_r1', B=8, b=0, h=128, w=128, feat_channels=512, timesteps=3, randsteps=1, bit_scale=0.01, num_layers=5,
         sd_seed=5, in_seed=50, stride=1, **_BEV_SCOPES),
    # the SHIPPED sampler setting: bev/configs/nuscenes/seg/ddp-fusion-bev256d2-lss-scale001-d5-lr5e-5.yaml:5  randsteps: 4
    dict(name='full_c5_r4', B=2, b=1, h=128, w=128, feat_channels=512, timesteps=3, randsteps=4, bit_scale=0.01, num_layers=5,
         sd_seed=5, in_seed=52, stride=2, **_BEV_SCOPES),
]


def gen_fullsize_seg():
    import ref_shim
    build_segmentor, Config, revert = ref_shim.import_seg()
    cfg_path = os.path.join(ref_shim.REF, 'segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py')
    for case in FULLSIZE_SEG:
        if ONLY and case['name'] not in ONLY:
            continue
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.timesteps, m.randsteps, m.bit_scale, m.accumulation = case['timesteps'], 1, 0.01, case['accumulation']
        m.decode_head.num_classes = case['num_classes']
        m.auxiliary_head.num_classes = case['num_classes']
        model = revert(build_segmentor(m)).eval()
        sd = synthetic.make_state_dict('seg', case['num_classes'], 6, 256, seed=case['sd_seed'], profile=case.get('profile', 'init'))
        load_hot_path(model, sd)
        xb, nb = synthetic.make_inputs(case['B'], case['h'], case['w'], 1, 256, 256, seed=case['in_seed'])
        b = case['b']
        x, noise = xb[b:b + 1].clone(), nb[b].clone()
        del xb, nb
        logits = []
        wrap_forward(model.decode_head, logits)
        with RandnPatch(noise):
            out = model.ddim_sample(x, None)
        K = case['timesteps']
        assert len(logits) == K
        dec = torch.stack([lg.argmax(1)[0] for lg in logits]).to(torch.uint8)                       # (K,h,w)
        scale = torch.tensor([float(lg.abs().max()) for lg in logits])
        gaps = []
        for s, lg in enumerate(logits):
            t2 = lg.topk(2, dim=1).values
            gaps.append((t2[:, 0] - t2[:, 1])[0].clamp(max=1e-2 * float(scale[s])))
        gap = torch.stack(gaps).to(torch.float16)
        flat = out[0].reshape(case['num_classes'], -1)
        idx = torch.arange(0, flat.shape[1], case['stride'])
        proj = (out[0] * class_projection_weights(case['num_classes']).view(-1, 1, 1)).sum(0)
        save(case['name'], dict(task='fullsize_seg', **case),
             dict(decisions=dec, gap=gap, score_scale=scale, final_cls=out[0].argmax(0).to(torch.uint8),
                  out_sub=flat[:, idx].contiguous(), out_absmax=float(out.abs().max()), out_proj=proj,
                  x_fp=fingerprint(x), noise_fp=fingerprint(noise), weights_fp=synthetic.checksum(sd)))


def gen_fullsize_depth():
    import ref_shim
    build_depther, Config = ref_shim.import_depth()
    from mmcv.cnn.utils import revert_sync_batchnorm
    cfg_path = os.path.join(ref_shim.REF, 'depth/configs/ddp_kitti/ddp_swint_1k_w7_kitti_bs2x8_scale01.py')
    for case in FULLSIZE_DEPTH:
        cfg = Config.fromfile(cfg_path)
        m = cfg.model
        m.backbone.init_cfg = None
        m.train_cfg = None
        m.timesteps, m.randsteps, m.bit_scale = case['timesteps'], 1, case['bit_scale']
        m.min_depth, m.max_depth = case['min_depth'], case['max_depth']
        model = revert_sync_batchnorm(build_depther(m)).eval()
        assert hasattr(model.decode_head.encoder.layers[0], 'time_mlp'), 'time-aware layer not registered'
        sd = synthetic.make_state_dict('depth', 1, 6, 256, seed=case['sd_seed'])
        load_hot_path(model, sd)
        xb, nb = synthetic.make_inputs(case['B'], case['h'], case['w'], 1, 256, 1, seed=case['in_seed'])
        x, noise = xb[case['b']:case['b'] + 1].clone(), nb[case['b']].clone()
        preds = []
        wrap_forward(model.decode_head, preds)
        with RandnPatch(noise):
            out = model.sample(x, None)
        ps = torch.stack([p.reshape(-1)[::16] for p in preds])
        save(case['name'], dict(task='fullsize_depth', **case),
             dict(out=out, pred_steps_sub=ps, x_fp=fingerprint(x), noise_fp=fingerprint(noise), weights_fp=synthetic.checksum(sd)))


def gen_fullsize_bev():
    import ref_shim
    ddp_mod, head_mod = ref_shim.import_bev()
    from mmcv.utils import ConfigDict
    for case in FULLSIZE_BEV:
        nl = case['num_layers']
        encoder = dict(
            type='DetrTransformerEncoder', num_layers=nl,
            transformerlayers=dict(
                type='BaseTransformerLayer', use_time_mlp=True,
                attn_cfgs=dict(type='MultiScaleDeformableAttention', embed_dims=256, num_levels=1, num_heads=8, dropout=0.0),
                ffn_cfgs=dict(type='FFN', embed_dims=256, feedforward_channels=1024, ffn_drop=0., act_cfg=dict(type='GELU')),
                operation_order=('self_attn', 'norm', 'ffn', 'norm')))
        head = head_mod.DeformableHeadWithTime(
            num_feature_levels=1, encoder=ConfigDict(encoder),
            positional_encoding=ConfigDict(dict(type='SinePositionalEncoding', num_feats=128, normalize=True, offset=-0.5)),
            classes=['a', 'b', 'c', 'd', 'e', 'f'], loss='focal',
            grid_transform=dict(input_scope=case['input_scope'], output_scope=case['output_scope']), in_channels=256).eval()
        model = ddp_mod.DDP(bit_scale=case['bit_scale'], timesteps=case['timesteps'], randsteps=case['randsteps'],
                            feat_channels=case['feat_channels']).eval()
        sd = synthetic.make_state_dict('bev', 6, nl, case['feat_channels'], seed=case['sd_seed'])
        res = model.load_state_dict({k: v for k, v in sd.items() if not k.startswith('decode_head.')}, strict=False)
        assert not res.unexpected_keys and not res.missing_keys, res
        res = head.load_state_dict({k[len('decode_head.'):]: v for k, v in sd.items() if k.startswith('decode_head.')}, strict=False)
        assert not res.unexpected_keys and not res.missing_keys, res
        xb, nb = synthetic.make_inputs(case['B'], case['h'], case['w'], case['randsteps'], case['feat_channels'], 256, seed=case['in_seed'])
        x, noise = xb[case['b']:case['b'] + 1].clone(), nb[case['b']].clone()
        del xb, nb
        probs = []
        hk = head.register_forward_hook(lambda mod, i, o: probs.append(o.clone()))
        with RandnPatch(noise):
            out = model.ddim_sample([x], head)
        hk.remove()
        # how close the fed-back thresholded maps come to a tie: smallest |prob - 0.5| per step
        margin = torch.tensor([float((p - 0.5).abs().min()) for p in probs])
        st = case['stride']
        save(case['name'], dict(task='fullsize_bev', **case),
             dict(out_sub=out[:, :, ::st, ::st].contiguous(), out_absmax=float(out.abs().max()), out_mean=out.mean(1)[0],
                  thr_margin=margin, x_fp=fingerprint(x), noise_fp=fingerprint(noise), weights_fp=synthetic.checksum(sd)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--task', choices=['seg', 'depth', 'bev', 'post', 'neck', 'fcn', 'fpn', 'aligned', 'loopfcn', 'aug', 'dpost', 'slide', 'fullsize',
                                      'fullsize_seg', 'fullsize_depth', 'fullsize_bev', 'all'], default='all')
    ap.add_argument('--only', default='', help='comma-separated fixture names: regenerate just these (seg / fullsize_seg tasks)')
    args = ap.parse_args()
    global ONLY
    ONLY = set(n for n in args.only.split(',') if n)
    torch.set_num_threads(8)
    if args.task == 'all':
        for t in ('seg', 'depth', 'bev', 'post', 'neck', 'fcn', 'fpn', 'aligned', 'loopfcn', 'aug', 'dpost', 'slide'):       # separate processes: the trees' registries collide
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--task', t])
        return
    if args.task == 'fullsize':       # not part of 'all': ~6 CPU-minutes, the reference at BASELINE.json's sizes
        for t in ('fullsize_seg', 'fullsize_depth', 'fullsize_bev'):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--task', t])
        return
    with torch.no_grad():
        {'seg': gen_seg, 'depth': gen_depth, 'bev': gen_bev, 'post': gen_post, 'neck': gen_neck, 'fcn': gen_fcn, 'fpn': gen_fpn,
         'aligned': gen_aligned, 'loopfcn': gen_loopfcn, 'aug': gen_aug, 'dpost': gen_dpost, 'slide': gen_slide,
         'fullsize_seg': gen_fullsize_seg, 'fullsize_depth': gen_fullsize_depth, 'fullsize_bev': gen_fullsize_bev}[args.task]()


if __name__ == '__main__':
    main()
