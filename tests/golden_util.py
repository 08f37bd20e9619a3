"""Helpers shared by the oracle and HIP parity tests: load a golden fixture and regenerate its
seeded inputs / weights (checking the stored fingerprints)."""
import glob
import json
import os

import numpy as np
import torch

from ddp_amd.utils import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def case_names(task=None):
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))
    if task is not None:
        names = [n for n in names if n.startswith(task)]
    else:
        names = [n for n in names if not n.startswith(('post', 'neck', 'fcn', 'fpn', 'aligned', 'loopfcn', 'aug', 'dpost', 'slide', 'full'))]   # other rows: load_post_case / ...
    return names


def fingerprint(t):
    t = t.double()
    return [float(t.sum()), float(t.abs().sum())]


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != 'config'}
    task = cfg['task']
    nl = cfg.get('num_layers', 6)
    cx = cfg.get('feat_channels', 256)
    ncls = cfg.get('num_classes', 6 if task == 'bev' else 1)
    sd = synthetic.make_state_dict(task, ncls, nl, cx, seed=cfg['seed'] + 100, profile=cfg.get('profile', 'init'))
    cm = 1 if task == 'depth' else 256
    x, noise = synthetic.make_inputs(1, cfg['h'], cfg['w'], cfg['randsteps'], cx, cm, seed=cfg['seed'])
    # the seeded generators must reproduce exactly what the reference was fed
    assert abs(synthetic.checksum(sd) - float(arrays['weights_fp'])) <= 1e-9 * abs(float(arrays['weights_fp']))
    assert np.allclose(fingerprint(x), arrays['x_fp'].numpy(), rtol=1e-12)
    assert np.allclose(fingerprint(noise), arrays['noise_fp'].numpy(), rtol=1e-12)
    step_noise = None
    if cfg.get('diffusion') == 'ddpm':
        g = torch.Generator().manual_seed(cfg['seed'] + 7)
        step_noise = torch.randn((cfg['timesteps'], cfg['randsteps'], 256, cfg['h'], cfg['w']), generator=g)
    return cfg, sd, x, noise[0], step_noise, arrays


def class_projection_weights(num_classes, seed=1234):
    """seeded positive per-class weights of the every-pixel projection stored in the full-size seg fixtures (``out_proj``)"""
    g = torch.Generator().manual_seed(seed)
    return torch.rand((num_classes,), generator=g, dtype=torch.float32) + 0.5


FULLSIZE_TASKS = {'fullsize_seg': 'seg', 'fullsize_depth': 'depth', 'fullsize_bev': 'bev'}


def load_fullsize_case(name):
    """Full-size fixture made by the REFERENCE at a BASELINE.json configuration (tests/golden/gen_golden.py --task fullsize):
    -> (cfg, sd, x (B,Cx,h,w), noise (B,r,Cm,h,w), arrays).  The reference ran image ``cfg['b']`` of this seeded batch; the
    stored fingerprints are of that image."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != 'config'}
    task = FULLSIZE_TASKS[cfg['task']]
    cx = cfg.get('feat_channels', 256)
    sd = synthetic.make_state_dict(task, cfg.get('num_classes', 6 if task == 'bev' else 1), cfg.get('num_layers', 6), cx, seed=cfg['sd_seed'],
                                 profile=cfg.get('profile', 'init'))
    x, noise = synthetic.make_inputs(cfg['B'], cfg['h'], cfg['w'], cfg.get('randsteps', 1), cx, 1 if task == 'depth' else 256, seed=cfg['in_seed'])
    b = cfg['b']
    assert abs(synthetic.checksum(sd) - float(arrays['weights_fp'])) <= 1e-9 * abs(float(arrays['weights_fp']))
    assert np.allclose(fingerprint(x[b:b + 1]), arrays['x_fp'].numpy(), rtol=1e-12)
    assert np.allclose(fingerprint(noise[b]), arrays['noise_fp'].numpy(), rtol=1e-12)
    return cfg, sd, x, noise, arrays


def max_rel(a, b):
    """max |a-b| / max |b|  (the parity metric of SURVEY.md §8d)."""
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def load_post_case(name):
    """Post-loop epilogue fixture: -> (cfg, scores (1,K,h,w), seg uint8 (oh,ow), top-2 margin)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    scores = synthetic.make_scores(1, cfg['num_classes'], cfg['h'], cfg['w'], cfg['seed'])
    assert np.allclose(fingerprint(scores), z['scores_fp'], rtol=1e-12)
    return cfg, scores, torch.from_numpy(z['seg']), torch.from_numpy(z['margin'])


def load_aug_case(name):
    """Test-time-augmentation fixture (reference ``aug_test``): -> (cfg, [scores_i (1,K,h_i,w_i)], metas, seg uint8, prob, margin)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    scores = [synthetic.make_scores(1, cfg['num_classes'], a['h'], a['w'], cfg['seed'] * 100 + i) for i, a in enumerate(cfg['augs'])]
    assert np.allclose(np.array([fingerprint(t) for t in scores]), z['scores_fp'], rtol=1e-12)
    metas = [dict(img_size=tuple(a['img']), crop_size=tuple(a['img_shape']), flip=a['flip']) for a in cfg['augs']]
    return cfg, scores, metas, torch.from_numpy(z['seg']), torch.from_numpy(z['prob']), torch.from_numpy(z['margin'])


def load_slide_case(name):
    """Sliding-window fixture (reference ``simple_test`` / ``inference`` with test_cfg.mode='slide'): -> (cfg, [window scores
    (1,K,h,w)] row-major over the grid, seg uint8 (oh,ow), prob (K,oh,ow), margin (oh,ow)).  cfg['windows'] = the (y1,x1,y2,x2)
    list the REFERENCE's slide_inference cut (recorded by the generator from the crops it handed to extract_feat), call order."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    cfg['windows'] = [tuple(int(v) for v in r) for r in z['windows']]
    scores = [synthetic.make_scores(1, cfg['num_classes'], cfg['h'], cfg['w'], cfg['seed'] * 100 + i) for i in range(cfg['n_windows'])]
    assert np.allclose(np.array([fingerprint(t) for t in scores]), z['scores_fp'], rtol=1e-12)
    return cfg, scores, torch.from_numpy(z['seg']), torch.from_numpy(z['prob']), torch.from_numpy(z['margin'])


def reference_window_grid(cfg):
    """(ys, xs, crop) of a slide fixture from the window list the REFERENCE cut (row-major: all columns of a row first)."""
    win = cfg['windows']
    ys = sorted({w[0] for w in win}, key=[w[0] for w in win].index)
    xs = sorted({w[1] for w in win}, key=[w[1] for w in win].index)
    crop = (win[0][2] - win[0][0], win[0][3] - win[0][1])
    assert win == [(y, x, y + crop[0], x + crop[1]) for y in ys for x in xs], 'reference windows are not a row-major grid'
    return ys, xs, crop


def load_dpost_case(name):
    """Depth epilogue fixture (reference ``simple_test`` / ``aug_test`` / ``model(return_loss=False, **data)`` of the depth
    toolbox): -> (cfg, [depth_i (B,1,h,w)], flips, out (B,1,H,W))."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    maps = [synthetic.make_depth_map(cfg['batch'], a['h'], a['w'], cfg['seed'] * 100 + i) for i, a in enumerate(cfg['augs'])]
    assert np.allclose(np.array([fingerprint(t) for t in maps]), z['maps_fp'], rtol=1e-12)
    return cfg, maps, [a['flip'] for a in cfg['augs']], torch.from_numpy(z['out'])


def load_neck_case(name):
    """MultiStageMerging fixture: -> (cfg, levels, state_dict, out)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    sd = synthetic.make_neck_state_dict(cfg['seed'])
    levels = synthetic.make_levels(cfg['batch'], cfg['h'], cfg['w'], cfg['seed'])
    assert abs(synthetic.checksum(sd) - float(z['weights_fp'])) <= 1e-9 * abs(float(z['weights_fp']))
    assert np.allclose(np.array([fingerprint(t) for t in levels]), z['levels_fp'], rtol=1e-12)
    return cfg, levels, sd, torch.from_numpy(z['out'])


def load_fcn_case(name):
    """FCNHeadWithTime fixture: -> (cfg, feat, temb or None, state_dict, out)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    sd = synthetic.make_fcn_state_dict(cfg['num_convs'], cfg['num_classes'], cfg['with_norm'], cfg['concat_input'], cfg['seed'])
    feat, temb = synthetic.make_fcn_inputs(cfg['maps'], cfg['h'], cfg['w'], cfg['seed'])
    assert abs(synthetic.checksum(sd) - float(z['weights_fp'])) <= 1e-9 * abs(float(z['weights_fp']))
    assert np.allclose(fingerprint(feat), z['feat_fp'], rtol=1e-12)
    return cfg, feat, (temb if cfg['with_time'] else None), sd, torch.from_numpy(z['out'])


def load_fpn_case(name):
    """FPN fixture: -> (cfg, backbone levels, state_dict, [4 outputs])."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    sd = synthetic.make_fpn_state_dict(cfg['in_channels'], cfg['seed'])
    levels = synthetic.make_backbone_levels(cfg['batch'], cfg['in_channels'], cfg['h'], cfg['w'], cfg['seed'])
    assert abs(synthetic.checksum(sd) - float(z['weights_fp'])) <= 1e-9 * abs(float(z['weights_fp']))
    assert np.allclose(np.array([fingerprint(t) for t in levels]), z['levels_fp'], rtol=1e-12)
    return cfg, levels, sd, [torch.from_numpy(z[f'out{l}']) for l in range(4)]


def load_aligned_case(name):
    """Self-aligned pre-pass fixture (self_aligned_ddp.py:150-164): -> (cfg, sd, x, noise, arrays)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    sd = synthetic.make_state_dict('seg', cfg['num_classes'], 6, 256, seed=cfg['seed'] + 100)
    x, noise = synthetic.make_inputs(cfg['batch'], cfg['h'], cfg['w'], 1, 256, 256, seed=cfg['seed'])
    noise = noise[:, 0].contiguous()
    assert abs(synthetic.checksum(sd) - float(z['weights_fp'])) <= 1e-9 * abs(float(z['weights_fp']))
    assert np.allclose(fingerprint(x), z['x_fp'], rtol=1e-12) and np.allclose(fingerprint(noise), z['noise_fp'], rtol=1e-12)
    return cfg, sd, x, noise, {k: torch.from_numpy(z[k]) for k in ('preds', 'logits')}


def load_loopfcn_case(name):
    """Sampler loop around FCNHeadWithTime: -> (cfg, sd, x, noise (r,256,h,w), step_noise, arrays)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = json.loads(str(z['config']))
    sd = synthetic.make_fcn_segmentor_state_dict(cfg['num_convs'], cfg['num_classes'], cfg['with_norm'], cfg['concat_input'],
                                                 cfg['seed'] + 100)
    x, noise = synthetic.make_inputs(1, cfg['h'], cfg['w'], cfg['randsteps'], 256, 256, seed=cfg['seed'])
    assert abs(synthetic.checksum(sd) - float(z['weights_fp'])) <= 1e-9 * abs(float(z['weights_fp']))
    assert np.allclose(fingerprint(x), z['x_fp'], rtol=1e-12) and np.allclose(fingerprint(noise), z['noise_fp'], rtol=1e-12)
    step_noise = None
    if cfg['diffusion'] == 'ddpm':
        g = torch.Generator().manual_seed(cfg['seed'] + 7)
        step_noise = torch.randn((cfg['timesteps'], cfg['randsteps'], 256, cfg['h'], cfg['w']), generator=g)
    return cfg, sd, x, noise[0], step_noise, {k: torch.from_numpy(z[k]) for k in ('out', 'logits_steps')}
