#!/usr/bin/env python
"""Drop-in proof against the REFERENCE's own registries.  Build container only: it imports the reference
packages from /root/reference through tests/golden/ref_shim.py, so it runs as a subprocess of
tests/test_dropin_reference_registry.py (the import mutates sys.modules and the mmcv registries) and is skipped where
/root/reference is absent (the GPU box).

    python tests/dropin_probe.py seg     # segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py
    python tests/dropin_probe.py depth   # depth/configs/ddp_kitti/ddp_swint_1k_w7_kitti_bs2x8_scale01.py
    python tests/dropin_probe.py controlnet   # the mmseg copy of the ControlNet demo (controlnet/annotator/ddp/mmseg, imported as
                                              # top-level ``mmseg`` the way controlnet/annotator/ddp/__init__.py:2 does) on the ADE config

Steps: (1) build the shipped config with the reference's ``build_segmentor`` / ``build_depther`` -> the reference model;
(2) ``ddp_amd.register_into_mmseg()``; (3) build the SAME config with the SAME reference builder again
(segmentation/mmseg/models/builder.py:38-49) -> must now resolve to ddp_amd's classes for the segmentor, the decode head and
the necks, while the frozen backbone stays the reference's; (4) the state_dict key -> shape maps of everything except the
training-only auxiliary head must be equal, and the reference's weights must load with strict=True; (5) the model is CALLED the
way the toolbox's harness calls it - ``model(return_loss=False, **data)`` with the nested augmentation lists the test
pipeline + collate produce (segmentation/mmseg/apis/test.py:87-89, depth/depth/apis/test.py:88,204; KITTI: two augmentations,
plain + flipped, depth/configs/_base_/datasets/kitti.py:30-33).  On this CPU-only container the call must travel through
``forward`` -> ``forward_test`` -> ``simple_test`` / ``aug_test`` -> ... and end in ddp_amd's explicit "no CPU path" error
- first at the neck (the reference backbone does run on the CPU), then, with the feature map handed in, inside the sampler
(``ddim_sample`` / ``sample``) - never in ``nn.Module._forward_unimplemented`` or an AttributeError.
Prints one JSON line.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, 'golden'))


def shapes(model, skip=('auxiliary_head.',)):
    return {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.startswith(skip)}


def call_like_the_harness(model, task):
    """(5) of the module docstring -> {'as_built': [...frames...], 'feature_given': [...frames...]}."""
    import traceback

    import torch
    model.eval()
    H, W = 64, 96
    meta = dict(filename='a.png', ori_filename='a.png', ori_shape=(H, W, 3), img_shape=(H, W, 3), pad_shape=(H, W, 3),
                scale_factor=1.0, flip=False, flip_direction='horizontal',
                img_norm_cfg=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True))
    n_aug = 2 if task == 'depth' else 1                     # KITTI: flip=True in MultiScaleFlipAug; ADE: single scale, no flip
    data = dict(img=[torch.zeros(1, 3, H, W) for _ in range(n_aug)],
                img_metas=[[dict(meta, flip=bool(i))] for i in range(n_aug)])
    sampler = 'sample' if task == 'depth' else 'ddim_sample'
    entry = ('forward', 'forward_test', 'aug_test') if task == 'depth' else ('forward', 'simple_test')

    def frames_of_call():
        try:
            with torch.no_grad():
                model(return_loss=False, **data)
        except Exception as e:                              # noqa: BLE001 - the probe reports what was raised
            names = [f.name for f in traceback.extract_tb(e.__traceback__)]
            assert 'no CPU path' in str(e), f'{type(e).__name__}: {e}  via {names}'
            assert '_forward_unimplemented' not in names, names
            for fn in entry:
                assert fn in names, (fn, names)
            return names
        raise AssertionError('the call returned on a CPU-only host: some CPU fallback ran')

    as_built = frames_of_call()                             # reference backbone on the CPU -> ddp_amd neck refuses
    assert 'extract_feat' in as_built, as_built
    model.extract_feat = lambda img: [torch.zeros(img.shape[0], 256, H // 4, W // 4)]
    given = frames_of_call()                                # feature handed in -> the sampler itself refuses
    assert given[-1] in (sampler, '_check_feature'), given
    assert sampler in given, given
    return dict(as_built=as_built[-4:], feature_given=given[-4:])


def main(task):
    import ref_shim
    flavour = task
    if task == 'controlnet':
        # the copy under controlnet/ FIRST on sys.path: ``mmseg`` resolves to it, not to segmentation/mmseg
        ref_shim.install()
        sys.path.insert(0, os.path.join(ref_shim.REF, 'controlnet', 'annotator', 'ddp'))
        from mmcv import Config
        from mmseg.models import build_segmentor as build
        import mmseg
        assert os.path.join('controlnet', 'annotator', 'ddp') in mmseg.__file__, mmseg.__file__
        cfg_path = os.path.join(ref_shim.REF, 'segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py')
        task = 'seg'
    elif task == 'seg':
        build, Config, _ = ref_shim.import_seg()
        cfg_path = os.path.join(ref_shim.REF, 'segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py')
    else:
        build, Config = ref_shim.import_depth()
        cfg_path = os.path.join(ref_shim.REF, 'depth/configs/ddp_kitti/ddp_swint_1k_w7_kitti_bs2x8_scale01.py')

    def model_cfg():
        m = Config.fromfile(cfg_path).model
        m.backbone.init_cfg = None          # no checkpoint download
        m.train_cfg = None
        if flavour == 'controlnet':         # that copy's DDP predates ``accumulation`` (its ddp.py has no such argument)
            m.pop('accumulation', None)
        return m

    ref = build(model_cfg())
    ref_cls = {'segmentor': type(ref), 'head': type(ref.decode_head), 'backbone': type(ref.backbone)}
    assert ref_cls['segmentor'].__module__.split('.')[0] in ('mmseg', 'depth'), ref_cls

    import ddp_amd
    touched = ddp_amd.register_into_mmseg(package='mmseg') if flavour == 'controlnet' else ddp_amd.register_into_mmseg()
    assert ('mmseg' if task == 'seg' else 'depth') in touched, touched
    ours = build(model_cfg())

    mod = lambda o: type(o).__module__
    assert mod(ours).startswith('ddp_amd.'), type(ours)
    assert mod(ours.decode_head).startswith('ddp_amd.'), type(ours.decode_head)
    assert type(ours) is not ref_cls['segmentor'] and type(ours.decode_head) is not ref_cls['head']
    assert type(ours.backbone) is ref_cls['backbone'], (type(ours.backbone), ref_cls['backbone'])   # frozen backbone: the host toolbox's
    necks = []
    if task == 'seg':
        assert type(ours) is ddp_amd.DDP and type(ours.decode_head) is ddp_amd.DeformableHeadWithTime
        necks = [type(n).__name__ for n in ours.neck]
        assert all(mod(n).startswith('ddp_amd.') for n in ours.neck), [mod(n) for n in ours.neck]
        assert necks == ['FPN', 'MultiStageMerging'], necks
    else:
        assert type(ours) is ddp_amd.DepthDDP and type(ours.decode_head) is ddp_amd.DepthDeformableHeadWithTime

    a, b = shapes(ref), shapes(ours)
    only_ref = sorted(set(a) - set(b))
    only_ours = sorted(set(b) - set(a))
    differ = sorted(k for k in set(a) & set(b) if a[k] != b[k])
    assert not only_ref and not only_ours and not differ, dict(only_ref=only_ref[:8], only_ours=only_ours[:8], differ=differ[:8])
    res = ours.load_state_dict({k: v for k, v in ref.state_dict().items() if not k.startswith('auxiliary_head.')}, strict=True)
    hot = [k for k in b if not k.startswith(('backbone.', 'neck.'))]
    called = call_like_the_harness(ours, task)
    print(json.dumps(dict(task=flavour, touched=touched, segmentor=f'{mod(ours)}.{type(ours).__name__}',
                          head=f'{mod(ours.decode_head)}.{type(ours.decode_head).__name__}', necks=necks,
                          backbone=f'{mod(ours.backbone)}.{type(ours.backbone).__name__}', keys=len(b), hot_path_keys=len(hot),
                          hot_path_params=int(sum(ours.state_dict()[k].numel() for k in hot)), strict_load=str(res),
                          called=called)))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'seg')
