"""Minimal registries mirroring the mmcv plugin mechanism the reference uses
(segmentation/mmseg/models/builder.py:8-15: SEGMENTORS/HEADS/NECKS/BACKBONES share one MODELS
registry; ``build_segmentor(cfg)`` resolves ``cfg['type']`` by name).  mmcv is not installed in
this image, so the drop-in classes register here; ``register_into_mmseg()`` additionally registers
them (force=True) into a real MMSegmentation / depth toolbox when one is importable, which is how
they replace the reference classes in segmentation/mmseg and depth/ without touching configs."""
import inspect


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
        args = dict(cfg)
        typ = args.pop('type')
        cls = typ if inspect.isclass(typ) else self.get(typ)
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        for k, v in default_args.items():
            args.setdefault(k, v)
        return cls(**args)


MODELS = Registry('models')
BACKBONES = NECKS = HEADS = LOSSES = SEGMENTORS = DEPTHER = FUSIONMODELS = MODELS


def _foreign_builder(kind):
    """``build_<kind>`` of an importable MMSegmentation / depth toolbox (None when neither is installed).  The frozen
    backbones (SwinTransformer, ConvNeXt, ...) are NOT part of ddp_amd (they stay PyTorch-ROCm modules of the
    host toolbox), so a config that names one is resolved where the reference resolves it
    (segmentation/mmseg/models/builder.py:18-31; depth/depth/models/builder.py)."""
    import importlib
    for pkg in ('mmseg.models.builder', 'depth.models.builder'):
        try:
            mod = importlib.import_module(pkg)
        except ImportError:
            continue
        fn = getattr(mod, 'build_' + kind, None)
        if fn is not None:
            yield pkg, fn


def _build_or_delegate(kind, registry, cfg):
    """Build ``cfg`` from ddp_amd's registry when it knows the type; an already constructed ``nn.Module`` passes through;
    any other type goes to the host toolbox's builder.  Fails loudly (KeyError naming every place that was tried)."""
    if cfg is None:
        return None
    if not isinstance(cfg, dict):
        import torch.nn as nn
        if isinstance(cfg, nn.Module):
            return cfg
        raise TypeError(f'{kind} must be a config dict or an nn.Module, got {type(cfg).__name__}')
    typ = cfg.get('type')
    if inspect.isclass(typ) or typ in registry:
        return registry.build(cfg)
    tried = [f'ddp_amd.{registry.name}']
    last = None
    for pkg, fn in _foreign_builder(kind):
        tried.append(pkg)
        # "is the type registered there?" is asked of the toolbox's registry itself; a KeyError raised INSIDE a registered
        # constructor (missing cfg key, nested build failure) is the caller's error and propagates unchanged
        import importlib
        reg = getattr(importlib.import_module(pkg), kind.upper() + 'S', None)
        # ask the registry's own RESOLVER, not its private dict: ``module_dict`` holds only that registry's unscoped names,
        # while scoped types ('mmcls.ConvNeXt' of the reference's ConvNeXt configs, configs/cityscapes/
        # ddp_convnext_t_4x4_512x1024_160k_cityscapes.py:17) and parent / child registries resolve through ``get``
        resolver = getattr(reg, 'get', None)
        if callable(resolver) and isinstance(typ, str):
            try:
                known = resolver(typ) is not None
            except Exception:                     # a resolver that cannot even parse the name: let the builder decide
                known = None
            if known:
                return fn(cfg)
            if known is False:
                continue
        elif getattr(reg, 'module_dict', None) is not None:
            if typ in reg.module_dict:
                return fn(cfg)
            continue
        try:
            return fn(cfg)
        except KeyError as e:                     # toolbox without an inspectable registry: keep the cause attached
            last = e
            continue
    raise KeyError(f'{kind} type {typ!r} is not registered in any of {tried}: ddp_amd implements the DDP hot path only - '
                   f'build the {kind} with the host toolbox (mmseg / depth) or pass a constructed nn.Module') from last


def build_backbone(cfg):
    return _build_or_delegate('backbone', BACKBONES, cfg)


def build_neck(cfg):
    return _build_or_delegate('neck', NECKS, cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_segmentor(cfg, train_cfg=None, test_cfg=None):
    """Same call convention as mmseg.models.builder.build_segmentor (builder.py:38-49)."""
    if train_cfg is not None or test_cfg is not None:
        cfg = dict(cfg)
        if train_cfg is not None:
            cfg.setdefault('train_cfg', train_cfg)
        if test_cfg is not None:
            cfg.setdefault('test_cfg', test_cfg)
    return SEGMENTORS.build(cfg)


def build_depther(cfg, train_cfg=None, test_cfg=None):
    """depth/depth/models/builder.py ``build_depther``: the depth toolbox registers its own classes
    under the same names ('DDP', 'DeformableHeadWithTime'); map them onto the depth variants here."""
    cfg = dict(cfg)
    if cfg.get('type') == 'DDP':
        cfg['type'] = 'DepthDDP'
    return build_segmentor(cfg, train_cfg, test_cfg)


def register_into_mmseg(package='mmseg', depth_package='depth'):
    """Register the MI355X classes over the reference ones in an importable mmseg / depth toolbox.
    ``package`` / ``depth_package``: the dotted name the toolbox is importable under - the top-level ``mmseg`` / ``depth`` of
    segmentation/ and depth/, equally the copy the ControlNet demo ships (controlnet/annotator/ddp/mmseg, put on sys.path as
    top-level ``mmseg`` by controlnet/annotator/ddp/__init__.py:2 and used by controlnet/gradio_seg2image_ddp.py:2,24,35), or a
    tree vendored under another root (``package='annotator.ddp.mmseg'``).  Names a toolbox does not define (the ControlNet copy has
    no SelfAlignedDDP) are registered all the same - a config that never asks for them never sees them.
    Returns the list of registries touched (empty when none is importable)."""
    import importlib
    touched = []
    from .segmentors.ddp import DDP, SelfAlignedDDP
    from .decode_heads.deformable_head_with_time import DeformableHeadWithTime
    from .depther.ddp import DDP as DepthDDP, DepthDeformableHeadWithTime
    # only "toolbox not installed" is tolerated: a failure half way through a registration would leave a partial
    # drop-in behind and must surface
    try:
        bld = importlib.import_module(package + '.models.builder')
        MS, MH, MN = bld.SEGMENTORS, bld.HEADS, bld.NECKS
    except ImportError:
        MS = None
    if MS is not None:
        from .necks import FPN, MultiStageMerging
        MN.register_module(name='MultiStageMerging', force=True, module=MultiStageMerging)
        MN.register_module(name='FPN', force=True, module=FPN)
        MS.register_module(name='DDP', force=True, module=DDP)
        MS.register_module(name='SelfAlignedDDP', force=True, module=SelfAlignedDDP)
        MH.register_module(name='DeformableHeadWithTime', force=True, module=DeformableHeadWithTime)
        from .decode_heads.fcn_head_with_time import FCNHeadWithTime
        MH.register_module(name='FCNHeadWithTime', force=True, module=FCNHeadWithTime)
        touched.append(package)
    try:
        bld = importlib.import_module(depth_package + '.models.builder')
        DD, DH = bld.DEPTHER, bld.HEADS
    except ImportError:
        DD = None
    if DD is not None:
        DD.register_module(name='DDP', force=True, module=DepthDDP)
        DH.register_module(name='DeformableHeadWithTime', force=True, module=DepthDeformableHeadWithTime)
        touched.append(depth_package)
    return touched


def register_into_mmdet3d():
    """Register the MI355X BEV classes under the REFERENCE's names in an importable BEVFusion-style ``mmdet3d``
    (bev/mmdet3d/models/fusion_models/ddp.py:65-66 ``@FUSIONMODELS.register_module() class DDP``;
    bev/mmdet3d/models/heads/segm/deformable_head_with_time.py ``@HEADS.register_module() class DeformableHeadWithTime``),
    force=True, so that ``type='DDP'`` / ``type='DeformableHeadWithTime'`` in the bev configs resolve to the HIP path.
    Only the sampling loop is replaced: the sensor encoders / fuser / BEV decoder of ``BEVFusion`` are out of scope
     and stay the toolbox's.  Returns the list of registries touched (empty when mmdet3d is not importable)."""
    try:
        from mmdet3d.models.builder import FUSIONMODELS as MF, HEADS as MH
    except ImportError:
        return []
    from .bev.ddp import DDP as BevDDP, BEVDeformableHeadWithTime
    MF.register_module(name='DDP', force=True, module=BevDDP)
    MH.register_module(name='DeformableHeadWithTime', force=True, module=BEVDeformableHeadWithTime)
    return ['mmdet3d.FUSIONMODELS', 'mmdet3d.HEADS']
