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


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


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


def register_into_mmseg():
    """Register the MI355X classes over the reference ones in an installed mmseg / depth toolbox.
    Returns the list of registries touched (empty when none is importable)."""
    touched = []
    from .segmentors.ddp import DDP, SelfAlignedDDP
    from .decode_heads.deformable_head_with_time import DeformableHeadWithTime
    from .depther.ddp import DDP as DepthDDP, DepthDeformableHeadWithTime
    try:
        from mmseg.models.builder import SEGMENTORS as MS, HEADS as MH, NECKS as MN
        from .necks import FPN, MultiStageMerging
        MN.register_module(name='MultiStageMerging', force=True, module=MultiStageMerging)
        MN.register_module(name='FPN', force=True, module=FPN)
        MS.register_module(name='DDP', force=True, module=DDP)
        MS.register_module(name='SelfAlignedDDP', force=True, module=SelfAlignedDDP)
        MH.register_module(name='DeformableHeadWithTime', force=True, module=DeformableHeadWithTime)
        from .decode_heads.fcn_head_with_time import FCNHeadWithTime
        MH.register_module(name='FCNHeadWithTime', force=True, module=FCNHeadWithTime)
        touched.append('mmseg')
    except Exception:
        pass
    try:
        from depth.models.builder import DEPTHER as DD, HEADS as DH
        DD.register_module(name='DDP', force=True, module=DepthDDP)
        DH.register_module(name='DeformableHeadWithTime', force=True, module=DepthDeformableHeadWithTime)
        touched.append('depth')
    except Exception:
        pass
    return touched
