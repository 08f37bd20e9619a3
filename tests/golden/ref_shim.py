"""Import shim: lets the *reference* Python packages under /root/reference be imported in the
build container so golden vectors can be generated from the reference itself.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ddp_amd/), bench.py's timed region or
the ``-m gpu`` tests imports this module; /root/reference does not exist on the GPU box.  The
only consumer is ``tests/golden/gen_golden.py`` (run by hand in the build container) and the
optional ``-m "not gpu"`` test that re-checks the oracle against the live reference when
/root/reference is present.

What it does:
  * the reference imports ``mmcv`` (mmcv-full==1.6.2, not vendored for segmentation/).  A pure
    python mmcv 1.3.17 is vendored at controlnet/annotator/uniformer/mmcv; a meta-path finder
    aliases ``mmcv[.x]`` to ``annotator.uniformer.mmcv[.x]``;
  * heavy ``annotator`` package __init__ files are bypassed with namespace stubs;
  * absent third-party modules (addict, yapf, cv2, torchvision, prettytable, timm, mmcv._ext)
    are replaced by minimal stand-ins that the hot path never executes.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types
from unittest.mock import MagicMock

REF = os.environ.get('DDP_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'segmentation', 'mmseg'))


class _AttrDict(dict):
    """Tiny stand-in for addict.Dict (base class of mmcv ConfigDict)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for d in args:
            for k, v in dict(d).items():
                self[k] = self._wrap(v)
        for k, v in kwargs.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, _AttrDict) else v) for k, v in self.items()}


class _MmcvAlias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if (name == 'mmcv' or name.startswith('mmcv.')) and name != 'mmcv._ext':
            return importlib.machinery.ModuleSpec(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module('annotator.uniformer.mmcv' + spec.name[4:])

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Idempotently install the stubs + alias finder."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f'reference tree not found under {REF}')
    ann = os.path.join(REF, 'controlnet', 'annotator')
    for name, path in (('annotator', ann), ('annotator.uniformer', os.path.join(ann, 'uniformer'))):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    sys.modules['addict'] = types.SimpleNamespace(Dict=_AttrDict)
    for n in ['yapf', 'yapf.yapflib', 'yapf.yapflib.yapf_api', 'cv2', 'torchvision',
              'torchvision.models', 'prettytable', 'timm', 'timm.models', 'timm.models.layers',
              'mmcv._ext']:
        sys.modules.setdefault(n, MagicMock())
    sys.meta_path.insert(0, _MmcvAlias())
    _installed = True


def _force_registry():
    """Make every mmcv Registry registration behave as force=True (package co-import)."""
    from mmcv.utils import Registry
    if getattr(Registry, '_ddp_forced', False):
        return
    orig = Registry._register_module

    def _reg(self, module_class, module_name=None, force=False):
        return orig(self, module_class, module_name=module_name, force=True)

    Registry._register_module = _reg
    Registry._ddp_forced = True


def import_seg():
    """-> (build_segmentor, Config, revert_sync_batchnorm) of the reference segmentation/ tree."""
    install()
    p = os.path.join(REF, 'segmentation')
    if p not in sys.path:
        sys.path.insert(0, p)
    from mmcv import Config
    from mmcv.cnn.utils import revert_sync_batchnorm
    from mmseg.models import build_segmentor
    return build_segmentor, Config, revert_sync_batchnorm


def load_time_aware_layer():
    """exec segmentation/mmseg/models/utils/transformer.py by path so that the time-aware
    BaseTransformerLayer is the registered one (needed by depth/ and bev/)."""
    install()
    _force_registry()
    path = os.path.join(REF, 'segmentation', 'mmseg', 'models', 'utils', 'transformer.py')
    spec = importlib.util.spec_from_file_location('_ddp_seg_transformer', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_depth():
    """-> (build_depther, Config) of the reference depth/ tree, with the time-aware layer."""
    install()
    _force_registry()
    p = os.path.join(REF, 'depth')
    if p not in sys.path:
        sys.path.insert(0, p)
    stub = types.ModuleType('depth.models.depther.regulardepth')
    stub.RegularDepth = None          # file is missing from the reference tree
    sys.modules['depth.models.depther.regulardepth'] = stub
    from mmcv import Config
    from depth.models import build_depther
    load_time_aware_layer()
    return build_depther, Config


def import_bev():
    """-> (DDP class, DeformableHeadWithTime class) of the reference bev/ tree (two hot-path
    files only; the mmdet3d package as a whole needs compiled ops that are not shipped)."""
    install()
    _force_registry()
    import torch.nn as nn
    from mmcv.utils import Registry
    root = os.path.join(REF, 'bev', 'mmdet3d')

    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    ns('mmdet3d', root)
    ns('mmdet3d.models', os.path.join(root, 'models'))
    ns('mmdet3d.models.fusion_models', os.path.join(root, 'models', 'fusion_models'))
    ns('mmdet3d.models.heads', os.path.join(root, 'models', 'heads'))
    ns('mmdet3d.models.heads.segm', os.path.join(root, 'models', 'heads', 'segm'))
    ops = ns('mmdet3d.ops', os.path.join(root, 'ops'))
    ops.Voxelization = MagicMock()
    ops.DynamicScatter = MagicMock()

    builder = types.ModuleType('mmdet3d.models.builder')
    builder.HEADS = Registry('bev_heads')
    builder.FUSIONMODELS = Registry('bev_fusion_models')
    for fn in ('build_backbone', 'build_fuser', 'build_head', 'build_neck', 'build_vtransform',
               'build_loss', 'build_fusion_model', 'build_model'):
        setattr(builder, fn, lambda *a, **k: None)
    sys.modules['mmdet3d.models.builder'] = builder
    sys.modules['mmdet3d.models'].builder = builder
    for k in ('HEADS', 'FUSIONMODELS'):
        setattr(sys.modules['mmdet3d.models'], k, getattr(builder, k))

    bevfusion = types.ModuleType('mmdet3d.models.fusion_models.bevfusion')

    class BEVFusion(nn.Module):
        def __init__(self, *args, **kwargs):
            super().__init__()

    bevfusion.BEVFusion = BEVFusion
    sys.modules['mmdet3d.models.fusion_models.bevfusion'] = bevfusion
    base = types.ModuleType('mmdet3d.models.fusion_models.base')

    class Base3DFusionModel(nn.Module):
        def __init__(self, *args, **kwargs):
            super().__init__()

    base.Base3DFusionModel = Base3DFusionModel
    sys.modules['mmdet3d.models.fusion_models.base'] = base

    load_time_aware_layer()
    ddp = importlib.import_module('mmdet3d.models.fusion_models.ddp')
    head = importlib.import_module('mmdet3d.models.heads.segm.deformable_head_with_time')
    return ddp, head
