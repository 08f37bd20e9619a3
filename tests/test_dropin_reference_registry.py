"""Drop-in proof in the reference's OWN registries (SURVEY.md §8b "drop-in boundary").

Runs ``tests/dropin_probe.py`` in a subprocess (the reference import mutates sys.modules / the mmcv registries):
``register_into_mmseg()`` + the reference's ``build_segmentor`` / ``build_depther`` on two shipped configs must resolve to
ddp_amd's segmentor / head / neck classes, keep the reference's backbone, and expose a state_dict whose key -> shape map
equals the reference model's (strict load both ways), and the built model must be CALLABLE the way the toolboxes' test
harnesses call it (``model(return_loss=False, **data)``): on this GPU-less host the call has to end in ddp_amd's explicit
"no CPU path" error inside the sampler, not in ``_forward_unimplemented``.  CPU only; skipped where /root/reference is absent (GPU box).
"""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason='reference tree not present (build container only)')


def _probe(task):
    r = subprocess.run([sys.executable, os.path.join(HERE, 'dropin_probe.py'), task], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])


def test_ade_config_builds_to_ddp_amd_classes_in_mmseg_registry():
    d = _probe('seg')
    assert d['touched'] == ['mmseg']
    assert d['segmentor'] == 'ddp_amd.segmentors.ddp.DDP'
    assert d['head'] == 'ddp_amd.decode_heads.deformable_head_with_time.DeformableHeadWithTime'
    assert d['necks'] == ['FPN', 'MultiStageMerging']
    assert d['backbone'].startswith('mmseg.models.backbones.')          # frozen backbone stays the host toolbox's
    assert d['hot_path_params'] == 8522462                              # SURVEY.md §8b probe of the reference model
    assert 'All keys matched' in d['strict_load']
    # the harness call model(return_loss=False, **data) reaches the sampler (and stops there: this host has no GPU)
    assert 'simple_test' in d['called']['feature_given'] and 'ddim_sample' in d['called']['feature_given']


def test_kitti_config_builds_to_ddp_amd_classes_in_depth_registry():
    d = _probe('depth')
    assert d['touched'] == ['depth']
    assert d['segmentor'] == 'ddp_amd.depther.ddp.DDP'
    assert d['head'] == 'ddp_amd.depther.ddp.DepthDeformableHeadWithTime'
    assert d['backbone'].startswith('depth.models.backbones.')
    assert 'All keys matched' in d['strict_load']
    # KITTI's two-augmentation harness call (depth/depth/apis/test.py:88) goes forward -> forward_test -> aug_test -> sample
    assert 'aug_test' in d['called']['feature_given'] and d['called']['feature_given'][-1] == 'sample'


def test_controlnet_mmseg_copy_builds_to_ddp_amd_classes():
    """VERDICT r05 missing #5: the ControlNet demo ships its own copy of the toolbox (controlnet/annotator/ddp/mmseg, imported as
    top-level ``mmseg``: controlnet/annotator/ddp/__init__.py:2; used by controlnet/gradio_seg2image_ddp.py:2,24,35).
    ``register_into_mmseg(package=...)`` covers it: the same ADE config built through THAT copy's builder resolves to ddp_amd's
    classes, strict state_dict both ways, harness call reaches the sampler."""
    d = _probe('controlnet')
    assert d['touched'] == ['mmseg']
    assert d['segmentor'] == 'ddp_amd.segmentors.ddp.DDP'
    assert d['head'] == 'ddp_amd.decode_heads.deformable_head_with_time.DeformableHeadWithTime'
    assert d['necks'] == ['FPN', 'MultiStageMerging']
    assert d['hot_path_params'] == 8522462
    assert 'All keys matched' in d['strict_load']
    assert 'ddim_sample' in d['called']['feature_given']


def test_register_into_a_vendored_package_name(tmp_path):
    """the ``package`` argument with a dotted name: a toolbox vendored under another root (here a stub tree
    ``vendored_pkg.sub.mmseg.models.builder`` holding three registries with mmcv's ``register_module`` signature)"""
    code = (
        "import sys, types\n"
        "sys.path.insert(0, %r)\n"
        "class Reg:\n"
        "    def __init__(self): self.module_dict = {}\n"
        "    def register_module(self, name=None, force=False, module=None): self.module_dict[name] = module\n"
        "names = ['vendored_pkg', 'vendored_pkg.sub', 'vendored_pkg.sub.mmseg', 'vendored_pkg.sub.mmseg.models', 'vendored_pkg.sub.mmseg.models.builder']\n"
        "for n in names:\n"
        "    m = types.ModuleType(n); m.__path__ = []; sys.modules[n] = m\n"
        "b = sys.modules[names[-1]]; b.SEGMENTORS, b.HEADS, b.NECKS = Reg(), Reg(), Reg()\n"
        "import ddp_amd\n"
        "t = ddp_amd.register_into_mmseg(package='vendored_pkg.sub.mmseg', depth_package='no_such_depth_toolbox')\n"
        "assert t == ['vendored_pkg.sub.mmseg'], t\n"
        "assert b.SEGMENTORS.module_dict['DDP'] is ddp_amd.DDP and 'DeformableHeadWithTime' in b.HEADS.module_dict and 'FPN' in b.NECKS.module_dict\n"
        "print('ok')\n" % os.path.dirname(HERE))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout + r.stderr
