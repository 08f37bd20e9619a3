"""ddp_amd - MI355X-native DDP (Diffusion model for Dense visual Prediction) inference loop.

The product is ``ddp_amd/lib/libddp_mi355x.so`` (HIP, gfx950) behind the C ABI of
``include/ddp_mi355x.h``; this package is the thin Python host layer that mirrors the reference's
plugin surface (registered ``DDP`` / ``DeformableHeadWithTime`` classes with the same kwargs and
state_dict keys) and passes torch device pointers to it.  There is no CPU / eager fallback.
"""
from .registry import (MODELS, SEGMENTORS, HEADS, build_segmentor, build_depther, build_head,  # noqa: F401
                       register_into_mmseg, register_into_mmdet3d)
from .segmentors.ddp import DDP, SelfAlignedDDP  # noqa: F401
from .decode_heads.deformable_head_with_time import DeformableHeadWithTime  # noqa: F401
from .decode_heads.fcn_head_with_time import FCNHeadWithTime  # noqa: F401
from .depther.ddp import DDP as DepthDDP, DepthDeformableHeadWithTime  # noqa: F401
from .bev.ddp import DDP as BEVDDP, BEVDeformableHeadWithTime  # noqa: F401
from .necks import FPN, MultiStageMerging, NeckChain  # noqa: F401
from .apis import single_gpu_test, multi_gpu_test, collect_results  # noqa: F401

__all__ = ['DDP', 'SelfAlignedDDP', 'DeformableHeadWithTime', 'FCNHeadWithTime', 'DepthDDP', 'DepthDeformableHeadWithTime', 'BEVDDP',
           'BEVDeformableHeadWithTime', 'FPN', 'MultiStageMerging', 'NeckChain', 'build_segmentor', 'build_depther', 'build_head', 'register_into_mmseg', 'register_into_mmdet3d',
           'single_gpu_test', 'multi_gpu_test', 'collect_results']
