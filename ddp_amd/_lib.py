"""ctypes binding of libddp_mi355x.so (include/ddp_mi355x.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing the
engine raises.  Build it with ``python -m ddp_amd.build`` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os

from . import build as _build

MAX_LAYERS = 12
MAX_STEPS = 64
ABI_VERSION = 5

TASK_SEG, TASK_DEPTH, TASK_BEV = 0, 1, 2
SAMPLER_DDIM, SAMPLER_DDPM = 0, 1
GEMM_F32_MFMA, GEMM_BF16X3 = 0, 1
FLAG_UNFUSED_LAYER, FLAG_UNFUSED_PROLOGUE, FLAG_RECORD_X0, FLAG_GATHER_GUESS_ZERO, FLAG_FCN_PREPARED, FLAG_FORCE_X0, FLAG_UNFUSED_TAIL, FLAG_SB_HEAD = 1, 2, 4, 8, 16, 32, 64, 128
FLAG_DEPTH_SCALE_UP, FLAG_DEPTH_NO_EPS = 256, 512
NECK_WEIGHTS_READY = 1

_fp = C.c_void_p  # device pointers travel as raw addresses


class DdpCfg(C.Structure):
    _fields_ = [
        ('abi_version', C.c_int32), ('task', C.c_int32), ('sampler', C.c_int32), ('batch', C.c_int32),
        ('randsteps', C.c_int32), ('timesteps', C.c_int32), ('num_layers', C.c_int32),
        ('num_classes', C.c_int32), ('feat_channels', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
        ('head_h', C.c_int32), ('head_w', C.c_int32), ('accumulation', C.c_int32),
        ('bit_scale', C.c_float), ('min_depth', C.c_float), ('max_depth', C.c_float),
        ('threshold', C.c_float),
        ('bev_in_min', C.c_float * 2), ('bev_in_max', C.c_float * 2),
        ('bev_out_first', C.c_float * 2), ('bev_out_step', C.c_float * 2),
        ('gemm_mode', C.c_int32), ('flags', C.c_int32),
    ]


LAYER_FIELDS = ['sampling_offsets_w', 'sampling_offsets_b', 'attention_weights_w', 'attention_weights_b',
                'value_proj_w', 'value_proj_b', 'output_proj_w', 'output_proj_b', 'ffn0_w', 'ffn0_b',
                'ffn1_w', 'ffn1_b', 'norm0_w', 'norm0_b', 'norm1_w', 'norm1_b', 'time_w', 'time_b']


class DdpLayerWeights(C.Structure):
    _fields_ = [(n, _fp) for n in LAYER_FIELDS]


TOP_FIELDS = ['transform_w', 'transform_b', 'time_freq', 'time1_w', 'time1_b', 'time3_w', 'time3_b',
              'embedding', 'head_w', 'head_b']


class DdpWeights(C.Structure):
    _fields_ = [(n, _fp) for n in TOP_FIELDS] + [('layers', DdpLayerWeights * MAX_LAYERS)]


class DdpStep(C.Structure):
    _fields_ = [('time_in', C.c_float), ('alpha', C.c_float), ('sigma', C.c_float),
                ('alpha_next', C.c_float), ('sigma_next', C.c_float), ('ddpm_c', C.c_float),
                ('ddpm_std', C.c_float), ('ddpm_add_noise', C.c_int32)]


class DdpFpnLevel(C.Structure):
    _fields_ = [('lat_w', _fp), ('lat_gn_w', _fp), ('lat_gn_b', _fp), ('out_w', _fp), ('out_gn_w', _fp), ('out_gn_b', _fp),
                ('in_channels', C.c_int), ('h', C.c_int), ('w', C.c_int)]


class DdpFcnConv(C.Structure):
    _fields_ = [('conv_w', _fp), ('conv_b', _fp), ('bn_w', _fp), ('bn_b', _fp), ('bn_mean', _fp), ('bn_var', _fp),
                ('bn_eps', C.c_float), ('time_w', _fp), ('time_b', _fp)]


class DdpSegAug(C.Structure):
    _fields_ = [('d_scores', _fp), ('h', C.c_int32), ('w', C.c_int32), ('img_h', C.c_int32), ('img_w', C.c_int32),
                ('crop_h', C.c_int32), ('crop_w', C.c_int32), ('flip', C.c_int32)]


class DdpDepthAug(C.Structure):
    _fields_ = [('d_depth', _fp), ('h', C.c_int32), ('w', C.c_int32), ('flip', C.c_int32)]


MAX_AUGS = 16
MAX_WINDOWS = 64

EXPORTS = ['ddp_last_error', 'ddp_abi_version', 'ddp_query_workspace', 'ddp_query_const_workspace', 'ddp_prepare',
           'ddp_prepare_geometry', 'ddp_sample', 'ddp_msda_forward_lds_workspace', 'ddp_msda_forward_lds', 'ddp_seg_aug_postprocess', 'ddp_seg_slide_postprocess', 'ddp_depth_postprocess',
           'ddp_x0_trace', 'ddp_head_forward', 'ddp_msda_forward', 'ddp_linear', 'ddp_linear_b3_workspace', 'ddp_linear_b3', 'ddp_time_embed', 'ddp_ddim_update_seg',
           'ddp_seg_x0_project', 'ddp_seg_postprocess', 'ddp_neck_msm_workspace', 'ddp_neck_msm', 'ddp_fcn_head_workspace', 'ddp_fcn_head_forward', 'ddp_sample_fcn_workspace', 'ddp_prepare_fcn', 'ddp_sample_fcn', 'ddp_neck_fpn_workspace', 'ddp_neck_fpn', 'ddp_neck_fpn_msm_workspace', 'ddp_neck_fpn_msm', 'ddp_profile_begin', 'ddp_profile_end', 'ddp_profile_read']

_libs = {}


class DdpError(RuntimeError):
    pass


def lib_path():
    # DDP_LIB_PATH: A/B another build of the same ABI on one GPU box (scripts/); default = the in-tree library
    return os.environ.get('DDP_LIB_PATH') or _build.LIB_PATH


def load(path=None):
    """Load (once per path) and return the shared library; raises if it is missing - there is no CPU path.
    ``path`` (default: DDP_LIB_PATH or the in-tree build) lets one process hold several builds of the same ABI, which is
    how scripts/ab_bench.py compares code states on one GPU box."""
    path = os.path.abspath(path or lib_path())
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise DdpError(f'{path} not found: build it with `python -m ddp_amd.build` '
                       '(hipcc --offload-arch=gfx950); ddp_amd has no non-HIP fallback')
    if path == os.path.abspath(_build.LIB_PATH) and _build.built_hash() != _build.source_hash():
        # the in-tree library is git-ignored: a checkout / snapshot may carry one built from OTHER sources
        raise DdpError(f'{path} is stale: built from sources {_build.built_hash() or "<no stamp>"}, the tree holds '
                       f'{_build.source_hash()}; rebuild with `python -m ddp_amd.build`')
    lib = C.CDLL(path)
    lib.ddp_last_error.restype = C.c_char_p
    lib.ddp_abi_version.restype = C.c_int
    lib.ddp_query_workspace.argtypes = [C.POINTER(DdpCfg), C.POINTER(C.c_size_t)]
    lib.ddp_prepare.argtypes = [C.POINTER(DdpCfg), C.POINTER(DdpWeights), C.POINTER(DdpStep), _fp, _fp]
    lib.ddp_query_const_workspace.argtypes = [C.POINTER(DdpCfg), C.POINTER(C.c_size_t)]
    lib.ddp_prepare_geometry.argtypes = [C.POINTER(DdpCfg), _fp, _fp]
    lib.ddp_msda_forward_lds_workspace.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    lib.ddp_msda_forward_lds.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]
    lib.ddp_seg_aug_postprocess.argtypes = [C.POINTER(DdpSegAug), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp,
                                            _fp]
    lib.ddp_seg_slide_postprocess.argtypes = [C.POINTER(_fp), C.POINTER(C.c_int), C.POINTER(C.c_int)] + [C.c_int] * 17 + [_fp, _fp, _fp]
    lib.ddp_depth_postprocess.argtypes = [C.POINTER(DdpDepthAug), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                          _fp, _fp]
    lib.ddp_sample.argtypes = [C.POINTER(DdpCfg), C.POINTER(DdpWeights), C.POINTER(DdpStep), _fp, _fp, _fp, _fp,
                               _fp, _fp]
    lib.ddp_x0_trace.argtypes = [C.POINTER(DdpCfg), _fp, C.POINTER(C.c_void_p)]
    lib.ddp_head_forward.argtypes = [C.POINTER(DdpCfg), C.POINTER(DdpWeights), _fp, _fp, _fp, _fp, _fp]
    lib.ddp_msda_forward.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]
    lib.ddp_linear.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
    lib.ddp_linear_b3_workspace.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    lib.ddp_linear_b3.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]
    lib.ddp_time_embed.argtypes = [C.POINTER(DdpWeights), C.c_int, C.POINTER(C.c_float), C.c_int, _fp, _fp, _fp, _fp]
    lib.ddp_ddim_update_seg.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp, C.c_int, C.POINTER(DdpStep), _fp]
    lib.ddp_seg_x0_project.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_float, _fp, _fp]
    lib.ddp_seg_postprocess.argtypes = [_fp] + [C.c_int] * 12 + [_fp, _fp]
    lib.ddp_neck_msm_workspace.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    lib.ddp_neck_msm.argtypes = [C.POINTER(_fp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, _fp,
                                 _fp, _fp]
    lib.ddp_fcn_head_workspace.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    lib.ddp_fcn_head_forward.argtypes = [C.POINTER(DdpFcnConv), C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int,
                                         C.c_int, _fp, _fp, _fp]
    lib.ddp_sample_fcn_workspace.argtypes = [C.POINTER(DdpCfg), C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    lib.ddp_prepare_fcn.argtypes = [C.POINTER(DdpCfg), C.POINTER(DdpWeights), C.POINTER(DdpFcnConv), C.c_int, C.c_int,
                                    C.POINTER(DdpStep), _fp, _fp]
    lib.ddp_sample_fcn.argtypes = [C.POINTER(DdpCfg), C.POINTER(DdpWeights), C.POINTER(DdpFcnConv), C.c_int, C.c_int,
                                   C.POINTER(DdpStep), _fp, _fp, _fp, _fp, _fp, _fp]
    lib.ddp_neck_fpn_workspace.argtypes = [C.POINTER(DdpFpnLevel), C.c_int, C.POINTER(C.c_size_t)]
    lib.ddp_neck_fpn.argtypes = [C.POINTER(DdpFpnLevel), C.c_int, C.POINTER(_fp), C.POINTER(_fp), C.c_int, _fp, _fp]
    lib.ddp_neck_fpn_msm_workspace.argtypes = [C.POINTER(DdpFpnLevel), C.c_int, C.POINTER(C.c_size_t)]
    lib.ddp_neck_fpn_msm.argtypes = [C.POINTER(DdpFpnLevel), C.c_int, C.POINTER(_fp), _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp]
    lib.ddp_profile_begin.argtypes = [C.c_int]
    lib.ddp_profile_end.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.ddp_profile_read.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    for n in EXPORTS:
        if n not in ('ddp_last_error',):
            getattr(lib, n).restype = C.c_int if n != 'ddp_last_error' else C.c_char_p
    lib.ddp_last_error.restype = C.c_char_p
    if lib.ddp_abi_version() != ABI_VERSION:
        raise DdpError(f'ABI mismatch: library {lib.ddp_abi_version()} vs binding {ABI_VERSION}')
    _libs[path] = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        raise DdpError(f'libddp_mi355x error {rc}: {(lib or load()).ddp_last_error().decode()}')
