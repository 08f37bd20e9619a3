"""Drop-in ``FCNHeadWithTime``: same registry name, constructor kwargs, parameter names (hence
``state_dict`` keys: ``convs.i.conv.weight``, ``convs.i.bn.*``, ``convs.i.time_mlp.1.*``, ``conv_cat.*``,
``conv_seg.*``) and ``forward(inputs, times)`` contract as
segmentation/mmseg/models/decode_heads/fcn_head_with_time.py:229-305, with the compute done by libddp_mi355x.so
(``ddp_fcn_head_forward``): every ConvWithTimeModule is ONE bf16x3 GEMM (K = 9 x 256) whose weights carry the
eval-mode norm x FiLM scale and whose epilogue carries the shift and the ReLU.  Inference only (eval-mode BatchNorm);
CUDA tensors only - no CPU path.  The nn.Modules below only HOLD parameters in the reference's layout.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from ..registry import HEADS


class _ConvWithTimeParams(nn.Module):
    def __init__(self, cin, cout, with_norm, time_in_channels=1024):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1, bias=not with_norm)      # ConvModule bias='auto'
        if with_norm:
            self.bn = nn.BatchNorm2d(cout)                                       # norm name 'bn' for BN / SyncBN
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_in_channels, cout * 2))
        nn.init.kaiming_normal_(self.conv.weight, a=0, nonlinearity='relu')


class _ConvParams(nn.Module):
    def __init__(self, cin, cout, k, with_norm):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2, bias=not with_norm)
        if with_norm:
            self.bn = nn.BatchNorm2d(cout)


@HEADS.register_module()
class FCNHeadWithTime(nn.Module):
    def __init__(self, num_convs=2, kernel_size=3, concat_input=True, dilation=1, in_channels=256, channels=256,
                 num_classes=150, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), in_index=-1,
                 input_transform=None, loss_decode=None, ignore_index=255, sampler=None, align_corners=False,
                 init_cfg=None, **_):
        super().__init__()
        if isinstance(in_channels, (list, tuple)):
            in_channels = in_channels[0]
        if in_channels != 256 or channels != 256 or kernel_size != 3 or input_transform is not None:
            raise NotImplementedError('ddp_amd FCNHeadWithTime: 256 -> 256 channels, 3x3 convs, single input')
        if norm_cfg is not None and norm_cfg.get('type') not in ('BN', 'SyncBN'):
            raise NotImplementedError('ddp_amd FCNHeadWithTime: norm_cfg None or (Sync)BN (eval-mode running statistics)')
        if act_cfg is None or act_cfg.get('type') != 'ReLU':
            raise NotImplementedError('ddp_amd FCNHeadWithTime: act_cfg=ReLU')
        assert num_convs >= 0 and dilation > 0
        self.num_convs, self.kernel_size, self.concat_input, self.dilation = num_convs, kernel_size, concat_input, dilation
        self.in_channels, self.channels, self.num_classes = [in_channels], channels, num_classes
        self.in_index, self.align_corners = in_index, align_corners
        with_norm = norm_cfg is not None
        self.convs = nn.ModuleList([_ConvWithTimeParams(256, 256, with_norm) for _ in range(num_convs)])
        if concat_input:          # built by the reference, never called by its _forward_feature (:285-299)
            self.conv_cat = _ConvParams(512, 256, kernel_size, with_norm)
        self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)
        self._ws = None

    def conv_array(self):
        """-> (ddp_fcn_conv array of the head's ConvWithTimeModules, list keeping the referenced tensors alive)"""
        keep = []

        def ptr(t):
            if t is None:
                return None
            t = t.detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()
        arr = (_lib.DdpFcnConv * max(self.num_convs, 1))()
        for i, m in enumerate(self.convs):
            bn = getattr(m, 'bn', None)
            arr[i].conv_w, arr[i].conv_b = ptr(m.conv.weight), ptr(m.conv.bias)
            arr[i].bn_w = ptr(bn.weight) if bn is not None else None
            arr[i].bn_b = ptr(bn.bias) if bn is not None else None
            arr[i].bn_mean = ptr(bn.running_mean) if bn is not None else None
            arr[i].bn_var = ptr(bn.running_var) if bn is not None else None
            arr[i].bn_eps = bn.eps if bn is not None else 0.0
            arr[i].time_w, arr[i].time_b = ptr(m.time_mlp[1].weight), ptr(m.time_mlp[1].bias)
        return arr, keep

    def forward(self, inputs, times):
        x = inputs[self.in_index] if isinstance(inputs, (list, tuple)) else inputs
        if not x.is_cuda:
            raise _lib.DdpError('FCNHeadWithTime: CUDA tensors only (ddp_amd has no CPU path)')
        if self.training:
            raise NotImplementedError('training is out of scope of ddp_amd (SURVEY.md §8)')
        x = x.contiguous().float()
        R, c, h, w = x.shape
        assert c == 256
        temb = None
        if times is not None:
            temb = times.reshape(-1, 1024)
            if temb.shape[0] > 1 and not torch.equal(temb[:1].expand_as(temb), temb):
                raise NotImplementedError('per-sample time embeddings (the samplers broadcast one time to the batch)')
            temb = temb[0].contiguous().float()
        lib = _lib.load()
        arr, keep = self.conv_array()

        def ptr(t):
            t = t.detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()
        nbytes = C.c_size_t(0)
        _lib.check(lib.ddp_fcn_head_workspace(R, h, w, self.num_classes, C.byref(nbytes)))
        if self._ws is None or self._ws.numel() * 4 < nbytes.value or self._ws.device != x.device:
            self._ws = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=x.device)
        out = torch.empty((R, self.num_classes, h, w), dtype=torch.float32, device=x.device)
        _lib.check(lib.ddp_fcn_head_forward(arr, self.num_convs, self.dilation,
                                            ptr(self.conv_seg.weight.reshape(self.num_classes, 256)), ptr(self.conv_seg.bias),
                                            self.num_classes, x.data_ptr(), temb.data_ptr() if temb is not None else None,
                                            R, h, w, out.data_ptr(), self._ws.data_ptr(),
                                            torch.cuda.current_stream(x.device).cuda_stream))
        return out

    def forward_test(self, inputs, times, img_metas=None, test_cfg=None):
        return self.forward(inputs, times)

    def forward_train(self, *a, **k):
        raise NotImplementedError('training is out of scope of ddp_amd (SURVEY.md §8)')
