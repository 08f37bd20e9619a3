"""MI355X drop-in for the reference's ``MultiStageMerging`` neck
(segmentation/mmseg/models/necks/multi_stage_merging.py:11-52): the step that produces the frozen
feature ``x`` of the sampling loop from the four FPN levels.

Same registry name, constructor kwargs, ``state_dict`` keys (``down.conv.weight`` (256,1024,1,1), ``down.gn.weight``,
``down.gn.bias``) and ``forward(inputs) -> [out]`` as the reference class.  The work happens in
``ddp_neck_msm`` of libddp_mi355x.so: resize + concat are written directly as the GEMM operand, the 1x1 conv runs on
the bf16x3 MFMA engine, GroupNorm is a deterministic two-stage reduction.  CUDA tensors only - no CPU path.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from ..registry import NECKS


class _Down(nn.Module):
    """parameter container with the reference's ConvModule key names (conv.weight, gn.weight, gn.bias)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.gn = nn.GroupNorm(32, cout)


@NECKS.register_module()
class MultiStageMerging(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 align_corners=False, init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        if list(in_channels) != [256, 256, 256, 256] or out_channels != 256 or kernel_size != 1:
            raise NotImplementedError('ddp_amd MultiStageMerging: four 256-channel levels -> 256, 1x1 conv (the DDP configs)')
        if not norm_cfg or norm_cfg.get('type') != 'GN' or norm_cfg.get('num_groups') != 32 or act_cfg is not None:
            raise NotImplementedError('ddp_amd MultiStageMerging: norm_cfg=GN(32), act_cfg=None (the DDP configs)')
        self.in_channels = list(in_channels)
        self.out_channels = out_channels
        self.align_corners = align_corners
        self.down = _Down(sum(in_channels), out_channels)
        nn.init.xavier_uniform_(self.down.conv.weight)
        self._ws = None
        self._ws_weights = None      # (weights version, workspace pointer) whose stage images the workspace holds

    def forward(self, inputs):
        assert len(inputs) == 4
        for t in inputs:
            if not t.is_cuda:
                raise _lib.DdpError('MultiStageMerging: CUDA tensors only (ddp_amd has no CPU path)')
        lv = [t.contiguous().float() for t in inputs]
        B = lv[0].shape[0]
        for t in lv:
            assert t.shape[0] == B and t.shape[1] == 256
        lib = _lib.load()
        lh = (C.c_int * 4)(*[t.shape[2] for t in lv])
        lw = (C.c_int * 4)(*[t.shape[3] for t in lv])
        nbytes = C.c_size_t(0)
        _lib.check(lib.ddp_neck_msm_workspace(B, lh, lw, C.byref(nbytes)))
        if self._ws is None or self._ws.numel() * 4 < nbytes.value or self._ws.device != lv[0].device:
            self._ws = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=lv[0].device)
        ptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in lv])
        out = torch.empty((B, 256, lv[0].shape[2], lv[0].shape[3]), dtype=torch.float32, device=lv[0].device)
        w = self.down.conv.weight.detach().reshape(256, 1024).contiguous().float()
        tag = (sum(p._version for p in self.parameters()), self._ws.data_ptr(), w.data_ptr())
        flags = _lib.NECK_WEIGHTS_READY if tag == self._ws_weights else 0
        _lib.check(lib.ddp_neck_msm(ptrs, lh, lw, B, w.data_ptr(), self.down.gn.weight.detach().float().contiguous().data_ptr(),
                                    self.down.gn.bias.detach().float().contiguous().data_ptr(), int(bool(self.align_corners)), flags,
                                    out.data_ptr(), self._ws.data_ptr(),
                                    torch.cuda.current_stream(lv[0].device).cuda_stream))
        self._ws_weights = tag
        return [out]
