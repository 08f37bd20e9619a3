"""The neck list of the DDP configs, ``neck=[dict(type='FPN', ...), dict(type='MultiStageMerging', ...)]``
(segmentation/configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py; built by segmentors/ddp.py into an ``nn.Sequential``): the
same container - same ``state_dict`` keys ``0.*`` / ``1.*`` - whose forward runs the pair as ONE C entry
(``ddp_neck_fpn_msm``): the four FPN outputs stay in the GEMM operand layout and feed the merging directly; their NCHW form
is never produced.  Any other composition (or CPU tensors, which raise inside the members) runs member by member."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .fpn import FPN
from .multi_stage_merging import MultiStageMerging


class NeckChain(nn.Sequential):
    def __init__(self, *mods):
        super().__init__(*mods)
        self._ws = None
        self._ws_weights = None

    def fused(self):
        return len(self) == 2 and isinstance(self[0], FPN) and isinstance(self[1], MultiStageMerging)

    def forward(self, inputs):
        if not self.fused() or not all(torch.is_tensor(t) and t.is_cuda for t in inputs):
            return super().forward(inputs)
        fpn, msm = self[0], self[1]
        xs, lv, keep = fpn._levels(inputs)
        B = xs[0].shape[0]
        lib = _lib.load()
        nbytes = C.c_size_t(0)
        _lib.check(lib.ddp_neck_fpn_msm_workspace(lv, B, C.byref(nbytes)), lib)
        if self._ws is None or self._ws.numel() * 4 < nbytes.value or self._ws.device != xs[0].device:
            self._ws = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=xs[0].device)
        w = msm.down.conv.weight.detach().reshape(256, 1024).contiguous().float()
        gw, gb = msm.down.gn.weight.detach().float().contiguous(), msm.down.gn.bias.detach().float().contiguous()
        tag = (sum(p._version for p in self.parameters()), self._ws.data_ptr(), tuple(k.data_ptr() for k in keep), w.data_ptr())
        flags = _lib.NECK_WEIGHTS_READY if tag == self._ws_weights else 0
        out = torch.empty((B, 256, xs[0].shape[2], xs[0].shape[3]), dtype=torch.float32, device=xs[0].device)
        pin = (C.c_void_p * 4)(*[x.data_ptr() for x in xs])
        _lib.check(lib.ddp_neck_fpn_msm(lv, B, pin, w.data_ptr(), gw.data_ptr(), gb.data_ptr(), int(bool(msm.align_corners)), flags,
                                        out.data_ptr(), self._ws.data_ptr(), torch.cuda.current_stream(xs[0].device).cuda_stream), lib)
        self._ws_weights = tag
        return [out]
