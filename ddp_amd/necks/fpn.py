"""MI355X drop-in for the reference's ``FPN`` neck as the DDP configs build it
(segmentation/mmseg/models/necks/fpn.py:12-213): 4 levels in, 4 levels out, ``norm_cfg=GN(32)``,
``act_cfg=None``, nearest-neighbour top-down path, no extra levels.

Same registry name, constructor kwargs, ``state_dict`` keys (``lateral_convs.l.conv.weight``, ``lateral_convs.l.gn.*``,
``fpn_convs.l.conv.weight``, ``fpn_convs.l.gn.*``) and ``forward(inputs) -> tuple`` as the reference class; the work
happens in ``ddp_neck_fpn`` of libddp_mi355x.so: activations in the fp32 fragment layout of the persistent stream-GEMM
kernel (``b3::k_layer`` MODE 5), which runs the four lateral 1x1 convolutions in one launch and the four 3x3 output
convolutions (implicit GEMM: the taps are shifted fragment loads) in another, GroupNorm partial sums in its epilogue
(deterministic fixed-order reductions).  CUDA tensors only - no CPU path.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from ..registry import NECKS


class _ConvGN(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2, bias=False)
        self.gn = nn.GroupNorm(32, cout)
        nn.init.xavier_uniform_(self.conv.weight)


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=False, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest'), init_cfg=None):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        ok = (len(in_channels) == 4 and out_channels == 256 and num_outs == 4 and start_level == 0 and end_level in (-1, 4)
              and not add_extra_convs and not no_norm_on_lateral and conv_cfg is None and act_cfg is None
              and norm_cfg is not None and norm_cfg.get('type') == 'GN' and norm_cfg.get('num_groups') == 32
              and dict(upsample_cfg) == dict(mode='nearest') and all(c % 32 == 0 and 64 <= c <= 4096 for c in in_channels))
        if not ok:
            raise NotImplementedError('ddp_amd FPN: the DDP configuration only (4 levels -> 4 x 256, GN(32), no activation, '
                                      'nearest top-down, input channels multiples of 32 in [64, 4096])')
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.lateral_convs = nn.ModuleList([_ConvGN(c, 256, 1) for c in in_channels])
        self.fpn_convs = nn.ModuleList([_ConvGN(256, 256, 3) for _ in in_channels])
        self._ws = None
        self._ws_weights = None      # (weights version, workspace pointer) whose stage images the workspace holds

    def _levels(self, inputs):
        """(xs, ddp_fpn_level[4], kept tensors) for a call"""
        assert len(inputs) == 4
        for t in inputs:
            if not t.is_cuda:
                raise _lib.DdpError('FPN: CUDA tensors only (ddp_amd has no CPU path)')
        xs = [t.contiguous().float() for t in inputs]
        B = xs[0].shape[0]
        keep = []

        def ptr(t):
            t = t.detach().float().contiguous()
            keep.append(t)
            return t.data_ptr()
        lv = (_lib.DdpFpnLevel * 4)()
        for l, x in enumerate(xs):
            assert x.shape[0] == B and x.shape[1] == self.in_channels[l]
            lat, out = self.lateral_convs[l], self.fpn_convs[l]
            lv[l].lat_w, lv[l].lat_gn_w, lv[l].lat_gn_b = ptr(lat.conv.weight), ptr(lat.gn.weight), ptr(lat.gn.bias)
            lv[l].out_w, lv[l].out_gn_w, lv[l].out_gn_b = ptr(out.conv.weight), ptr(out.gn.weight), ptr(out.gn.bias)
            lv[l].in_channels, lv[l].h, lv[l].w = x.shape[1], x.shape[2], x.shape[3]
        return xs, lv, keep

    def forward(self, inputs):
        xs, lv, keep = self._levels(inputs)
        B = xs[0].shape[0]
        lib = _lib.load()
        nbytes = C.c_size_t(0)
        _lib.check(lib.ddp_neck_fpn_workspace(lv, B, C.byref(nbytes)))
        if self._ws is None or self._ws.numel() * 4 < nbytes.value or self._ws.device != xs[0].device:
            self._ws = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=xs[0].device)
        # the weight region at the head of the workspace (stage images of the stream GEMM) is built once per set of weights
        tag = (sum(p._version for p in self.parameters()), self._ws.data_ptr(), tuple(k.data_ptr() for k in keep))
        flags = _lib.NECK_WEIGHTS_READY if tag == self._ws_weights else 0
        outs = [torch.empty((B, 256, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device) for x in xs]
        pin = (C.c_void_p * 4)(*[x.data_ptr() for x in xs])
        pout = (C.c_void_p * 4)(*[o.data_ptr() for o in outs])
        _lib.check(lib.ddp_neck_fpn(lv, B, pin, pout, flags, self._ws.data_ptr(), torch.cuda.current_stream(xs[0].device).cuda_stream))
        self._ws_weights = tag
        return tuple(outs)
