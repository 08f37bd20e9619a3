"""DDPEngine: owns the device-side state of one configured sampling problem and calls the C ABI.

PyTorch is plumbing here: it allocates device memory (weights blob, workspace, inputs/outputs) and
provides the current HIP stream; every FLOP of the loop runs in libddp_mi355x.so.
"""
import ctypes as C

import torch

from . import _lib
from . import schedule

TASKS = {'seg': _lib.TASK_SEG, 'depth': _lib.TASK_DEPTH, 'bev': _lib.TASK_BEV}
SAMPLERS = {'ddim': _lib.SAMPLER_DDIM, 'ddpm': _lib.SAMPLER_DDPM}
GEMM_MODES = {'f32': _lib.GEMM_F32_MFMA, 'bf16x3': _lib.GEMM_BF16X3}


def default_gemm_mode():
    """'bf16x3' (fp32-accurate 3-way bf16 split on the bf16 matrix cores) unless DDP_GEMM_MODE=f32 selects the
    exact f32-input MFMA path."""
    import os
    return os.environ.get('DDP_GEMM_MODE', 'bf16x3')

_LAYER_KEYS = {
    'sampling_offsets_w': 'attentions.0.sampling_offsets.weight', 'sampling_offsets_b': 'attentions.0.sampling_offsets.bias',
    'attention_weights_w': 'attentions.0.attention_weights.weight', 'attention_weights_b': 'attentions.0.attention_weights.bias',
    'value_proj_w': 'attentions.0.value_proj.weight', 'value_proj_b': 'attentions.0.value_proj.bias',
    'output_proj_w': 'attentions.0.output_proj.weight', 'output_proj_b': 'attentions.0.output_proj.bias',
    'ffn0_w': 'ffns.0.layers.0.0.weight', 'ffn0_b': 'ffns.0.layers.0.0.bias',
    'ffn1_w': 'ffns.0.layers.1.weight', 'ffn1_b': 'ffns.0.layers.1.bias',
    'norm0_w': 'norms.0.weight', 'norm0_b': 'norms.0.bias', 'norm1_w': 'norms.1.weight', 'norm1_b': 'norms.1.bias',
    'time_w': 'time_mlp.1.weight', 'time_b': 'time_mlp.1.bias',
}


def hot_path_keys(task, num_layers, head_prefix='decode_head.'):
    """(struct field, state_dict key) pairs of the hot path (SURVEY.md §8b checkpoint layout)."""
    conv = 'down.conv' if task == 'depth' else 'transform.conv'
    top = [('transform_w', conv + '.weight'), ('transform_b', conv + '.bias'),
           ('time_freq', 'time_mlp.0.weights'), ('time1_w', 'time_mlp.1.weight'), ('time1_b', 'time_mlp.1.bias'),
           ('time3_w', 'time_mlp.3.weight'), ('time3_b', 'time_mlp.3.bias')]
    if task != 'depth':
        top.append(('embedding', 'embedding_table.weight'))
    hc = 'conv_depth' if task == 'depth' else 'conv_seg'
    top += [('head_w', head_prefix + hc + '.weight'), ('head_b', head_prefix + hc + '.bias')]
    layers = []
    for l in range(num_layers):
        p = f'{head_prefix}encoder.layers.{l}.'
        layers.append([(f, p + k) for f, k in _LAYER_KEYS.items()])
    return top, layers


def count_layers(state_dict, head_prefix='decode_head.'):
    n = 0
    while f'{head_prefix}encoder.layers.{n}.norms.0.weight' in state_dict:
        n += 1
    return n


class PackedWeights:
    """All hot-path parameters in ONE flat fp32 device buffer (256-byte aligned sub-tensors): a
    single allocation, a single RCCL broadcast, and stable pointers for the ``ddp_weights`` table."""

    def __init__(self, state_dict, task, num_layers, device, head_prefix='decode_head.'):
        top, layers = hot_path_keys(task, num_layers, head_prefix)
        entries = []
        for f, k in top:
            entries.append((None, f, k))
        for l, lk in enumerate(layers):
            for f, k in lk:
                entries.append((l, f, k))
        offs, total = {}, 0
        for l, f, k in entries:
            if k not in state_dict:
                if f in ('time_w', 'time_b'):     # layer built without use_time_mlp
                    continue
                raise KeyError(f'hot-path parameter {k!r} missing from state_dict')
            n = state_dict[k].numel()
            offs[(l, f)] = (total, k)
            total += (n + 63) // 64 * 64
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        for (l, f), (o, k) in offs.items():
            flat[o:o + state_dict[k].numel()].copy_(state_dict[k].detach().reshape(-1))
        self.flat = flat
        self.offsets = offs
        self.task = task
        self.num_layers = num_layers
        self.struct = self._build_struct()

    def _build_struct(self):
        w = _lib.DdpWeights()
        base = self.flat.data_ptr()
        for (l, f), (o, _) in self.offsets.items():
            tgt = w if l is None else w.layers[l]
            setattr(tgt, f, base + 4 * o)
        return w

    def broadcast(self, src=0, force=False):
        """Replicate the frozen weights from rank ``src`` to every rank (RCCL over xGMI; the only
        collective of the inference path - SURVEY.md §8e).  ``force``: run the collective at world size 1 too."""
        from .parallel import broadcast_weights
        broadcast_weights(self.flat, src=src, force=force)


class DDPEngine:
    """One configured problem: (task, sizes, schedule) + weights + workspace."""

    def __init__(self, state_dict, task='seg', *, h, w, batch=1, randsteps=1, timesteps=3, num_classes=150,
                 feat_channels=256, bit_scale=0.01, time_difference=1, sample_range0=0.0, noise_schedule='cosine',
                 sampler='ddim', accumulation=False, min_depth=1e-3, max_depth=80.0, threshold=0.5,
                 head_hw=None, bev_input_scope=None, bev_output_scope=None, device=None, head_prefix='decode_head.',
                 weights=None, gemm=None, fused_layer=None, fused_prologue=None, lib_path=None, record_x0=False,
                 gather_guess_zero=False, force_x0=False, fused_tail=None, nchw_head=None, depth_scale_up=False,
                 depth_use_eps=True):
        self.lib = _lib.load(lib_path)
        if not torch.cuda.is_available():
            raise _lib.DdpError('no HIP device visible: ddp_amd has no CPU path')
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.task = task
        num_layers = count_layers(state_dict, head_prefix) if weights is None else weights.num_layers
        self.weights = weights if weights is not None else PackedWeights(state_dict, task, num_layers, self.device,
                                                                        head_prefix)
        cfg = _lib.DdpCfg()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.task = TASKS[task]
        cfg.sampler = SAMPLERS[sampler]
        cfg.batch, cfg.randsteps, cfg.timesteps, cfg.num_layers = batch, randsteps, timesteps, num_layers
        cfg.num_classes = 1 if task == 'depth' else num_classes
        cfg.feat_channels = feat_channels
        cfg.h, cfg.w = h, w
        if task == 'bev':
            if bev_input_scope is None or bev_output_scope is None:
                raise ValueError('bev needs bev_input_scope / bev_output_scope (grid_transform of the head)')
            out_sizes = []
            for a, ((imin, imax, _), (omin, omax, ostep)) in enumerate(zip(bev_input_scope, bev_output_scope)):
                n = int(torch.arange(omin + ostep / 2, omax, ostep).numel())
                out_sizes.append(n)
                cfg.bev_in_min[a], cfg.bev_in_max[a] = imin, imax
                cfg.bev_out_first[a], cfg.bev_out_step[a] = omin + ostep / 2, ostep
            cfg.head_h, cfg.head_w = out_sizes
        else:
            cfg.head_h, cfg.head_w = (h, w) if head_hw is None else head_hw
        self.gemm = gemm if gemm is not None else default_gemm_mode()
        cfg.gemm_mode = GEMM_MODES[self.gemm]
        # diagnostics (A/B runs, tests of the unfused kernels): DDP_LAYER_FUSED=0 / DDP_PROLOGUE_FUSED=0 or the kwargs
        import os
        if fused_layer is None:
            fused_layer = os.environ.get('DDP_LAYER_FUSED', '1') != '0'
        if fused_prologue is None:
            fused_prologue = os.environ.get('DDP_PROLOGUE_FUSED', '1') != '0'
        if fused_tail is None:     # DDP_TAIL_FUSED=0: the step boundary as the launches it was fused from (DDP_FLAG_UNFUSED_TAIL: seg - the last
                                   # layer and the tail as two kernels; depth / bev - the round-5 GEMM heads and update kernels; A/B runs, tests)
            fused_tail = os.environ.get('DDP_TAIL_FUSED', '1') != '0'
        if nchw_head is None:      # DDP_NCHW_HEAD=0: the first step's head through the SB conversions + x-projection GEMM + MODE 2
            nchw_head = os.environ.get('DDP_NCHW_HEAD', '1') != '0'
        cfg.flags = ((0 if fused_layer else _lib.FLAG_UNFUSED_LAYER) | (0 if fused_prologue else _lib.FLAG_UNFUSED_PROLOGUE) |
                     (0 if fused_tail else _lib.FLAG_UNFUSED_TAIL) | (0 if nchw_head else _lib.FLAG_SB_HEAD))
        if task == 'depth':        # head variants of depth_pred (decode_head.py:252-262)
            cfg.flags |= (_lib.FLAG_DEPTH_SCALE_UP if depth_scale_up else 0) | (0 if depth_use_eps else _lib.FLAG_DEPTH_NO_EPS)
        if record_x0:
            cfg.flags |= _lib.FLAG_RECORD_X0
        if force_x0:               # test instrument (seg): teacher forcing, see set_x0_decisions()
            if task != 'seg':
                raise ValueError('force_x0 is a segmentation test instrument')
            cfg.flags |= _lib.FLAG_FORCE_X0
        if gather_guess_zero:      # diagnostic: forces the LDS gather's refill branch (identical results)
            cfg.flags |= _lib.FLAG_GATHER_GUESS_ZERO
        self.fused_layer = bool(fused_layer)
        cfg.accumulation = int(bool(accumulation))
        cfg.bit_scale, cfg.min_depth, cfg.max_depth, cfg.threshold = bit_scale, min_depth, max_depth, threshold
        self.cfg = cfg
        self.sampler = sampler
        recs = schedule.step_records(task, timesteps, time_difference, sample_range0, noise_schedule, sampler)
        self.steps = (_lib.DdpStep * timesteps)()
        for i, r in enumerate(recs):
            for k, v in r.items():
                setattr(self.steps[i], k, v)
        nbytes = C.c_size_t(0)
        _lib.check(self.lib.ddp_query_workspace(C.byref(cfg), C.byref(nbytes)), self.lib)
        self.workspace = torch.empty(nbytes.value // 4, dtype=torch.float32, device=self.device)
        cbytes = C.c_size_t(0)
        _lib.check(self.lib.ddp_query_const_workspace(C.byref(cfg), C.byref(cbytes)), self.lib)
        self._const_floats = cbytes.value // 4       # model region: a prefix of the workspace, independent of the geometry
        self._prepared = False
        self.geometry_changes = 0

    # ------------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def geometry(self):
        return (self.cfg.batch, self.cfg.h, self.cfg.w)

    def set_geometry(self, batch=None, h=None, w=None, grow=1.25):
        """Switch the engine to another (batch, h, w) WITHOUT touching anything derived from the weights: the reference's
        test protocol is one image per call with a new size almost every call (segmentation/tools/test.py:214-219).
        The model region of the workspace (split weight planes, weight streams, LUTs, time / FiLM vectors) is kept - moved
        with one device-to-device copy if the buffer has to grow - and only ``ddp_prepare_geometry`` runs (positional
        tables + the zero border of the padded value maps)."""
        c = self.cfg
        nb = c.batch if batch is None else int(batch)
        nh = c.h if h is None else int(h)
        nw = c.w if w is None else int(w)
        if (nb, nh, nw) == (c.batch, c.h, c.w):
            return self
        # everything is validated / grown / prepared on a COPY of the cfg; self.cfg changes only on success, so a failed
        # switch (too many tokens, out of memory, a launch error) leaves the engine on its old, still prepared geometry
        n = _lib.DdpCfg.from_buffer_copy(c)
        n.batch, n.h, n.w = nb, nh, nw
        if self.task != 'bev' and (c.head_h, c.head_w) == (c.h, c.w):
            n.head_h, n.head_w = nh, nw             # the decoder grid follows the map (bev: the grid transform's output scope;
                                                    # a constructor head_hw that differs from the map is kept as given)
        nbytes = C.c_size_t(0)
        _lib.check(self.lib.ddp_query_workspace(C.byref(n), C.byref(nbytes)), self.lib)
        need = nbytes.value // 4
        ws = self.workspace
        if need > ws.numel():
            ws = torch.empty(int(need * grow), dtype=torch.float32, device=self.device)
            if self._prepared:
                ws[:self._const_floats].copy_(self.workspace[:self._const_floats])
        if self._prepared:
            with torch.cuda.device(self.device):
                _lib.check(self.lib.ddp_prepare_geometry(C.byref(n), ws.data_ptr(), self._stream()), self.lib)
        self.cfg, self.workspace = n, ws
        self.geometry_changes += 1
        self._x0_set = False       # force_x0: the decisions buffer belongs to the geometry region
        return self

    def out_shape(self):
        c = self.cfg
        if self.task == 'depth':
            return (c.batch, 1, c.h, c.w)
        return (c.batch, c.num_classes, c.head_h, c.head_w)

    def prepare(self):
        with torch.cuda.device(self.device):       # launches go to the CURRENT device: make it the workspace's
            _lib.check(self.lib.ddp_prepare(C.byref(self.cfg), C.byref(self.weights.struct), self.steps,
                                            self.workspace.data_ptr(), self._stream()), self.lib)
        self._prepared = True

    def sample(self, x, noise, step_noise=None, out=None):
        """x (B,Cx,h,w); noise (B,r,Cm,h,w); step_noise (K,B,r,Cm,h,w) for ddpm -> output tensor."""
        c = self.cfg
        cm = 1 if self.task == 'depth' else 256
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert tuple(x.shape) == (c.batch, c.feat_channels, c.h, c.w), (tuple(x.shape), (c.batch, c.feat_channels, c.h, c.w))
        assert noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous()
        assert noise.numel() == c.batch * c.randsteps * cm * c.h * c.w
        if self.sampler == 'ddpm':
            assert step_noise is not None and step_noise.is_contiguous() and step_noise.numel() == c.timesteps * noise.numel()
        if not self._prepared:
            self.prepare()
        if (c.flags & _lib.FLAG_FORCE_X0) and not getattr(self, '_x0_set', False):
            raise _lib.DdpError('force_x0 engine: call set_x0_decisions() before sample()')
        if out is None:
            out = torch.empty(self.out_shape(), dtype=torch.float32, device=self.device)
        if x.device != self.device or noise.device != self.device or out.device != self.device:
            raise _lib.DdpError(f'engine lives on {self.device}: x / noise / out must be on the same device')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ddp_sample(C.byref(c), C.byref(self.weights.struct), self.steps, x.data_ptr(),
                                           noise.data_ptr(), step_noise.data_ptr() if step_noise is not None else None,
                                           out.data_ptr(), self.workspace.data_ptr(), self._stream()), self.lib)
        return out

    def capture(self, x, noise, step_noise=None):
        """One ``sample()`` call as a hipGraph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm): ``ddp_sample`` neither
        synchronises nor touches host memory after enqueue - its ~45 launches (kernels, device-to-device copies, memsets) go to
        the stream it is given, the schedule scalars travel as kernel arguments - so the whole K-step loop records under stream
        capture and replays with ONE host call.  What that buys is host time: one image per call (the reference's protocol, and
        the strong-scaling shard) is ~4 ms of GPU work behind ~45 launches, issued by 8 ranks that share the node's cores with
        their data loaders.  Inputs are copied into the graph's own static buffers at replay; -> ``SampleGraph``."""
        if not self._prepared:
            self.prepare()
        sx, sn = x.detach().clone(), noise.detach().clone()
        ssn = step_noise.detach().clone() if step_noise is not None else None
        out = torch.empty(self.out_shape(), dtype=torch.float32, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):              # warm-up on the capture stream: one-time attribute calls happen here
            self.sample(sx, sn, ssn, out=out)
        cur.wait_stream(side)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            self.sample(sx, sn, ssn, out=out)
        return SampleGraph(self, g, sx, sn, ssn, out)

    def _x0_view(self, which):
        c = self.cfg
        p = C.c_void_p()
        _lib.check(self.lib.ddp_x0_trace(C.byref(c), self.workspace.data_ptr(), C.byref(p)), self.lib)
        n = c.timesteps * c.batch * c.randsteps * c.head_h * c.head_w
        off = p.value - self.workspace.data_ptr() + which * n
        return self.workspace.view(torch.uint8)[off:off + n].view(c.timesteps, c.batch * c.randsteps, c.head_h, c.head_w)

    def x0_trace(self):
        """(K, B*r, h, w) uint8: the argmax class every step of the LAST sample() call found (needs record_x0=True or
        force_x0=True; record_x0: that is also what the step fed back, force_x0: the step fed back set_x0_decisions()'s)."""
        return self._x0_view(1 if self.cfg.flags & _lib.FLAG_FORCE_X0 else 0).clone()

    def set_x0_decisions(self, decisions):
        """force_x0=True (test instrument, DDP_FLAG_FORCE_X0): the classes (K, B*r, h, w) every step of the following sample()
        calls feeds back instead of its own argmax - e.g. the decisions a reference run recorded (tests/golden/full_*.npz)."""
        if not self.cfg.flags & _lib.FLAG_FORCE_X0:
            raise _lib.DdpError('set_x0_decisions needs an engine built with force_x0=True')
        v = self._x0_view(0)
        d = decisions.to(device=self.device, dtype=torch.uint8).reshape(v.shape)
        if int(d.max()) >= self.cfg.num_classes:
            raise ValueError('decision outside [0, num_classes)')
        v.copy_(d)
        self._x0_set = True

    def head_forward(self, feat, temb):
        """DeformableHeadWithTime.forward on (R,256,h,w) + (1|R,1024) time embedding."""
        c = self.cfg
        R = c.batch * c.randsteps
        assert feat.is_cuda and feat.is_contiguous() and tuple(feat.shape) == (R, 256, c.h, c.w)
        t = None
        if temb is not None:
            t = temb.reshape(-1, 1024)[0].contiguous()
        shape = (R, 1, c.h, c.w) if self.task == 'depth' else (R, c.num_classes, c.head_h, c.head_w)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ddp_head_forward(C.byref(c), C.byref(self.weights.struct), feat.data_ptr(),
                                                 t.data_ptr() if t is not None else None, out.data_ptr(),
                                                 self.workspace.data_ptr(), self._stream()), self.lib)
        self._prepared = False      # head_forward rewrites the FiLM slot of step 0
        return out


class SampleGraph:
    """A captured ``DDPEngine.sample`` call (``DDPEngine.capture``).  ``replay(x, noise)`` copies the inputs into the graph's
    static buffers (stream-ordered, on the current stream) and launches the graph; the returned tensor is the graph's static
    output buffer (clone it to keep a result across replays).  Valid for the geometry and schedule it was captured with."""

    def __init__(self, engine, graph, x, noise, step_noise, out):
        self.engine, self.graph = engine, graph
        self.x, self.noise, self.step_noise, self.out = x, noise, step_noise, out
        self.geometry = engine.geometry()
        # the graph has the workspace's ADDRESS baked into every launch: keep that allocation alive for as long as the graph
        # exists (a replay can then never touch freed memory) and refuse to replay once the engine has moved to another one
        self.workspace = engine.workspace
        self.workspace_ptr = engine.workspace.data_ptr()

    def replay(self, x=None, noise=None, step_noise=None):
        eng = self.engine
        if eng.geometry() != self.geometry:
            raise _lib.DdpError(f'graph captured for geometry {self.geometry}, engine is now at {eng.geometry()}')
        if eng.workspace.data_ptr() != self.workspace_ptr:
            raise _lib.DdpError('stale graph: the engine\'s workspace was reallocated after the capture (set_geometry grew it); '
                                'capture again')
        if not eng._prepared:      # head_forward() rewrote the FiLM slot of step 0: restore the constants (same addresses)
            eng.prepare()
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if noise is not None:
            self.noise.copy_(noise.reshape(self.noise.shape), non_blocking=True)
        if step_noise is not None:
            self.step_noise.copy_(step_noise.reshape(self.step_noise.shape), non_blocking=True)
        self.graph.replay()
        return self.out


class FcnSamplerEngine:
    """The K-step sampler with ``FCNHeadWithTime`` as decode head (``ddp_sample_fcn``): same inputs,
    schedule and outputs as ``DDPEngine`` for task 'seg'.  ``head`` is a ``ddp_amd.FCNHeadWithTime`` (parameter holder)."""

    def __init__(self, state_dict, head, *, h, w, batch=1, randsteps=1, timesteps=3, num_classes=150, bit_scale=0.01,
                 time_difference=1, sample_range0=0.0, noise_schedule='cosine', sampler='ddim', accumulation=False,
                 device=None, head_prefix='decode_head.', lib_path=None):
        self.lib = _lib.load(lib_path)
        if not torch.cuda.is_available():
            raise _lib.DdpError('no HIP device visible: ddp_amd has no CPU path')
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.weights = PackedWeights(state_dict, 'seg', 0, self.device, head_prefix)      # transform, time_mlp, embedding, conv_seg
        self.head = head
        self.convs, self._keep = head.conv_array()
        for t in self._keep:
            if t.device != self.device:
                raise _lib.DdpError(f'FCN head parameters live on {t.device}, engine on {self.device}')
        cfg = _lib.DdpCfg()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.task, cfg.sampler = _lib.TASK_SEG, SAMPLERS[sampler]
        cfg.batch, cfg.randsteps, cfg.timesteps, cfg.num_layers = batch, randsteps, timesteps, 0
        cfg.num_classes, cfg.feat_channels = num_classes, 256
        cfg.h = cfg.head_h = h
        cfg.w = cfg.head_w = w
        cfg.gemm_mode, cfg.flags = _lib.GEMM_BF16X3, 0
        cfg.accumulation, cfg.bit_scale = int(bool(accumulation)), bit_scale
        self.cfg, self.sampler = cfg, sampler
        recs = schedule.step_records('seg', timesteps, time_difference, sample_range0, noise_schedule, sampler)
        self.steps = (_lib.DdpStep * timesteps)()
        for i, r in enumerate(recs):
            for k, v in r.items():
                setattr(self.steps[i], k, v)
        nbytes = C.c_size_t(0)
        _lib.check(self.lib.ddp_sample_fcn_workspace(C.byref(cfg), head.num_convs, head.dilation, C.byref(nbytes)), self.lib)
        self.workspace = torch.empty(nbytes.value // 4 + 64, dtype=torch.float32, device=self.device)
        self._prepared = False

    def prepare(self):
        """``ddp_prepare_fcn``: everything that depends on weights and schedule only (time embeddings, x0 table, concat-conv
        column blocks, per (step, conv) FiLM -> folded affine -> scaled, split weights as stage images) once per engine; every
        ``sample()`` afterwards runs with ``DDP_FLAG_FCN_PREPARED`` and launches none of those kernels."""
        c = self.cfg
        c.flags &= ~_lib.FLAG_FCN_PREPARED
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ddp_prepare_fcn(C.byref(c), C.byref(self.weights.struct), self.convs, self.head.num_convs,
                                                self.head.dilation, self.steps, self.workspace.data_ptr(),
                                                torch.cuda.current_stream(self.device).cuda_stream), self.lib)
        c.flags |= _lib.FLAG_FCN_PREPARED
        self._prepared = True

    def sample(self, x, noise, step_noise=None, out=None):
        c = self.cfg
        if not self._prepared:
            self.prepare()
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (c.batch, 256, c.h, c.w)
        assert noise.is_cuda and noise.is_contiguous() and noise.numel() == c.batch * c.randsteps * 256 * c.h * c.w
        if self.sampler == 'ddpm':
            assert step_noise is not None and step_noise.is_contiguous() and step_noise.numel() == c.timesteps * noise.numel()
        if out is None:
            out = torch.empty((c.batch, c.num_classes, c.h, c.w), dtype=torch.float32, device=self.device)
        if x.device != self.device or noise.device != self.device or out.device != self.device:
            raise _lib.DdpError(f'engine lives on {self.device}: x / noise / out must be on the same device')
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ddp_sample_fcn(C.byref(c), C.byref(self.weights.struct), self.convs, self.head.num_convs,
                                               self.head.dilation, self.steps, x.data_ptr(), noise.data_ptr(),
                                               step_noise.data_ptr() if step_noise is not None else None, out.data_ptr(),
                                               self.workspace.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream),
                       self.lib)
        return out


def seg_postprocess(scores, img_size, crop_size=None, out_size=None, align_corners=False, flip=None, out=None):
    """Fused post-loop epilogue: class map (B,out_h,out_w) uint8 from scores (B,K,h,w).

    Replaces resize -> crop -> resize -> softmax -> flip -> argmax of the reference
    (segmentors/ddp.py:124-128, encoder_decoder.py:229-296).  CUDA tensors only; no CPU path.
    """
    if not scores.is_cuda:
        raise _lib.DdpError('seg_postprocess: scores must be a CUDA tensor (no CPU path)')
    scores = scores.contiguous().float()
    B, K, h, w = scores.shape
    H, W = int(img_size[0]), int(img_size[1])
    ch, cw = (H, W) if crop_size is None else (int(crop_size[0]), int(crop_size[1]))
    oh, ow = (ch, cw) if out_size is None else (int(out_size[0]), int(out_size[1]))
    fl = {None: 0, False: 0, 'horizontal': 1, 'vertical': 2}[flip]
    if out is None:
        out = torch.empty((B, oh, ow), dtype=torch.uint8, device=scores.device)
    lib = _lib.load()
    with torch.cuda.device(scores.device):
        _lib.check(lib.ddp_seg_postprocess(scores.data_ptr(), B, K, h, w, H, W, ch, cw, oh, ow, int(bool(align_corners)), fl,
                                           out.data_ptr(), torch.cuda.current_stream(scores.device).cuda_stream))
    return out


def seg_aug_postprocess(scores_list, metas, out_size, align_corners=False, return_prob=False):
    """Fused multi-scale / flip epilogue (``ddp_seg_aug_postprocess``): class map (B,out_h,out_w) uint8 from the low-resolution
    scores of every augmentation.  Replaces, per augmentation, resize -> crop -> resize -> softmax -> flip of the reference's
    ``inference`` and the running mean + argmax of ``aug_test`` (encoder_decoder.py:229-331) without materialising any
    (B,K,H,W) tensor.  ``metas[i]`` = dict(img_size=(H,W), crop_size=(h,w) or None, flip=None|'horizontal'|'vertical')."""
    if not scores_list or len(scores_list) != len(metas) or len(scores_list) > _lib.MAX_AUGS:
        raise ValueError(f'seg_aug_postprocess: 1..{_lib.MAX_AUGS} augmentations with one meta each')
    dev = scores_list[0].device
    if not scores_list[0].is_cuda:
        raise _lib.DdpError('seg_aug_postprocess: scores must be CUDA tensors (no CPU path)')
    B, K = scores_list[0].shape[:2]
    keep = []
    augs = (_lib.DdpSegAug * len(scores_list))()
    for i, (sc, m) in enumerate(zip(scores_list, metas)):
        sc = sc.contiguous().float()
        if sc.device != dev or sc.shape[0] != B or sc.shape[1] != K:
            raise ValueError('seg_aug_postprocess: every augmentation needs the same batch / classes / device')
        keep.append(sc)
        H, W = int(m['img_size'][0]), int(m['img_size'][1])
        ch, cw = (H, W) if m.get('crop_size') is None else (int(m['crop_size'][0]), int(m['crop_size'][1]))
        augs[i].d_scores = sc.data_ptr()
        augs[i].h, augs[i].w = sc.shape[2], sc.shape[3]
        augs[i].img_h, augs[i].img_w, augs[i].crop_h, augs[i].crop_w = H, W, ch, cw
        augs[i].flip = {None: 0, False: 0, 'horizontal': 1, 'vertical': 2}[m.get('flip')]
    oh, ow = int(out_size[0]), int(out_size[1])
    seg = torch.empty((B, oh, ow), dtype=torch.uint8, device=dev)
    prob = torch.empty((B, K, oh, ow), dtype=torch.float32, device=dev) if return_prob else None
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.ddp_seg_aug_postprocess(augs, len(keep), B, K, oh, ow, int(bool(align_corners)), seg.data_ptr(),
                                               prob.data_ptr() if prob is not None else None,
                                               torch.cuda.current_stream(dev).cuda_stream), lib)
    return (seg, prob) if return_prob else seg


def slide_windows(img_hw, crop_size, stride):
    """The window grid of ``slide_inference`` (encoder_decoder.py:186-206): -> (row origins y1, column origins x1, window size).
    A crop larger than the image shrinks to the image ("the small patch will be used to decode without padding")."""
    H, W = int(img_hw[0]), int(img_hw[1])
    (h_crop, w_crop), (h_stride, w_stride) = crop_size, stride
    h_grids = max(H - h_crop + h_stride - 1, 0) // h_stride + 1
    w_grids = max(W - w_crop + w_stride - 1, 0) // w_stride + 1
    ys = [max(min(i * h_stride + h_crop, H) - h_crop, 0) for i in range(h_grids)]
    xs = [max(min(j * w_stride + w_crop, W) - w_crop, 0) for j in range(w_grids)]
    return ys, xs, (min(h_crop, H), min(w_crop, W))


def seg_slide_postprocess(scores, ys, xs, crop_hw, img_size, keep_size=None, out_size=None, align_corners=False, flip=None,
                          want='seg'):
    """Fused sliding-window epilogue (``ddp_seg_slide_postprocess``).  ``scores`` (n_rows*n_cols, B, K, h, w): the sampler's
    low-resolution output for every window, row-major over the grid (ys x xs).  want: 'seg' -> uint8 class map (B,oh,ow);
    'prob' -> softmax probabilities (B,K,oh,ow) (``inference``); 'scores' -> the window-averaged scores (``slide_inference``).
    Replaces, per window, the resize of ``encode_decode`` and ``preds += F.pad(...)``, then ``/ count_mat``, crop, resize, softmax,
    flip, argmax (encoder_decoder.py:180-227, 266-296).  CUDA tensors only; no CPU path."""
    if not scores.is_cuda:
        raise _lib.DdpError('seg_slide_postprocess: scores must be a CUDA tensor (no CPU path)')
    n_rows, n_cols = len(ys), len(xs)
    scores = scores.contiguous().float()
    if scores.dim() != 5 or scores.shape[0] != n_rows * n_cols or n_rows * n_cols > _lib.MAX_WINDOWS:
        raise ValueError(f'seg_slide_postprocess: scores (windows, B, K, h, w) with windows = {n_rows} x {n_cols} <= {_lib.MAX_WINDOWS}')
    _, B, K, h, w = scores.shape
    H, W = int(img_size[0]), int(img_size[1])
    kh, kw = (H, W) if keep_size is None else (int(keep_size[0]), int(keep_size[1]))
    oh, ow = (kh, kw) if out_size is None else (int(out_size[0]), int(out_size[1]))
    dev = scores.device
    ptrs = (_lib._fp * (n_rows * n_cols))(*[scores[i].data_ptr() for i in range(n_rows * n_cols)])
    y1 = (C.c_int * n_rows)(*[int(v) for v in ys])
    x1 = (C.c_int * n_cols)(*[int(v) for v in xs])
    seg = torch.empty((B, oh, ow), dtype=torch.uint8, device=dev) if want == 'seg' else None
    prob = torch.empty((B, K, oh, ow), dtype=torch.float32, device=dev) if want != 'seg' else None
    mode = {'seg': 0, 'prob': 1, 'scores': 2}[want]
    fl = {None: 0, False: 0, 'horizontal': 1, 'vertical': 2}[flip]
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.ddp_seg_slide_postprocess(ptrs, y1, x1, n_rows, n_cols, B, K, h, w, int(crop_hw[0]), int(crop_hw[1]), H, W, kh, kw,
                                                 oh, ow, int(bool(align_corners)), fl, mode, seg.data_ptr() if seg is not None else None,
                                                 prob.data_ptr() if prob is not None else None,
                                                 torch.cuda.current_stream(dev).cuda_stream), lib)
    return seg if want == 'seg' else prob


def depth_postprocess(depth_list, flips, out_size, min_depth, max_depth, align_corners=False, out=None):
    """Fused post-loop epilogue of the depth toolbox (``ddp_depth_postprocess``): (B,1,out_h,out_w) fp32 from the low-resolution
    maps of every augmentation.  Replaces clamp -> bilinear resize (depth/depth/models/depther/ddp.py:95-109), the flip-undo of
    ``inference`` (encoder_decoder.py:187-194) and the running mean of ``aug_test`` (:210-229).  ``depth_list[i]`` (B,1,h_i,w_i)
    = the sampler's output for augmentation i; ``flips[i]`` None | 'horizontal' | 'vertical'.  One augmentation = ``simple_test``.
    CUDA tensors only; no CPU path."""
    if not depth_list or len(depth_list) != len(flips) or len(depth_list) > _lib.MAX_AUGS:
        raise ValueError(f'depth_postprocess: 1..{_lib.MAX_AUGS} augmentations with one flip entry each')
    if not depth_list[0].is_cuda:
        raise _lib.DdpError('depth_postprocess: depth maps must be CUDA tensors (no CPU path)')
    dev, B = depth_list[0].device, depth_list[0].shape[0]
    keep = []
    augs = (_lib.DdpDepthAug * len(depth_list))()
    for i, (d, fl) in enumerate(zip(depth_list, flips)):
        d = d.contiguous().float()
        if d.dim() != 4 or d.shape[1] != 1 or d.shape[0] != B or d.device != dev:
            raise ValueError('depth_postprocess: every augmentation needs a (B,1,h,w) map on the same device')
        keep.append(d)
        augs[i].d_depth = d.data_ptr()
        augs[i].h, augs[i].w = d.shape[2], d.shape[3]
        augs[i].flip = {None: 0, False: 0, 'horizontal': 1, 'vertical': 2}[fl]
    oh, ow = int(out_size[0]), int(out_size[1])
    if out is None:
        out = torch.empty((B, 1, oh, ow), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.ddp_depth_postprocess(augs, len(keep), B, oh, ow, int(bool(align_corners)), C.c_float(min_depth),
                                             C.c_float(max_depth), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), lib)
    return out


def msda_forward_lds(value, samp, h, w, guess=None):
    """The deformable-attention core computed by the kernel the sampling loop runs (``ddp_msda_forward_lds``):
    value (R, h*w, 256), samp (R*h*w, 96) as for ``ddp_msda_forward``; guess (8,2) optional per-head window guess."""
    if not value.is_cuda:
        raise _lib.DdpError('msda_forward_lds: CUDA tensors only (no CPU path)')
    value, samp = value.contiguous().float(), samp.contiguous().float()
    rows = samp.shape[0]
    lib = _lib.load()
    nbytes = C.c_size_t(0)
    _lib.check(lib.ddp_msda_forward_lds_workspace(rows, h, w, C.byref(nbytes)), lib)
    ws = torch.empty(nbytes.value // 4 + 64, dtype=torch.float32, device=value.device)
    out = torch.empty((rows, 256), dtype=torch.float32, device=value.device)
    g = guess.contiguous().float() if guess is not None else None
    with torch.cuda.device(value.device):
        _lib.check(lib.ddp_msda_forward_lds(value.data_ptr(), samp.data_ptr(), g.data_ptr() if g is not None else None,
                                            out.data_ptr(), rows, h, w, ws.data_ptr(),
                                            torch.cuda.current_stream(value.device).cuda_stream), lib)
    return out
