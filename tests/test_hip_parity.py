"""Parity of the HIP path (through the C ABI of libddp_mi355x.so) against the CPU oracle and the
golden vectors generated from the reference.  Needs an MI355X: ``pytest -m gpu``.

Tolerances: the reference is fp32; the HIP path computes fp32-class results - by default on the bf16 matrix cores with
every fp32 operand split exactly into three bf16 pieces and the six significant cross products accumulated in fp32
(``gemm='bf16x3'``, csrc/gemm_bf16x3.h; tests/test_b3_arithmetic.py bounds it against fp64), or with exact products on
the f32-input MFMA (``gemm='f32'``); both in another summation order than the reference.  north_star bar: final logits /
depth within 1e-3 relative (max|a-b| / max|b|); measured here ~1e-5, asserted at 2e-4 to leave room for libm
differences, with argmax agreement reported for the classification outputs.  ``seg_trained_small`` (content-dependent
sampling offsets, synthetic.PROFILES['trained_like']): the NETWORK amplifies rounding - the reference itself sits 1.1e-3
from its own fp64 evaluation on that map with every argmax decision equal - so its bar is "fp32-class": within 4 x the
reference's distance to the fp64 oracle, computed in the test (tests/test_full_size_parity.py::test_c2_size_trained_like_weights
does the same at C2 size).
"""
import ctypes as C

import pytest
import torch

from golden_util import case_names, load_case, max_rel

pytestmark = pytest.mark.gpu

REL = 2e-4          # asserted;  north_star gate is 1e-3
GATE = 1e-3


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


def _engine(cfg, sd, dev, batch=1, **over):
    from ddp_amd.engine import DDPEngine
    task = cfg['task']
    kw = dict(h=cfg['h'], w=cfg['w'], batch=batch, randsteps=cfg['randsteps'], timesteps=cfg['timesteps'],
              bit_scale=cfg['bit_scale'], time_difference=cfg.get('time_difference', 1), device=dev)
    if task == 'seg':
        kw.update(num_classes=cfg['num_classes'], accumulation=cfg['accumulation'],
                  noise_schedule=cfg['noise_schedule'], sampler=cfg['diffusion'],
                  sample_range0=cfg.get('sample_range', (0.0, 0.999))[0])
    elif task == 'depth':
        kw.update(min_depth=cfg['min_depth'], max_depth=cfg['max_depth'], depth_scale_up=cfg.get('scale_up', False),
                  depth_use_eps=cfg.get('use_eps', True))
    else:
        kw.update(num_classes=6, feat_channels=cfg['feat_channels'], bev_input_scope=cfg['input_scope'],
                  bev_output_scope=cfg['output_scope'])
    kw.update(over)
    return DDPEngine(sd, task, **kw)


def test_library_is_loaded_native():
    from ddp_amd import _lib
    lib = _lib.load()
    assert lib.ddp_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize('m,n,k,gelu', [(128, 256, 256, 0), (300, 256, 256, 1), (442, 1024, 256, 1),
                                         (257, 256, 1024, 0), (1000, 96, 256, 0), (77, 152, 256, 0),
                                         (513, 20, 512, 0), (4096, 256, 256, 0)])
def test_linear(dev, m, n, k, gelu):
    """fp32 MFMA GEMM + bias (+GELU) vs torch fp64 -> fp32."""
    from ddp_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    ref = torch.nn.functional.linear(a.double(), w.double(), b.double())
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    da, dw, db = a.to(dev), w.to(dev), b.to(dev)
    out = torch.full((m, n), float('nan'), device=dev)
    _lib.check(lib.ddp_linear(da.data_ptr(), dw.data_ptr(), db.data_ptr(), out.data_ptr(), m, n, k, gelu,
                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert max_rel(out.cpu().double(), ref) < 2e-6


def test_linear_transpose_detect(dev):
    """A = I-like probe with asymmetric W: catches swapped MFMA operand / C-layout mistakes."""
    from ddp_amd import _lib
    lib = _lib.load()
    m = n = k = 256
    a = torch.eye(m, k)
    w = torch.arange(n * k, dtype=torch.float32).reshape(n, k) / (n * k)
    out = torch.empty(m, n, device=dev)
    da, dw = a.to(dev), w.to(dev)       # keep the device copies alive across the call
    _lib.check(lib.ddp_linear(da.data_ptr(), dw.data_ptr(), None, out.data_ptr(), m, n, k, 0,
                              torch.cuda.current_stream().cuda_stream))
    assert torch.equal(out.cpu(), w.t().contiguous())


@pytest.mark.parametrize('h,w,r', [(13, 17, 2), (32, 48, 1), (5, 70, 3)])
def test_msda_forward(dev, h, w, r):
    """bilinear gather + weighted sum vs the oracle's explicit-tap and grid_sample restatements,
    including samples far outside the map and exactly on integer positions."""
    from ddp_amd import _lib
    from oracle import ddp_oracle as O
    lib = _lib.load()
    n = h * w
    g = torch.Generator().manual_seed(h * 100 + w)
    value = torch.randn(r, n, 8, 32, generator=g)
    off = torch.randn(r, n, 8, 4, 2, generator=g) * 3.0
    off[:, ::7] = torch.round(off[:, ::7])          # integer offsets: exactly-on-pixel samples
    off[:, 3::11] *= 20.0                            # far outside
    aw = torch.randn(r, n, 8, 4, generator=g).softmax(-1)
    jj = torch.arange(w, dtype=torch.float32).repeat(h)
    ii = torch.arange(h, dtype=torch.float32).repeat_interleave(w)
    px = jj[None, :, None, None] + off[..., 0]
    py = ii[None, :, None, None] + off[..., 1]
    ref = O.msda_core_taps(value, h, w, px, py, aw)
    loc = torch.stack(((px + 0.5) / w, (py + 0.5) / h), -1)
    ref2 = O.msda_core_gridsample(value, h, w, loc, aw)
    assert max_rel(ref, ref2) < 1e-5
    samp = torch.cat([torch.stack((px, py), -1).reshape(r * n, 64), aw.reshape(r * n, 32)], 1).contiguous()
    out = torch.empty(r * n, 256, device=dev)
    dv, ds = value.reshape(r * n, 256).to(dev), samp.to(dev)
    _lib.check(lib.ddp_msda_forward(dv.data_ptr(), ds.data_ptr(), out.data_ptr(), r * n, h, w,
                                    torch.cuda.current_stream().cuda_stream))
    assert max_rel(out.cpu().reshape(r, n, 256), ref) < 1e-5


@pytest.mark.parametrize('h,w,r', [(13, 17, 2), (32, 48, 1), (5, 70, 3), (70, 9, 1), (8, 16, 1), (41, 67, 2)])
@pytest.mark.parametrize('table', ['wild', 'ring'])
def test_msda_forward_lds(dev, h, w, r, table):
    """The gather of the SAMPLING LOOP (k_msda_gather_lds: zero-padded map, head-major table, LDS window per (tile, head),
    mixed LDS / global path) behind the plain msda interface (ddp_msda_forward_lds), on adversarial sample tables:
      'wild'  N(0, 3 px) offsets, every 7th token exactly on pixel centres, every 11th 20x further out (far outside the
              map: clamped, zero contribution), so nearly every 8-token group leaves its window (mixed path);
      'ring'  the reference's initialisation - a per-head ring of radius 1..4 px (multi_scale_deform_attn.py:233-244) - plus
              0.3 px of content-dependent noise: windows hold, the staged path serves the taps; maps smaller than the
              8 x 16 tile in either direction, windows straddling every border.
    Against the oracle's explicit-tap restatement (vendored mmcv multi_scale_deform_attn.py:94-151); the window guess
    (none / per-head mean / deliberately wrong) must not change a single bit."""
    import math
    from ddp_amd.engine import msda_forward_lds
    from oracle import ddp_oracle as O
    n = h * w
    g = torch.Generator().manual_seed(h * 100 + w + (7 if table == 'ring' else 0))
    value = torch.randn(r, n, 8, 32, generator=g)
    if table == 'wild':
        off = torch.randn(r, n, 8, 4, 2, generator=g) * 3.0
        off[:, ::7] = torch.round(off[:, ::7])
        off[:, 3::11] *= 20.0
    else:
        th = torch.arange(8, dtype=torch.float32) * (2.0 * math.pi / 8)
        grid = torch.stack([th.cos(), th.sin()], -1)
        grid = grid / grid.abs().max(-1, keepdim=True).values                       # (8,2) unit ring, max-normalised
        ring = grid[:, None, :] * (torch.arange(4, dtype=torch.float32) + 1)[None, :, None]      # radius 1..4 per point
        off = ring[None, None] + 0.3 * torch.randn(r, n, 8, 4, 2, generator=g)
    aw = torch.randn(r, n, 8, 4, generator=g).softmax(-1)
    jj = torch.arange(w, dtype=torch.float32).repeat(h)
    ii = torch.arange(h, dtype=torch.float32).repeat_interleave(w)
    px = jj[None, :, None, None] + off[..., 0]
    py = ii[None, :, None, None] + off[..., 1]
    ref = O.msda_core_taps(value, h, w, px, py, aw)
    samp = torch.cat([torch.stack((px, py), -1).reshape(r * n, 64), aw.reshape(r * n, 32)], 1).contiguous()
    dv, ds = value.reshape(r, n, 256).to(dev), samp.to(dev)
    out = msda_forward_lds(dv, ds, h, w)
    assert max_rel(out.cpu().reshape(r, n, 256), ref) < 1e-5
    mean_guess = off.mean(dim=(0, 1, 3)).to(dev)                                    # (8,2): what the loop derives from bias + pos tables
    assert torch.equal(msda_forward_lds(dv, ds, h, w, guess=mean_guess), out)
    assert torch.equal(msda_forward_lds(dv, ds, h, w, guess=mean_guess + 5.0), out)  # wrong guess -> refill branch
    # the wave-per-token kernel of the unfused path computes the same values
    from ddp_amd import _lib
    lib = _lib.load()
    out2 = torch.empty(r * n, 256, device=dev)
    _lib.check(lib.ddp_msda_forward(dv.data_ptr(), ds.data_ptr(), out2.data_ptr(), r * n, h, w, torch.cuda.current_stream().cuda_stream))
    assert max_rel(out2.cpu(), out.cpu()) < 1e-5


def test_msda_forward_lds_nan_and_inf_coordinates(dev):
    """NaN / +-inf sample coordinates (they only arise from NaN inputs): the product gather does not fault, the affected
    (token, head) outputs stay finite (the point is clamped to the zero border), every other output is unchanged."""
    from ddp_amd.engine import msda_forward_lds
    h, w, r = 19, 37, 1
    n = h * w
    g = torch.Generator().manual_seed(5)
    value = torch.randn(r, n, 256, generator=g)
    off = torch.randn(r, n, 8, 4, 2, generator=g)
    aw = torch.randn(r, n, 8, 4, generator=g).softmax(-1)
    jj = torch.arange(w, dtype=torch.float32).repeat(h)
    ii = torch.arange(h, dtype=torch.float32).repeat_interleave(w)
    xy = torch.stack((jj[None, :, None, None] + off[..., 0], ii[None, :, None, None] + off[..., 1]), -1)
    clean = torch.cat([xy.reshape(n, 64), aw.reshape(n, 32)], 1).contiguous()
    bad_xy = xy.clone()
    bad_xy[0, 5::13, 2, 1, 0] = float('nan')
    bad_xy[0, 6::17, 4, 3, 1] = float('inf')
    bad_xy[0, 7::19, 6, 0, 0] = float('-inf')
    bad = torch.cat([bad_xy.reshape(n, 64), aw.reshape(n, 32)], 1).contiguous()
    a = msda_forward_lds(value.to(dev), clean.to(dev), h, w).cpu().reshape(n, 8, 32)
    b = msda_forward_lds(value.to(dev), bad.to(dev), h, w).cpu().reshape(n, 8, 32)
    assert torch.isfinite(b).all()
    touched = torch.zeros(n, 8, dtype=torch.bool)
    touched[5::13, 2] = True
    touched[6::17, 4] = True
    touched[7::19, 6] = True
    assert torch.equal(a[~touched], b[~touched])


def test_time_embed(dev):
    """LearnedSinusoidalPosEmb + time_mlp + per-layer FiLM on device vs oracle, at the ill-conditioned
    log-SNR values of the real schedule."""
    from ddp_amd import _lib, schedule
    from ddp_amd.engine import PackedWeights
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    lib = _lib.load()
    sd = synthetic.make_state_dict('seg', 19, 6, 256, seed=5)
    pw = PackedWeights(sd, 'seg', 6, dev)
    recs = schedule.step_records('seg', 10)
    tin = [r['time_in'] for r in recs]
    S = len(tin)
    temb = torch.empty(S, 1024, device=dev)
    film = torch.empty(S, 6, 512, device=dev)
    scratch = torch.empty(64 + S * (64 + 1024) + 4096, device=dev)
    arr = (C.c_float * S)(*tin)
    _lib.check(lib.ddp_time_embed(C.byref(pw.struct), 6, arr, S, temb.data_ptr(), film.data_ptr(),
                                  scratch.data_ptr(), torch.cuda.current_stream().cuda_stream))
    t = torch.tensor(tin, dtype=torch.float32)
    ref = O.time_mlp(t, sd)
    assert max_rel(temb.cpu(), ref) < 2e-5
    for l in range(6):
        sc, sh = O.film_vectors(ref, sd, l)
        assert max_rel(film[:, l].cpu(), torch.cat([sc, sh], 1)) < 2e-5


@pytest.mark.parametrize('name', ['seg_city_r2'])
def test_head_forward_trace(dev, name):
    """decode_head plugin surface: DeformableHeadWithTime.forward(feat, temb) vs the reference's
    recorded logits for the same feat."""
    from oracle import ddp_oracle as O
    cfg, sd, x, noise, _, g = load_case(name)
    eng = _engine(cfg, sd, dev)
    feat = g['feat_step0'].to(dev).contiguous()
    ls = O.alpha_cosine_log_snr(torch.tensor([1.0]))
    temb = O.time_mlp(ls, sd).to(dev)
    out = eng.head_forward(feat, temb)
    assert max_rel(out.cpu(), g['logits_steps'][0]) < REL


# every engine / fusion level of the library runs the reference-made fixtures: the default path (bf16x3 split products,
# persistent layer kernel, fused step head and seg tail), the same arithmetic as separate tile GEMMs, and the exact
# f32-input MFMA engine (include/ddp_mi355x.h DDP_GEMM_*, DDP_FLAG_*)
VARIANTS = {'bf16x3': dict(gemm='bf16x3'), 'f32': dict(gemm='f32'),
            'bf16x3-unfused-layer': dict(gemm='bf16x3', fused_layer=False),
            'bf16x3-unfused-prologue': dict(gemm='bf16x3', fused_prologue=False),
            # the last layer of a step and the seg tail as the two kernels they were fused from (DDP_FLAG_UNFUSED_TAIL)
            'bf16x3-unfused-tail': dict(gemm='bf16x3', fused_tail=False),
            # the first step's head as NCHW -> SB conversions + x-projection GEMM + k_layer MODE 2 (DDP_FLAG_SB_HEAD)
            'bf16x3-sb-head': dict(gemm='bf16x3', nchw_head=False),
            # the LDS gather's "actual mean offset is far from the guess: refill the window" branch (DDP_FLAG_GATHER_GUESS_ZERO)
            'bf16x3-gather-refill': dict(gemm='bf16x3', gather_guess_zero=True)}


_FP64_DIST = {}


def _fp64_distance(name, cfg, sd, x, noise, ref):
    """max-rel of the fixture's (fp32, reference-made) output against the oracle evaluated in fp64 - once per fixture"""
    if name not in _FP64_DIST:
        from oracle import ddp_oracle as O
        r64 = O.ddim_sample_seg(x.double(), noise.double(), {k: v.double() for k, v in sd.items()}, timesteps=cfg['timesteps'],
                                randsteps=cfg['randsteps'], bit_scale=cfg['bit_scale'], accumulation=cfg['accumulation'])
        _FP64_DIST[name] = max_rel(ref, r64.float())
    return _FP64_DIST[name]


@pytest.mark.parametrize('variant', sorted(VARIANTS))
@pytest.mark.parametrize('name', case_names())
def test_sample_golden(dev, name, variant):
    """the whole K-step loop vs the golden output recorded from the reference."""
    cfg, sd, x, noise, step_noise, g = load_case(name)
    eng = _engine(cfg, sd, dev, **VARIANTS[variant])
    sn = step_noise.unsqueeze(1).contiguous().to(dev) if step_noise is not None else None
    dx, dn = x.to(dev), noise.unsqueeze(0).contiguous().to(dev)
    out = eng.sample(dx, dn, sn)
    torch.cuda.synchronize()
    ref = g['out']
    assert out.shape == ref.shape
    err = max_rel(out.cpu(), ref)
    bar = REL
    if cfg.get('profile', 'init') == 'trained_like':
        # content-dependent sampling offsets: rounding is amplified by the network.  Yardstick = how far the REFERENCE (fp32) is
        # from the fp64 oracle on this fixture (the oracle is bit-identical to the reference in fp32: tests/test_oracle_golden.py)
        bar = 4 * _fp64_distance(name, cfg, sd, x, noise, ref) + REL
    if cfg['task'] != 'depth':
        agree = (out.cpu().argmax(1) == ref.argmax(1)).float().mean().item()
        print(f'{name}: max-rel {err:.3e} (bar {bar:.1e}), argmax agreement {agree:.4f}')
        assert agree > (0.99 if bar > REL else 0.999)
    else:
        print(f'{name}: max-rel {err:.3e}')
    assert err < bar and REL < GATE


@pytest.mark.parametrize('name', ['seg_ade_k3', 'seg_city_k10', 'seg_td2'])
def test_fused_and_unfused_tail_same_bits(dev, name):
    """k_layer MODE 6 (last layer of a step + seg tail + next head in ONE kernel) against the two kernels it was fused from
    (DDP_FLAG_UNFUSED_TAIL: MODE 0 + MODE 4 / MODE 1): the same contractions in the same order, the layer output merely stays in
    registers - the outputs must be bit-identical (150 and 19 classes, accumulation on and off, r = 1 and 2)."""
    cfg, sd, x, noise, step_noise, g = load_case(name)
    dx, dn = x.to(dev), noise.unsqueeze(0).contiguous().to(dev)
    a = _engine(cfg, sd, dev, fused_tail=True).sample(dx, dn).clone()
    b = _engine(cfg, sd, dev, fused_tail=False).sample(dx, dn)
    assert torch.equal(a, b)


@pytest.mark.parametrize('r', [1, 2])
@pytest.mark.parametrize('ncls', [19, 64, 65, 100, 128, 150, 192, 193, 256])
def test_fused_and_unfused_tail_same_bits_class_counts(dev, ncls, r):
    """VERDICT r05 "next" #2: k_layer MODE 6 exists for every class count 1..256 (1, 2, 3, 4 chunks of 64 classes) and for r > 1
    (res_rn: r noisy maps share one x projection) - no segmentation configuration falls back to the three-kernel tail silently.
    Fused == unfused bit for bit on seeded inputs (segmentors/ddp.py:219-245), odd map size, K = 3, accumulation."""
    from ddp_amd.utils import synthetic
    sd = synthetic.make_state_dict('seg', ncls, 6, 256, seed=300 + ncls)
    h, w = 11, 19
    x, noise = synthetic.make_inputs(2, h, w, r, 256, 256, seed=ncls + r)
    cfg = dict(task='seg', h=h, w=w, randsteps=r, timesteps=3, bit_scale=0.01, num_classes=ncls, accumulation=True,
               noise_schedule='cosine', diffusion='ddim')
    dx, dn = x.to(dev), noise.to(dev)
    a = _engine(cfg, sd, dev, batch=2, fused_tail=True).sample(dx, dn).clone()
    b = _engine(cfg, sd, dev, batch=2, fused_tail=False).sample(dx, dn)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert torch.allclose(a.sum(1), torch.ones_like(a.sum(1)), atol=1e-5)


@pytest.mark.parametrize('name', case_names('depth') + case_names('bev'))
def test_fused_and_unfused_step_boundary_depth_bev(dev, name):
    """Round 6: the depth and BEV step boundaries fused the way MODE 6 / 7 fused the segmentation sampler's - the last layer of a
    step carries the head convolution as its tail (k_layer MODE 9: the nine taps of conv_depth; MODE 8: conv_seg + sigmoid +
    accumulation + the thresholded x0 code), and the BEV sampler runs the u chain (loop-invariant resample(W_x x + b), u = W_m m
    updated through the 2^K-row table).  Against DDP_FLAG_UNFUSED_TAIL = the round-5 launches (head GEMM on the SB layer output,
    k_depth_update / k_bev_update on the 256-channel map).  Not bit-identical: the head contraction runs in the layer kernel's
    K order instead of the tile GEMM's, and resample(a + b) != resample(a) + resample(b) in fp32 - asserted at rounding level,
    both against each other and (test_sample_golden) against the reference fixture.
    depth/depth/models/depther/ddp.py:229-247; bev/mmdet3d/models/fusion_models/ddp.py:268-301."""
    cfg, sd, x, noise, _, g = load_case(name)
    dx, dn = x.to(dev), noise.unsqueeze(0).contiguous().to(dev)
    a = _engine(cfg, sd, dev, fused_tail=True).sample(dx, dn).clone().cpu()
    b = _engine(cfg, sd, dev, fused_tail=False).sample(dx, dn).cpu()
    err = max_rel(a, b)
    print(f'{name}: fused vs unfused step boundary max-rel {err:.3e}; vs reference fused {max_rel(a, g["out"]):.3e} unfused {max_rel(b, g["out"]):.3e}')
    assert err < 5e-5
    if cfg['task'] == 'bev':
        assert float(((a > 0.5) == (b > 0.5)).float().mean()) > 0.9995


@pytest.mark.parametrize('task,h,w,r', [('depth', 5, 37, 1), ('depth', 9, 11, 2), ('bev', 12, 20, 1), ('bev', 16, 16, 2)])
def test_depth_bev_batch_equals_independent_runs(dev, task, h, w, r):
    """The round-6 step heads (k_depth_head, k_bev_q, k_bev_u_update, the tails of k_layer MODE 8 / 9, layer 0 as MODE 10) index
    tokens of ALL maps of a call; the reference-made fixtures are single-image.  Three images of an odd size in ONE call (token counts
    that are no multiple of 32: groups straddle images) must give, image by image, the bits of three single-image calls."""
    from ddp_amd.engine import DDPEngine
    from ddp_amd.utils import synthetic
    B = 3
    if task == 'depth':
        sd = synthetic.make_state_dict('depth', 1, 6, 256, seed=77)
        x, noise = synthetic.make_inputs(B, h, w, r, 256, 1, seed=78)
        kw = dict(h=h, w=w, randsteps=r, timesteps=4, bit_scale=0.1, min_depth=1e-3, max_depth=80.0, device=dev)
    else:
        sd = synthetic.make_state_dict('bev', 6, 5, 256, seed=79)
        x, noise = synthetic.make_inputs(B, h, w, r, 256, 256, seed=80)
        kw = dict(h=h, w=w, randsteps=r, timesteps=3, bit_scale=0.01, num_classes=6, feat_channels=256, device=dev,
                  bev_input_scope=[[-51.2, 51.2, 102.4 / h], [-51.2, 51.2, 102.4 / w]],
                  bev_output_scope=[[-50, 50, 100.0 / (h + 7)], [-50, 50, 100.0 / (w + 9)]])
    eb = DDPEngine(sd, task, batch=B, **kw)
    out = eb.sample(x.to(dev), noise.to(dev)).clone()
    assert torch.isfinite(out).all()
    e1 = DDPEngine(sd, task, batch=1, **kw)
    for b in range(B):
        one = e1.sample(x[b:b + 1].contiguous().to(dev), noise[b:b + 1].contiguous().to(dev))
        assert torch.equal(one[0], out[b]), f'image {b} differs between the batched and the single-image call'


def test_sample_batch_matches_per_image_oracle(dev):
    """B=3 images in ONE call (what the reference cannot do: its loop is b=1) == three independent
    oracle runs, each with its own noise."""
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    sd = synthetic.make_state_dict('seg', 150, 6, 256, seed=42)
    B, r, h, w = 3, 2, 12, 20
    x, noise = synthetic.make_inputs(B, h, w, r, 256, 256, seed=9)
    cfg = dict(task='seg', h=h, w=w, randsteps=r, timesteps=3, bit_scale=0.01, num_classes=150, accumulation=True,
               noise_schedule='cosine', diffusion='ddim')
    eng = _engine(cfg, sd, dev, batch=B)
    dx, dn = x.to(dev), noise.to(dev)
    out = eng.sample(dx, dn).cpu()
    for b in range(B):
        ref = O.ddim_sample_seg(x[b:b + 1], noise[b], sd, timesteps=3, randsteps=r, bit_scale=0.01, accumulation=True)
        assert max_rel(out[b:b + 1], ref) < REL
    # accumulated softmax means are probability vectors
    assert torch.allclose(out.sum(1), torch.ones(B, h, w), atol=1e-5)


# ---- post-loop epilogue -----------------------------------------------------------------------
from golden_util import load_post_case  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize('name', case_names('post'))
def test_post_epilogue_golden(name):
    """fused resize/crop/resize/softmax/flip/argmax kernel vs the reference's class map: identical wherever the
    reference's top-2 probability margin is above fp32 interpolation noise (the CPU kernel may contract to FMA)."""
    from ddp_amd.engine import seg_postprocess
    cfg, scores, seg, margin = load_post_case(name)
    got = seg_postprocess(scores.cuda(), cfg['img'], cfg['img_shape'], cfg['ori_shape'], cfg['align_corners'], cfg['flip'])
    torch.cuda.synchronize()
    got = got[0].cpu()
    diff = got != seg
    assert diff.float().mean() < 1e-3
    assert not (diff & (margin > 1e-5)).any()


@pytest.mark.gpu
@pytest.mark.parametrize('name', case_names('aug'))
def test_aug_epilogue_golden(name):
    """fused multi-scale / flip epilogue (ddp_seg_aug_postprocess) vs the reference's own aug_test: mean probabilities to
    rounding, class map identical wherever the reference's top-2 margin is above interpolation / exp rounding noise."""
    from golden_util import load_aug_case
    from ddp_amd.engine import seg_aug_postprocess
    cfg, scores, metas, seg, prob, margin = load_aug_case(name)
    got, p = seg_aug_postprocess([t.cuda() for t in scores], metas, cfg['ori_shape'], cfg['align_corners'], return_prob=True)
    torch.cuda.synchronize()
    assert max_rel(p[0].cpu(), prob) < 2e-6
    diff = got[0].cpu() != seg
    assert diff.float().mean() < 1e-3
    assert not (diff & (margin > 1e-5)).any()
    # without the probability output: the same class map
    assert torch.equal(seg_aug_postprocess([t.cuda() for t in scores], metas, cfg['ori_shape'], cfg['align_corners']), got)


@pytest.mark.gpu
def test_aug_epilogue_full_size_and_errors():
    """ADE-size multi-scale + flip (6 augmentations of a 512x683 image, 150 classes): equals the oracle's aug_test on the
    class map up to near-ties; one augmentation without flip / rescale == the plain epilogue kernel; bad arguments raise."""
    from ddp_amd.engine import seg_aug_postprocess, seg_postprocess
    from ddp_amd import _lib
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    ori = (512, 683)
    augs = []
    for i, s in enumerate((0.5, 1.0, 1.5)):
        H, W = int(ori[0] * s + 0.5), int(ori[1] * s + 0.5)
        Hp, Wp = (H + 31) // 32 * 32, (W + 31) // 32 * 32
        for flip in (None, 'horizontal'):
            augs.append((synthetic.make_scores(1, 150, Hp // 4, Wp // 4, 70 + i), dict(img_size=(Hp, Wp), crop_size=(H, W), flip=flip)))
    scores, metas = [a[0] for a in augs], [a[1] for a in augs]
    seg, p = seg_aug_postprocess([t.cuda() for t in scores], metas, ori, False, return_prob=True)
    ref, rp = O.seg_aug_test(scores, metas, ori, False)
    assert max_rel(p.cpu(), rp) < 2e-6
    top2 = rp.topk(2, dim=1).values
    diff = seg.cpu().long() != ref
    assert diff.float().mean() < 1e-3 and not (diff & ((top2[:, 0] - top2[:, 1]) > 1e-5)).any()
    assert torch.allclose(p.sum(1), torch.ones_like(p[:, 0]), atol=1e-5)
    one = seg_aug_postprocess([scores[2].cuda()], [dict(img_size=metas[2]['img_size'], crop_size=metas[2]['crop_size'], flip=None)], ori)
    plain = seg_postprocess(scores[2].cuda(), metas[2]['img_size'], metas[2]['crop_size'], ori)
    assert (one != plain).float().mean() < 1e-4          # softmax is monotone: only exp-rounding ties may differ
    with pytest.raises(_lib.DdpError):
        seg_aug_postprocess([torch.zeros(1, 19, 4, 4)], [dict(img_size=(16, 16))], (16, 16))          # CPU tensor
    with pytest.raises(_lib.DdpError):
        seg_aug_postprocess([torch.zeros(1, 19, 4, 4).cuda()], [dict(img_size=(16, 16), crop_size=(32, 16))], (16, 16))
    with pytest.raises(ValueError):
        seg_aug_postprocess([torch.zeros(1, 19, 4, 4).cuda()] * 17, [dict(img_size=(16, 16))] * 17, (16, 16))


@pytest.mark.gpu
@pytest.mark.parametrize('name', case_names('slide'))
def test_slide_epilogue_golden(name):
    """fused sliding-window epilogue (ddp_seg_slide_postprocess) vs the reference's own slide_inference / inference / simple_test:
    probabilities to rounding, class map identical wherever the reference's top-2 margin is above interpolation / exp noise;
    the three outputs (class map, probabilities, averaged scores) are consistent with each other."""
    from golden_util import load_slide_case, reference_window_grid
    from ddp_amd.engine import seg_slide_postprocess, slide_windows
    cfg, scores, seg, prob, margin = load_slide_case(name)
    ys, xs, crop = reference_window_grid(cfg)                           # the grid the REFERENCE cut, recorded in the fixture
    assert slide_windows(cfg['img'], cfg['crop_size'], cfg['stride']) == (ys, xs, crop)      # ... and the product's is the same
    sc = torch.stack(scores).cuda()                                     # (windows, 1, K, h, w)
    args = (sc, ys, xs, crop, cfg['img'], cfg['img_shape'], cfg['ori_shape'], cfg['align_corners'])
    got = seg_slide_postprocess(*args, flip=cfg['flip'], want='seg')[0].cpu()
    p = seg_slide_postprocess(*args, flip=cfg['flip'], want='prob')[0].cpu()
    raw = seg_slide_postprocess(*args, flip=None, want='scores')[0].cpu()
    assert max_rel(p, prob) < 2e-6
    diff = got != seg
    assert diff.float().mean() < 1e-3 and not (diff & (margin > 1e-5)).any()
    un = torch.softmax(raw, dim=0)
    if cfg['flip']:
        un = un.flip(dims=(2,) if cfg['flip'] == 'horizontal' else (1,))
    assert max_rel(un, prob) < 2e-6


@pytest.mark.gpu
def test_slide_epilogue_cityscapes_size_and_errors():
    """Cityscapes protocol of mmseg (1024 x 2048, crop 512 x 1024, stride 341 x 683 -> 3 x 3 windows, 19 classes, b = 2):
    equals the oracle's slide_inference; one window covering the whole image == the plain epilogue; bad grids are refused."""
    from ddp_amd import _lib
    from ddp_amd.engine import seg_postprocess, seg_slide_postprocess, slide_windows
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    img, crop_size, stride = (1024, 2048), (512, 1024), (341, 683)
    ys, xs, crop = slide_windows(img, crop_size, stride)
    assert (len(ys), len(xs), crop) == (3, 3, (512, 1024)) and ys[-1] == 512 and xs[-1] == 1024
    scores = [synthetic.make_scores(2, 19, 128, 256, 900 + i) for i in range(9)]
    sc = torch.stack(scores).cuda()
    seg = seg_slide_postprocess(sc, ys, xs, crop, img).cpu()
    raw = seg_slide_postprocess(sc, ys, xs, crop, img, want='scores').cpu()
    ref = O.seg_slide_inference(scores, ys, xs, crop, img)
    assert max_rel(raw, ref) < 2e-6
    top2 = ref.topk(2, dim=1).values
    diff = seg.long() != ref.argmax(1)
    assert diff.float().mean() < 1e-3 and not (diff & ((top2[:, 0] - top2[:, 1]) > 1e-4)).any()
    one = synthetic.make_scores(1, 19, 16, 24, 5).cuda()
    a = seg_slide_postprocess(one[None], [0], [0], (64, 96), (64, 96), (61, 90), (70, 101), flip='horizontal')
    b = seg_postprocess(one, (64, 96), (61, 90), (70, 101), flip='horizontal')
    assert (a != b).float().mean() < 1e-3
    with pytest.raises(_lib.DdpError):
        seg_slide_postprocess(torch.zeros(1, 1, 19, 4, 4), [0], [0], (16, 16), (16, 16))                       # CPU tensor
    with pytest.raises(_lib.DdpError, match='gap|cover'):
        seg_slide_postprocess(torch.zeros(2, 1, 19, 4, 4).cuda(), [0], [0, 20], (16, 16), (16, 40))           # columns 16..19 uncovered
    with pytest.raises(_lib.DdpError, match='more than 4'):
        seg_slide_postprocess(torch.zeros(7, 1, 19, 4, 4).cuda(), [0], [0, 2, 4, 6, 8, 10, 12], (16, 16), (16, 28))   # stride < crop / 4
    # four-fold overlap: the last window clamped back into the image - crop 9, stride 3, H = 16 -> origins 0, 3, 6, 7,
    # pixels 7 and 8 lie under all four - is a grid the reference accepts; against the oracle's slide_inference
    ys4, xs4, crop4 = slide_windows((16, 20), (9, 9), (3, 5))
    assert ys4 == [0, 3, 6, 7] and crop4 == (9, 9)
    sc4 = [synthetic.make_scores(1, 19, 4, 4, 700 + i) for i in range(len(ys4) * len(xs4))]      # (16-byte aligned window maps)
    raw4 = seg_slide_postprocess(torch.stack(sc4).cuda(), ys4, xs4, crop4, (16, 20), want='scores').cpu()
    assert max_rel(raw4, O.seg_slide_inference(sc4, ys4, xs4, crop4, (16, 20))) < 2e-6
    with pytest.raises(ValueError):
        seg_slide_postprocess(torch.zeros(3, 1, 19, 4, 4).cuda(), [0], [0, 8], (16, 16), (16, 24))            # 3 score maps for 2 windows


@pytest.mark.gpu
@pytest.mark.parametrize('name', case_names('dpost'))
def test_depth_epilogue_golden(name):
    """fused depth epilogue (ddp_depth_postprocess) vs the reference's own ``model(return_loss=False, **data)`` output
    (depth/depth/apis/test.py:88 -> base.py forward_test -> simple_test / aug_test): clamp, bilinear resize, flip-undo, mean."""
    from golden_util import load_dpost_case
    from ddp_amd.engine import depth_postprocess
    cfg, maps, flips, out = load_dpost_case(name)
    size = cfg['img'] if cfg['rescale'] else (cfg['augs'][0]['h'], cfg['augs'][0]['w'])
    got = depth_postprocess([m.cuda() for m in maps], flips, size, cfg['min_depth'], cfg['max_depth'], cfg['align_corners'])
    torch.cuda.synchronize()
    assert got.shape == out.shape
    err = float((got.cpu() - out).abs().max())
    print(f'DEPTH EPILOGUE {name}: max abs diff {err:.3e} on a depth scale of {cfg["max_depth"]:g}')
    assert err <= 2e-5 * cfg['max_depth']                     # the CPU interpolation kernel may contract to FMA
    assert float(got.min()) >= cfg['min_depth'] and float(got.max()) <= cfg['max_depth']


@pytest.mark.gpu
def test_depth_epilogue_full_size_properties_and_errors():
    """C4-size batch (16 x 88 x 304 -> 16 x 352 x 1216), plain + flipped augmentation: equals the oracle; flipping an
    augmentation's map and its flag together changes nothing; batch entries independent; NaN survives the clamp as in
    torch.clamp; rows that are not a multiple of 4 wide take the scalar store path; bad arguments raise."""
    from ddp_amd import _lib
    from ddp_amd.engine import depth_postprocess
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    a, b = synthetic.make_depth_map(16, 88, 304, 50), synthetic.make_depth_map(16, 88, 304, 51)
    got = depth_postprocess([a.cuda(), b.cuda()], [None, 'horizontal'], (352, 1216), 1e-3, 80.0)
    ref = O.depth_postprocess([a, b], [None, 'horizontal'], (352, 1216), 1e-3, 80.0)
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * 80
    same = depth_postprocess([a.cuda(), b.flip(dims=(3,)).cuda()], [None, None], (352, 1216), 1e-3, 80.0)
    assert float((same - got).abs().max()) <= 2e-5 * 80       # resize commutes with the flip up to rounding
    one = depth_postprocess([a[5:6].cuda(), b[5:6].cuda()], [None, 'horizontal'], (352, 1216), 1e-3, 80.0)
    assert torch.equal(one[0], got[5])
    odd = depth_postprocess([a[:1].cuda()], ['vertical'], (351, 1213), 1e-3, 80.0)           # 1213 % 4 != 0
    assert float((odd.cpu() - O.depth_postprocess([a[:1]], ['vertical'], (351, 1213), 1e-3, 80.0)).abs().max()) <= 2e-5 * 80
    n = a[:1].clone()
    n[0, 0, 3, 7] = float('nan')
    gn = depth_postprocess([n.cuda()], [None], (88, 304), 1e-3, 80.0).cpu()
    assert torch.isnan(gn[0, 0, 3, 7]) and int(torch.isnan(gn).sum()) == 1
    with pytest.raises(_lib.DdpError):
        depth_postprocess([torch.zeros(1, 1, 4, 4)], [None], (16, 16), 1e-3, 80.0)             # CPU tensor: no CPU path
    with pytest.raises(_lib.DdpError):
        depth_postprocess([torch.zeros(1, 1, 4, 4).cuda()], [None], (16, 16), 80.0, 1e-3)      # empty depth range
    with pytest.raises(ValueError):
        depth_postprocess([torch.zeros(1, 2, 4, 4).cuda()], [None], (16, 16), 1e-3, 80.0)      # not a (B,1,h,w) map
    with pytest.raises(ValueError):
        depth_postprocess([torch.zeros(1, 1, 4, 4).cuda()] * 17, [None] * 17, (16, 16), 1e-3, 80.0)


@pytest.mark.gpu
def test_post_epilogue_full_size_properties():
    """C2-size scores (8,150,128,256) -> (8,512,1024): equals the oracle on a sampled image, flip == flipped map,
    batch entries independent."""
    from ddp_amd.engine import seg_postprocess
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    s = synthetic.make_scores(8, 150, 128, 256, seed=77).cuda()
    a = seg_postprocess(s, (512, 1024))
    f = seg_postprocess(s, (512, 1024), flip='horizontal')
    one = seg_postprocess(s[3:4], (512, 1024))
    torch.cuda.synchronize()
    assert torch.equal(f, a.flip(dims=(2,)))
    assert torch.equal(one[0], a[3])
    ref = O.seg_postprocess(s[3:4].cpu(), (512, 1024))[0]
    assert (a[3].cpu().long() != ref).float().mean() < 1e-4


@pytest.mark.gpu
def test_post_epilogue_errors():
    from ddp_amd import _lib
    from ddp_amd.engine import seg_postprocess
    with pytest.raises(_lib.DdpError):
        seg_postprocess(torch.zeros(1, 19, 4, 4), (16, 16))                       # CPU tensor: no CPU path
    with pytest.raises(_lib.DdpError):
        seg_postprocess(torch.zeros(1, 19, 4, 4).cuda(), (16, 16), (32, 16), (16, 16))   # crop larger than the image


# ---- MultiStageMerging neck ----------------------------------------------------------------------
from golden_util import load_neck_case  # noqa: E402


def _neck(sd, align_corners):
    import ddp_amd
    neck = ddp_amd.MultiStageMerging([256] * 4, 256, kernel_size=1, norm_cfg=dict(type='GN', num_groups=32), act_cfg=None,
                                     align_corners=align_corners)
    neck.load_state_dict(sd, strict=True)
    return neck.cuda().eval()


@pytest.mark.gpu
@pytest.mark.parametrize('name', case_names('neck'))
def test_neck_golden(name):
    cfg, levels, sd, out = load_neck_case(name)
    neck = _neck(sd, cfg['align_corners'])
    got = neck([t.cuda() for t in levels])[0]
    torch.cuda.synchronize()
    assert got.shape == out.shape
    err = max_rel(got.cpu(), out)
    print(f'{name}: max-rel {err:.3e}')
    assert err < REL


@pytest.mark.gpu
def test_neck_full_size_properties():
    """C2-size levels (8 images, stride-4 grid 128x256): per-group statistics of the GroupNorm output, batch
    independence, bit-reproducibility, and the oracle on one image."""
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    sd = synthetic.make_neck_state_dict(5)
    levels = [t.cuda() for t in synthetic.make_levels(8, 128, 256, 5)]
    neck = _neck(sd, False)
    a = neck(levels)[0]
    b = neck(levels)[0]
    one = neck([t[2:3] for t in levels])[0]
    torch.cuda.synchronize()
    assert torch.equal(a, b)                                    # deterministic two-stage GroupNorm statistics
    assert torch.equal(one[0], a[2])                            # images are independent
    y = (a - sd['down.gn.bias'].cuda().view(1, 256, 1, 1)) / sd['down.gn.weight'].cuda().view(1, 256, 1, 1)
    g = y.view(8, 32, 8 * 128 * 256)
    assert g.mean(-1).abs().max() < 1e-4 and (g.var(-1, unbiased=False) - 1).abs().max() < 1e-3
    ref = O.neck_multi_stage_merging([t[2:3].cpu() for t in levels], sd)
    assert max_rel(a[2:3].cpu(), ref) < REL


# ---- FCNHeadWithTime ----------------------------------------------------------------------------
from golden_util import load_fcn_case  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize('name', case_names('fcn'))
def test_fcn_head_golden(name):
    import ddp_amd
    cfg, feat, temb, sd, out = load_fcn_case(name)
    head = ddp_amd.FCNHeadWithTime(num_convs=cfg['num_convs'], kernel_size=3, concat_input=cfg['concat_input'],
                                   dilation=cfg['dilation'], in_channels=256, channels=256, num_classes=cfg['num_classes'],
                                   in_index=0, norm_cfg=dict(type='BN') if cfg['with_norm'] else None)
    head.load_state_dict(sd, strict=True)
    head = head.cuda().eval()
    times = temb.expand(cfg['maps'], 1024).cuda() if temb is not None else None
    got = head([feat.cuda()], times)
    torch.cuda.synchronize()
    assert got.shape == out.shape
    err = max_rel(got.cpu(), out)
    print(f'{name}: max-rel {err:.3e}')
    assert err < REL
    with pytest.raises(Exception):
        head([feat], None)                       # CPU tensors: no CPU path


# ---- FPN neck -----------------------------------------------------------------------------------
from golden_util import load_fpn_case  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize('name', case_names('fpn'))
def test_fpn_golden(name):
    import ddp_amd
    cfg, levels, sd, outs = load_fpn_case(name)
    neck = ddp_amd.FPN(in_channels=cfg['in_channels'], out_channels=256, act_cfg=None, norm_cfg=dict(type='GN', num_groups=32),
                       num_outs=4)
    neck.load_state_dict(sd, strict=True)
    neck = neck.cuda().eval()
    got = neck([t.cuda() for t in levels])
    torch.cuda.synchronize()
    for l, (g, o) in enumerate(zip(got, outs)):
        assert g.shape == o.shape
        err = max_rel(g.cpu(), o)
        print(f'{name} level {l}: max-rel {err:.3e}')
        assert err < REL


@pytest.mark.gpu
def test_fpn_then_merging_chain_matches_oracle():
    """the two necks chained as in the DDP configs (configs/ade/ddp_swin_t...:39-54) == the oracle chain"""
    import ddp_amd
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    inc = [96, 192, 384, 768]
    sdf, sdm = synthetic.make_fpn_state_dict(inc, 3), synthetic.make_neck_state_dict(3)
    fpn = ddp_amd.FPN(in_channels=inc, out_channels=256, act_cfg=None, norm_cfg=dict(type='GN', num_groups=32), num_outs=4)
    msm = ddp_amd.MultiStageMerging([256] * 4, 256, kernel_size=1, norm_cfg=dict(type='GN', num_groups=32), act_cfg=None)
    fpn.load_state_dict(sdf)
    msm.load_state_dict(sdm)
    chain = torch.nn.Sequential(fpn, msm).cuda().eval()
    levels = synthetic.make_backbone_levels(2, inc, 32, 48, 3)
    got = chain([t.cuda() for t in levels])[0]
    torch.cuda.synchronize()
    ref = O.neck_multi_stage_merging(list(O.neck_fpn(levels, sdf)), sdm)
    assert max_rel(got.cpu(), ref) < REL
    # the container the segmentor builds from the config's neck list runs the pair as ONE C entry (ddp_neck_fpn_msm: the FPN
    # outputs never take their NCHW form): same state_dict keys as nn.Sequential, same result as the member-by-member chain
    fused = ddp_amd.NeckChain(fpn, msm).cuda().eval()
    assert fused.fused() and list(fused.state_dict()) == list(chain.state_dict())
    a = fused([t.cuda() for t in levels])[0]
    assert max_rel(a.cpu(), ref) < REL and max_rel(a.cpu(), got.cpu()) < 1e-5
    b = fused([t.cuda() for t in levels])[0]                   # second call: weight region of the workspace re-used
    assert torch.equal(a, b) and fused._ws_weights is not None
    with torch.no_grad():                                      # changed weights are re-packed
        msm.down.gn.bias.add_(0.5)
    c = fused([t.cuda() for t in levels])[0]
    assert max_rel(c.cpu(), O.neck_multi_stage_merging(list(O.neck_fpn(levels, sdf)), {k: v.cpu() for k, v in msm.state_dict().items()})) < REL
    # maps whose token count is not a multiple of 32 (GroupNorm statistics by the separate kernels), odd sizes
    lv2 = synthetic.make_backbone_levels(1, inc, 13, 19, 4)
    g2 = fused([t.cuda() for t in lv2])[0]
    r2 = O.neck_multi_stage_merging(list(O.neck_fpn(lv2, sdf)), {k: v.cpu() for k, v in msm.state_dict().items()})
    assert max_rel(g2.cpu(), r2) < REL


# ---- edge geometry: single-token maps, one layer (no "next layer" projections), the class-count limits ---------------
@pytest.mark.gpu
@pytest.mark.parametrize('gemm', ['bf16x3', 'f32'])
@pytest.mark.parametrize('h,w,L,K,r', [(1, 1, 1, 2, 1), (2, 3, 2, 256, 2), (5, 4, 1, 19, 1), (3, 50, 3, 150, 1)])
def test_sample_edge_geometry_vs_oracle(dev, h, w, L, K, r, gemm):
    """maps smaller than one 32-token group / one 128-token tile, a single decoder layer, 2 and 256 classes"""
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    sd = synthetic.make_state_dict('seg', K, L, 256, seed=11)
    x, noise = synthetic.make_inputs(2, h, w, r, 256, 256, seed=12)
    cfg = dict(task='seg', h=h, w=w, randsteps=r, timesteps=2, bit_scale=0.01, num_classes=K, accumulation=True,
               noise_schedule='cosine', diffusion='ddim')
    eng = _engine(cfg, sd, dev, batch=2, gemm=gemm)
    out = eng.sample(x.to(dev), noise.to(dev)).cpu()
    for b in range(2):
        ref = O.ddim_sample_seg(x[b:b + 1], noise[b], sd, timesteps=2, randsteps=r, bit_scale=0.01, accumulation=True)
        assert max_rel(out[b:b + 1], ref) < REL


@pytest.mark.gpu
def test_sample_more_tiles_than_cus_ragged(dev):
    """33 150 tokens = 259 layer-kernel tiles on 256 CUs: blocks that walk two tiles, and a ragged last tile"""
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    h, w, L, K = 130, 255, 2, 19
    sd = synthetic.make_state_dict('seg', K, L, 256, seed=21)
    x, noise = synthetic.make_inputs(1, h, w, 1, 256, 256, seed=22)
    cfg = dict(task='seg', h=h, w=w, randsteps=1, timesteps=1, bit_scale=0.01, num_classes=K, accumulation=False,
               noise_schedule='cosine', diffusion='ddim')
    eng = _engine(cfg, sd, dev, batch=1)
    out = eng.sample(x.to(dev), noise.to(dev)).cpu()
    ref = O.ddim_sample_seg(x, noise[0], sd, timesteps=1, randsteps=1, bit_scale=0.01, accumulation=False)
    assert max_rel(out, ref) < REL


@pytest.mark.gpu
@pytest.mark.parametrize('h,w', [(37, 53), (8, 16), (70, 9)])
def test_sample_spread_offsets_ragged_maps(dev, h, w):
    """the LDS-staged gather's mixed path (sampling offsets spread far beyond the window: 'trained_like' profile) on maps that are
    not multiples of its 8 x 16 tile, windows clamped at every border, through one decoder pass of two images.  This profile
    amplifies rounding, so the bar is the fp32 oracle's own distance to its fp64 evaluation."""
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    K = 19
    sd = synthetic.make_state_dict('seg', K, 6, 256, seed=31, profile='trained_like')
    sdd = {k: v.double() for k, v in sd.items()}
    x, noise = synthetic.make_inputs(2, h, w, 1, 256, 256, seed=32)
    cfg = dict(task='seg', h=h, w=w, randsteps=1, timesteps=1, bit_scale=0.01, num_classes=K, accumulation=False,
               noise_schedule='cosine', diffusion='ddim')
    out = _engine(cfg, sd, dev, batch=2).sample(x.to(dev), noise.to(dev)).cpu()
    out_u = _engine(cfg, sd, dev, batch=2, fused_layer=False, fused_prologue=False).sample(x.to(dev), noise.to(dev)).cpu()
    assert torch.isfinite(out).all()
    for b in range(2):
        r32 = O.ddim_sample_seg(x[b:b + 1], noise[b], sd, timesteps=1, randsteps=1, bit_scale=0.01, accumulation=False)
        r64 = O.ddim_sample_seg(x[b:b + 1].double(), noise[b].double(), sdd, timesteps=1, randsteps=1, bit_scale=0.01, accumulation=False)
        dc = float((r32.double() - r64).abs().max())
        scale = float(r64.abs().max())
        for name, o in (('fused', out), ('unfused', out_u)):
            dg = float((o[b:b + 1].double() - r64).abs().max())
            print(f'spread offsets {h}x{w} image {b} {name}: gpu vs fp64 {dg:.3e}, fp32 oracle vs fp64 {dc:.3e} (scale {scale:.1f})')
            assert dg <= 4 * dc + 1e-5 * scale


# ---- the exported stand-alone DDIM update, and NaN robustness of the argmax -> LUT step -----------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('K,ldl,rows', [(150, 160, 1000), (19, 32, 333), (256, 256, 64), (2, 32, 5)])
def test_ddim_update_seg_entry(dev, K, ldl, rows):
    """ddp_ddim_update_seg (x0 = LUT[argmax], DDIM step; segmentors/ddp.py:235-239) on token-major buffers vs torch"""
    from ddp_amd import _lib, schedule
    lib = _lib.load()
    g = torch.Generator().manual_seed(K * 1000 + rows)
    logits = torch.randn(rows, ldl, generator=g)
    logits[:, K:] = 100.0                                    # padding columns must be ignored
    logits[::7, 3 % K] = logits[::7, :K].max(1).values      # ties: the first maximum wins (torch.argmax)
    emb = torch.randn(K + 1, 256, generator=g)
    bit_scale = 0.01
    lut = (torch.sigmoid(emb) * 2 - 1) * bit_scale
    mask = torch.randn(rows, 256, generator=g)
    rec = schedule.step_records('seg', 3)[1]
    step = _lib.DdpStep()
    for k, v in rec.items():
        setattr(step, k, v)
    d_logits, d_lut, d_mask = logits.to(dev), lut.to(dev), mask.clone().to(dev)
    _lib.check(lib.ddp_ddim_update_seg(d_logits.data_ptr(), ldl, K, d_lut.data_ptr(), d_mask.data_ptr(), rows, C.byref(step),
                                       torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    idx = logits[:, :K].argmax(1)
    x0 = lut[idx]
    pred_noise = (mask - rec['alpha'] * x0) / max(rec['sigma'], 1e-8)
    ref = x0 * rec['alpha_next'] + pred_noise * rec['sigma_next']
    assert max_rel(d_mask.cpu(), ref) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('variant', ['bf16x3', 'f32', 'bf16x3-unfused-layer', 'ddpm'])
def test_nan_input_does_not_fault(dev, variant):
    """a NaN in x spreads over its token's scores through LayerNorm: argmax finds no maximum.  Every update kernel must
    keep the LUT index in range (NaN propagates, nothing faults) and a clean call afterwards is unaffected."""
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    sd = synthetic.make_state_dict('seg', 19, 2, 256, seed=31)
    h, w = 9, 14
    x, noise = synthetic.make_inputs(1, h, w, 1, 256, 256, seed=32)
    cfg = dict(task='seg', h=h, w=w, randsteps=1, timesteps=2, bit_scale=0.01, num_classes=19, accumulation=True,
               noise_schedule='cosine', diffusion='ddpm' if variant == 'ddpm' else 'ddim')
    over = VARIANTS[variant] if variant in VARIANTS else {}
    eng = _engine(cfg, sd, dev, **over)
    sn = torch.randn(2, 1, 1, 256, h, w).to(dev) if variant == 'ddpm' else None
    bad = x.clone()
    bad[0, :, 4, 7] = float('nan')
    bad[0, 5, 0, 0] = float('inf')
    out = eng.sample(bad.to(dev), noise.to(dev), sn)
    torch.cuda.synchronize()                                 # an out-of-range LUT read would fault here
    assert torch.isnan(out[0, :, 4, 7]).any()
    if variant != 'ddpm':
        out = eng.sample(x.to(dev), noise.to(dev), sn).cpu()
        ref = O.ddim_sample_seg(x, noise[0], sd, timesteps=2, randsteps=1, bit_scale=0.01, accumulation=True)
        assert max_rel(out, ref) < REL


# ---- self-aligned pre-pass, sampler loop around FCNHeadWithTime ----------------------------------------
from golden_util import load_aligned_case, load_loopfcn_case  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize('name', case_names('aligned'))
def test_self_aligned_prepass_golden(dev, name):
    """SelfAlignedDDP.self_aligned_predict (one decoder pass at t = 1 + ddp_seg_x0_project) vs the tensors recorded
    inside the reference's forward_train (self_aligned_ddp.py:150-164)."""
    import ddp_amd
    cfg, sd, x, noise, g = load_aligned_case(name)
    from test_host_logic import seg_cfg
    mc = seg_cfg(type='SelfAlignedDDP', bit_scale=cfg['bit_scale'], noise_schedule=cfg['noise_schedule'])
    mc['decode_head']['num_classes'] = cfg['num_classes']
    model = ddp_amd.build_segmentor(mc)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    preds, logits = model.self_aligned_predict(x.to(dev), noise.to(dev), return_logits=True)
    torch.cuda.synchronize()
    assert max_rel(logits.cpu(), g['logits']) < REL
    same = (logits.cpu().argmax(1) == g['logits'].argmax(1))
    assert same.float().mean() > 0.999
    # x0 rows are table look-ups through sigmoid: equal to a few ulp of bit_scale wherever the argmax agrees
    diff = (preds.cpu() - g['preds']).abs().amax(1)
    assert float(diff[same].max()) < 1e-6 * cfg['bit_scale']


@pytest.mark.gpu
@pytest.mark.parametrize('name', case_names('loopfcn'))
def test_sampler_loop_around_fcn_head_golden(dev, name):
    """DDP(decode_head=FCNHeadWithTime).ddim_sample / ddpm_sample (ddp_sample_fcn) vs the reference sampler driving the
    reference FCN head."""
    import ddp_amd
    cfg, sd, x, noise, step_noise, g = load_loopfcn_case(name)
    model = ddp_amd.build_segmentor(dict(
        type='DDP', timesteps=cfg['timesteps'], randsteps=cfg['randsteps'], bit_scale=cfg['bit_scale'],
        accumulation=cfg['accumulation'], diffusion=cfg['diffusion'],
        decode_head=dict(type='FCNHeadWithTime', num_convs=cfg['num_convs'], concat_input=cfg['concat_input'],
                         dilation=cfg['dilation'], in_channels=256, channels=256, num_classes=cfg['num_classes'], in_index=0,
                         norm_cfg=dict(type='BN') if cfg['with_norm'] else None)))
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    dn = noise.unsqueeze(0).contiguous().to(dev)
    if cfg['diffusion'] == 'ddpm':
        out = model.ddpm_sample(x.to(dev), noise=dn, step_noise=step_noise.unsqueeze(1).contiguous().to(dev))
    else:
        out = model.ddim_sample(x.to(dev), noise=dn)
    torch.cuda.synchronize()
    assert out.shape == g['out'].shape
    err = max_rel(out.cpu(), g['out'])
    agree = (out.cpu().argmax(1) == g['out'].argmax(1)).float().mean().item()
    print(f'{name}: max-rel {err:.3e}, argmax agreement {agree:.4f}')
    assert err < REL and agree > 0.999
    # the engine ran prepared (ddp_prepare_fcn once, then DDP_FLAG_FCN_PREPARED); a second call re-uses the constants and the
    # self-preparing form of the C entry (flag off: every constant rebuilt inside the call) gives the same bits
    import ctypes as C
    from ddp_amd import _lib
    eng = next(reversed(model._engine_cache.values()))
    assert eng._prepared and (eng.cfg.flags & _lib.FLAG_FCN_PREPARED)
    dx = x.to(dev)
    dsn = step_noise.unsqueeze(1).contiguous().to(dev) if cfg['diffusion'] == 'ddpm' else None
    again = eng.sample(dx, dn, dsn).clone()
    assert torch.equal(again, out)
    eng.workspace.zero_()                                      # nothing prepared any more
    eng.cfg.flags &= ~_lib.FLAG_FCN_PREPARED
    raw = torch.empty_like(out)
    with torch.cuda.device(dev):
        _lib.check(eng.lib.ddp_sample_fcn(C.byref(eng.cfg), C.byref(eng.weights.struct), eng.convs, eng.head.num_convs, eng.head.dilation,
                                          eng.steps, dx.data_ptr(), dn.data_ptr(), dsn.data_ptr() if dsn is not None else None,
                                          raw.data_ptr(), eng.workspace.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), eng.lib)
    assert torch.equal(raw, out)
    eng.cfg.flags |= _lib.FLAG_FCN_PREPARED                     # (the workspace holds the constants again: the call above rebuilt them)
    assert torch.equal(eng.sample(dx, dn, dsn), out)
