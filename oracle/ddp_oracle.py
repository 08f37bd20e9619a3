"""CPU ORACLE for the DDP multi-step denoising inference loop.  TEST INFRASTRUCTURE, NOT PRODUCT.

A functional, mmcv-free restatement (torch CPU fp32 ops) of the reference's hot path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module - and only as the checker / the CPU baseline, never as the thing shipped: ``ddp_amd``
(the product) never imports it and fails loudly when its HIP library is missing.

PARITY PIN.  The reference holds no tests, golden vectors or fixtures for this path (SURVEY.md
§4, §8c: "parity unpinned" by the reference's own tests).  The oracle is therefore pinned against
outputs of the reference itself, imported in the build container through
``tests/golden/ref_shim.py`` and run on seeded inputs by ``tests/golden/gen_golden.py``; the
resulting vectors are committed under ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py`` checks this file against every one of them (plus the
known-answer values of the schedule and positional encoding recorded in SURVEY.md §3.2/§8 a9).
Caveat recorded with the fixtures: the reference's arithmetic core lives in the third-party
mmcv-full==1.6.2 (segmentation/README.md:25) which is not vendored; the import used the in-tree
pure-python mmcv 1.3.17 (controlnet/annotator/uniformer/mmcv) on torch 2.10 CPU.

Every function cites the reference lines it follows.  Paths are relative to /root/reference;
``MSDA`` = controlnet/annotator/uniformer/mmcv/ops/multi_scale_deform_attn.py,
``XFMR`` = segmentation/mmseg/models/utils/transformer.py,
``SEGDDP`` = segmentation/mmseg/models/segmentors/ddp.py,
``DHWT`` = segmentation/mmseg/models/decode_heads/deformable_head_with_time.py.
"""
import math

import torch
import torch.nn.functional as F

EMBED = 256
HEADS = 8
POINTS = 4

# test hook: when a list, msda_forward appends the largest |sampling offset| (pixels) of every call - the radius by which one
# decoder layer can carry a changed input sideways (tests/test_full_size_parity.py: dependency cone of a flipped decision)
OFFSET_LOG = None


# --------------------------------------------------------------------------------------------
# schedules  (SEGDDP:14-28; depth/depth/models/depther/ddp.py:207-208)
# --------------------------------------------------------------------------------------------
def log_clamped(t, eps=1e-20):
    """SEGDDP:14-15."""
    return torch.log(t.clamp(min=eps))


def beta_linear_log_snr(t):
    """SEGDDP:18-19."""
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))


def alpha_cosine_log_snr(t, ns=0.0002, ds=0.00025):
    """SEGDDP:22-24.  Evaluated in fp32 in exactly this op order: the value at t=1 is
    ill-conditioned (SURVEY.md §7 hard part 8)."""
    return -log_clamped((torch.cos((t + ns) / (1 + ds) * math.pi * 0.5) ** -2) - 1, eps=1e-5)


def log_snr_to_alpha_sigma(log_snr):
    """SEGDDP:27-28."""
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


def gamma_cosine(t, ns=0.0002, ds=0.00025):
    """depth/depth/models/depther/ddp.py:207-208."""
    return torch.cos(((t + ns) / (1 + ds)) * math.pi / 2) ** 2


def sampling_time_pairs(timesteps, time_difference=1, sample_range0=0.0):
    """SEGDDP:204-213 (depth/bev: same with sample_range0 == 0).  -> list of (t_now, t_next)
    python floats; the reference wraps them in torch.tensor([t_now, t_next]) (fp32)."""
    pairs = []
    for step in range(timesteps):
        t_now = 1 - (step / timesteps) * (1 - sample_range0)
        t_next = max(1 - (step + 1 + time_difference) / timesteps * (1 - sample_range0), sample_range0)
        pairs.append((t_now, t_next))
    return pairs


# --------------------------------------------------------------------------------------------
# time embedding + per-layer FiLM  (SEGDDP:31-46,107-112; XFMR:275-278,413-417)
# --------------------------------------------------------------------------------------------
def learned_sinusoidal(x, weights):
    """SEGDDP:41-46.  x (r,) -> (r, 17): [x, sin(2 pi x w), cos(2 pi x w)]."""
    x = x[:, None]
    freqs = x * weights[None, :] * 2 * math.pi
    return torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)


def time_mlp(t_in, sd):
    """SEGDDP:107-112: Linear(17,1024) -> GELU(erf) -> Linear(1024,1024)."""
    u = learned_sinusoidal(t_in, sd['time_mlp.0.weights'])
    hid = F.gelu(F.linear(u, sd['time_mlp.1.weight'], sd['time_mlp.1.bias']))
    return F.linear(hid, sd['time_mlp.3.weight'], sd['time_mlp.3.bias'])


def film_vectors(temb, sd, layer):
    """XFMR:275-278,413-416: Linear(1024,512)(SiLU(temb)) -> (scale, shift) each (r,256)."""
    p = f'decode_head.encoder.layers.{layer}.time_mlp.1.'
    g = F.linear(F.silu(temb), sd[p + 'weight'], sd[p + 'bias'])
    return g[:, :EMBED], g[:, EMBED:]


# --------------------------------------------------------------------------------------------
# positional encoding / reference points  (XFMR:78-113; DHWT:63-88)
# --------------------------------------------------------------------------------------------
def sine_positional_encoding(h, w, num_feats=128, temperature=10000, scale=2 * math.pi, eps=1e-6,
                             offset=-0.5):
    """XFMR:78-113 with normalize=True and an all-valid mask.  -> (256, h, w)."""
    ones = torch.ones((1, h, w), dtype=torch.int)
    y_embed = ones.cumsum(1, dtype=torch.float32)
    x_embed = ones.cumsum(2, dtype=torch.float32)
    y_embed = (y_embed + offset) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed + offset) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(1, h, w, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(1, h, w, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)[0]


def reference_points(h, w):
    """DHWT:63-88 for a single level.  -> (N, 2) as (x, y) in [0,1]."""
    ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, dtype=torch.float32),
                                  torch.linspace(0.5, w - 0.5, w, dtype=torch.float32), indexing='ij')
    return torch.stack((ref_x.reshape(-1) / w, ref_y.reshape(-1) / h), -1)


# --------------------------------------------------------------------------------------------
# deformable attention  (MSDA:94-151 core, MSDA:299-358 module forward)
# --------------------------------------------------------------------------------------------
def msda_core_gridsample(value, h, w, sampling_locations, attention_weights):
    """MSDA:94-151 (``multi_scale_deformable_attn_pytorch``) for one level.
    value (bs, N, heads, d); sampling_locations (bs, Nq, heads, P, 2) in [0,1];
    attention_weights (bs, Nq, heads, P) -> (bs, Nq, heads*d)."""
    bs, n, heads, d = value.shape
    nq = sampling_locations.shape[1]
    grids = 2 * sampling_locations - 1
    value_l = value.flatten(2).transpose(1, 2).reshape(bs * heads, d, h, w)
    grid_l = grids.transpose(1, 2).flatten(0, 1)                      # (bs*heads, Nq, P, 2)
    sampled = F.grid_sample(value_l, grid_l, mode='bilinear', padding_mode='zeros', align_corners=False)
    aw = attention_weights.transpose(1, 2).reshape(bs * heads, 1, nq, POINTS)
    out = (sampled * aw).sum(-1).view(bs, heads * d, nq)
    return out.transpose(1, 2).contiguous()


def msda_core_taps(value, h, w, px, py, attention_weights):
    """Independent restatement of the same sampling with explicit 4-corner bilinear taps in PIXEL
    units (for a single level the sample point of token (i,j) is simply
    (j + o_x, i + o_y); taps outside [0,w-1]x[0,h-1] contribute 0).  px, py (bs,Nq,heads,P)."""
    bs, n, heads, d = value.shape
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    fx = px - x0
    fy = py - y0
    out = torch.zeros(bs, px.shape[1], heads, d)
    vb = value.permute(0, 2, 1, 3)                                     # (bs, heads, N, d)
    for dy, dx, wgt in ((0, 0, (1 - fy) * (1 - fx)), (0, 1, (1 - fy) * fx),
                        (1, 0, fy * (1 - fx)), (1, 1, fy * fx)):
        xi = (x0 + dx).long()
        yi = (y0 + dy).long()
        valid = ((xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)).float()
        idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1))              # (bs,Nq,heads,P)
        coef = wgt * valid * attention_weights
        for p in range(POINTS):
            ind = idx[..., p].permute(0, 2, 1)                           # (bs, heads, Nq)
            g = torch.gather(vb, 2, ind[..., None].expand(-1, -1, -1, d))  # (bs, heads, Nq, d)
            out += g.permute(0, 2, 1, 3) * coef[..., p][..., None]
    return out.reshape(bs, px.shape[1], heads * d)


def msda_forward(q, pos, h, w, sd, prefix, core='gridsample'):
    """MSDA:299-358 with batch_first=False folded away: q (bs,N,256) token-major, pos (N,256).
    value from q WITHOUT pos; offsets / attention logits from q+pos; residual = q; dropout p=0."""
    bs, n, _ = q.shape
    qp = q + pos[None]
    value = F.linear(q, sd[prefix + 'value_proj.weight'], sd[prefix + 'value_proj.bias'])
    value = value.view(bs, n, HEADS, -1)
    off = F.linear(qp, sd[prefix + 'sampling_offsets.weight'], sd[prefix + 'sampling_offsets.bias'])
    off = off.view(bs, n, HEADS, POINTS, 2)
    if OFFSET_LOG is not None:
        OFFSET_LOG.append(float(off.abs().max()))
    aw = F.linear(qp, sd[prefix + 'attention_weights.weight'], sd[prefix + 'attention_weights.bias'])
    aw = aw.view(bs, n, HEADS, POINTS).softmax(-1)
    if core == 'gridsample':
        ref = reference_points(h, w)                                    # (N,2)
        normalizer = torch.tensor([w, h], dtype=torch.float32)          # MSDA:330-331 (w, h)
        loc = ref[None, :, None, None, :] + off / normalizer
        s = msda_core_gridsample(value, h, w, loc, aw)
    else:
        jj = torch.arange(w, dtype=torch.float32).repeat(h)
        ii = torch.arange(h, dtype=torch.float32).repeat_interleave(w)
        px = jj[None, :, None, None] + off[..., 0]
        py = ii[None, :, None, None] + off[..., 1]
        s = msda_core_taps(value, h, w, px, py, aw)
    out = F.linear(s, sd[prefix + 'output_proj.weight'], sd[prefix + 'output_proj.bias'])
    return out + q


def encoder_layer(q, pos, temb, h, w, sd, layer, core='gridsample'):
    """XFMR:317-419 with operation_order ('self_attn','norm','ffn','norm') + FiLM (XFMR:413-417);
    FFN = controlnet/annotator/uniformer/mmcv/cnn/bricks/transformer.py:269-280."""
    p = f'decode_head.encoder.layers.{layer}.'
    y = msda_forward(q, pos, h, w, sd, p + 'attentions.0.', core)
    q1 = F.layer_norm(y, (EMBED,), sd[p + 'norms.0.weight'], sd[p + 'norms.0.bias'], 1e-5)
    hid = F.gelu(F.linear(q1, sd[p + 'ffns.0.layers.0.0.weight'], sd[p + 'ffns.0.layers.0.0.bias']))
    y2 = q1 + F.linear(hid, sd[p + 'ffns.0.layers.1.weight'], sd[p + 'ffns.0.layers.1.bias'])
    q2 = F.layer_norm(y2, (EMBED,), sd[p + 'norms.1.weight'], sd[p + 'norms.1.bias'], 1e-5)
    if temb is not None and (p + 'time_mlp.1.weight') in sd:
        scale, shift = film_vectors(temb, sd, layer)
        q2 = q2 * (scale[:, None, :] + 1) + shift[:, None, :]
    return q2


def num_layers_of(sd):
    n = 0
    while f'decode_head.encoder.layers.{n}.norms.0.weight' in sd:
        n += 1
    return n


def encoder_forward(feat, temb, sd, core='gridsample', trace=None):
    """DHWT:90-128: flatten NCHW -> tokens, sine pos-enc, L layers.  feat (bs,256,h,w) ->
    memory (bs, N, 256) token-major."""
    bs, c, h, w = feat.shape
    pos = sine_positional_encoding(h, w).flatten(1).transpose(0, 1)      # (N,256)
    q = feat.flatten(2).transpose(1, 2)                                  # (bs,N,256)
    for l in range(num_layers_of(sd)):
        q = encoder_layer(q, pos, temb, h, w, sd, l, core)
        if trace is not None:
            trace.append(q.clone())
    return q


def head_forward_seg(feat, temb, sd, core='gridsample', trace=None):
    """DHWT:90-132: encoder then conv_seg (1x1, dropout bypassed).  -> logits (bs,K,h,w)."""
    bs, c, h, w = feat.shape
    mem = encoder_forward(feat, temb, sd, core, trace)
    mem = mem.permute(0, 2, 1).reshape(bs, c, h, w).contiguous()
    return F.conv2d(mem, sd['decode_head.conv_seg.weight'], sd['decode_head.conv_seg.bias'])


def head_forward_depth(feat, temb, sd, min_depth=1e-3, core='gridsample', trace=None, scale_up=False, use_eps=True, max_depth=80.0):
    """depth/depth/models/decode_heads/deformable_head_with_time.py:90-131 +
    depth/depth/models/decode_heads/decode_head.py:100,252-262: relu(conv3x3(memory)) + eps (eps = min_depth, or 0 with
    use_eps=False), or with scale_up sigmoid(conv3x3(memory)) * eps (eps = max_depth, or 1)."""
    bs, c, h, w = feat.shape
    mem = encoder_forward(feat, temb, sd, core, trace)
    mem = mem.permute(0, 2, 1).reshape(bs, c, h, w).contiguous()
    d = F.conv2d(mem, sd['decode_head.conv_depth.weight'], sd['decode_head.conv_depth.bias'], padding=1)
    if scale_up:
        return torch.sigmoid(d) * (max_depth if use_eps else 1)
    return F.relu(d) + (min_depth if use_eps else 0)


# --------------------------------------------------------------------------------------------
# samplers
# --------------------------------------------------------------------------------------------
def x0_from_logits_seg(logits, sd, bit_scale, idx=None):
    """SEGDDP:235-237: argmax -> embedding -> (sigmoid*2-1)*bit_scale.  (r,K,h,w)->(r,256,h,w).
    ``idx`` (r,h,w) replaces the argmax (teacher-forced decisions, see ddim_sample_seg)."""
    if idx is None:
        idx = torch.argmax(logits, dim=1)
    e = F.embedding(idx, sd['embedding_table.weight']).permute(0, 3, 1, 2)
    return (torch.sigmoid(e) * 2 - 1) * bit_scale


def ddim_sample_seg(x, noise, sd, timesteps=3, randsteps=1, bit_scale=0.01, time_difference=1,
                    sample_range0=0.0, noise_schedule='cosine', accumulation=False,
                    core='gridsample', trace=None, head=None, x0_index=None, decisions=None):
    """SEGDDP:215-246 for ONE image.  x (1,256,h,w); noise (r,256,h,w) replaces the in-method
    ``torch.randn`` (SEGDDP:220).  -> (1,K,h,w).  ``head(feat, temb) -> logits`` replaces the decode head that
    ``_decode_head_forward_test`` (SEGDDP:192-196) dispatches to (default: DeformableHeadWithTime).
    ``x0_index``: optional K maps (r,h,w) of class indices used INSTEAD of argmax(logits) in the x0 projection - the
    loop's only discrete decision.  Feeding the decisions another implementation took removes the feedback
    discontinuity from a comparison (everything else is continuous in the inputs).
    ``decisions``: optional list that receives the class map (r,h,w) uint8 every step fed back (a light-weight
    alternative to ``trace`` at full size)."""
    log_snr_fn = alpha_cosine_log_snr if noise_schedule == 'cosine' else beta_linear_log_snr
    xr = x.repeat(randsteps, 1, 1, 1)
    mask_t = noise.clone()
    outs = []
    logits = None
    for step, (t_now, t_next) in enumerate(sampling_time_pairs(timesteps, time_difference, sample_range0)):
        times_now = torch.tensor([t_now], dtype=torch.float32)
        times_next = torch.tensor([t_next], dtype=torch.float32)
        feat = torch.cat([xr, mask_t], dim=1)
        feat = F.conv2d(feat, sd['transform.conv.weight'], sd['transform.conv.bias'])
        log_snr = log_snr_fn(times_now)
        log_snr_next = log_snr_fn(times_next)
        alpha, sigma = log_snr_to_alpha_sigma(log_snr.view(-1, 1, 1, 1))
        alpha_next, sigma_next = log_snr_to_alpha_sigma(log_snr_next.view(-1, 1, 1, 1))
        temb = time_mlp(log_snr, sd)
        layer_trace = [] if trace is not None else None
        logits = head(feat, temb) if head is not None else head_forward_seg(feat, temb, sd, core, layer_trace)
        x0 = x0_from_logits_seg(logits, sd, bit_scale, None if x0_index is None else x0_index[step].long())
        if decisions is not None:
            decisions.append((torch.argmax(logits, dim=1) if x0_index is None else x0_index[step]).to(torch.uint8))
        pred_noise = (mask_t - alpha * x0) / sigma.clamp(min=1e-8)
        mask_t = x0 * alpha_next + pred_noise * sigma_next
        if accumulation:
            outs.append(logits.softmax(1))
        if trace is not None:
            trace.append(dict(feat=feat, temb=temb, layers=layer_trace, logits=logits, mask_t=mask_t.clone()))
    if accumulation:
        logits = torch.cat(outs, dim=0)
    return logits.mean(dim=0, keepdim=True)


def reference_drift_seg(x, noise, sd, *, timesteps, accumulation, bit_scale=0.01, randsteps=1, variants=('taps', 'fp64'),
                        base=None, base_decisions=None):
    """How far does the REFERENCE, restated twice, drift from itself when the sampler runs freely (each run taking its
    own argmax, SEGDDP:235)?  Base = this oracle as pinned (fp32, ``F.grid_sample`` core = the reference's CPU path,
    MSDA:94-151); variants differ from it only in ways the reference itself does between deployments:
      'taps'  the deformable-attention core as explicit 4-corner bilinear taps in pixel units - the arithmetic of mmcv's
              compiled ``ms_deform_attn_forward`` (MSDA:47-53), i.e. the reference's own GPU path, still fp32;
      'fp64'  every tensor and weight in double precision (the schedule scalars stay the reference's fp32 values).
    The loop is continuous in its inputs except for the argmax fed back each step, so any two fp32-class evaluations
    agree to rounding until a near-tie pixel falls the other way; from then on that neighbourhood differs by O(1e-3).
    Returns {'base': scores, 'ref_vs_ref': worst max-rel over the variants, 'variants': {name: {max_rel,
    pixels_above_1e-4, decisions_differ}}} - the yardstick the engine's free-running distance is judged against
    (tests/test_full_size_parity.py, bench.py)."""
    if base is None:
        base_decisions = []
        base = ddim_sample_seg(x, noise, sd, timesteps=timesteps, randsteps=randsteps, bit_scale=bit_scale,
                               accumulation=accumulation, decisions=base_decisions)
    res = {}
    for v in variants:
        dec = []
        if v == 'taps':
            o = ddim_sample_seg(x, noise, sd, timesteps=timesteps, randsteps=randsteps, bit_scale=bit_scale,
                                accumulation=accumulation, core='taps', decisions=dec)
        elif v == 'fp64':
            o = ddim_sample_seg(x.double(), noise.double(), {k: t.double() for k, t in sd.items()}, timesteps=timesteps,
                                randsteps=randsteps, bit_scale=bit_scale, accumulation=accumulation, decisions=dec)
        else:
            raise ValueError(v)
        rel = (o.to(base.dtype) - base).abs().amax(1) / base.abs().max()
        res[v] = {'max_rel': float(rel.max()), 'pixels_above_1e-4': int((rel > 1e-4).sum()),
                  'decisions_differ': (sum(int((a != b).sum()) for a, b in zip(dec, base_decisions))
                                       if base_decisions is not None else None)}
    return {'base': base, 'base_decisions': base_decisions, 'ref_vs_ref': max(r['max_rel'] for r in res.values()) if res else 0.0,
            'variants': res}


def ddpm_sample_seg(x, noise, step_noise, sd, timesteps=3, randsteps=1, bit_scale=0.01,
                    time_difference=1, sample_range0=0.0, noise_schedule='cosine', accumulation=False,
                    core='gridsample', head=None):
    """SEGDDP:248-290.  ``step_noise`` (timesteps, r,256,h,w) replaces the per-step
    ``torch.randn_like`` (SEGDDP:280).  ``head``: see ddim_sample_seg."""
    log_snr_fn = alpha_cosine_log_snr if noise_schedule == 'cosine' else beta_linear_log_snr
    xr = x.repeat(randsteps, 1, 1, 1)
    mask_t = noise.clone()
    outs = []
    logits = None
    for s, (t_now, t_next) in enumerate(sampling_time_pairs(timesteps, time_difference, sample_range0)):
        times_now = torch.tensor([t_now], dtype=torch.float32)
        times_next = torch.tensor([t_next], dtype=torch.float32)
        feat = F.conv2d(torch.cat([xr, mask_t], dim=1), sd['transform.conv.weight'], sd['transform.conv.bias'])
        log_snr = log_snr_fn(times_now)
        log_snr_next = log_snr_fn(times_next)
        pl = log_snr.view(-1, 1, 1, 1)
        pln = log_snr_next.view(-1, 1, 1, 1)
        alpha, sigma = log_snr_to_alpha_sigma(pl)
        alpha_next, sigma_next = log_snr_to_alpha_sigma(pln)
        temb = time_mlp(log_snr, sd)
        logits = head(feat, temb) if head is not None else head_forward_seg(feat, temb, sd, core)
        x0 = x0_from_logits_seg(logits, sd, bit_scale)
        # times carry the image batch b == 1 (SEGDDP:204-212), so every schedule quantity is a
        # single scalar broadcast over the r noise replicas.
        c = -torch.special.expm1(pl - pln)
        mean = alpha_next * (mask_t * (1 - c) / alpha + c * x0)
        variance = (sigma_next ** 2) * c
        log_variance = log_clamped(variance)
        nz = step_noise[s] if t_next > 0 else torch.zeros_like(mask_t)
        mask_t = mean + (0.5 * log_variance).exp() * nz
        if accumulation:
            outs.append(logits.softmax(1))
    if accumulation:
        logits = torch.cat(outs, dim=0)
    return logits.mean(dim=0, keepdim=True)


def self_aligned_predict(x, noise, sd, bit_scale=0.01, noise_schedule='cosine', core='gridsample'):
    """The no-grad "self-aligned denoising" pre-pass of SelfAlignedDDP.forward_train
    (segmentation/mmseg/models/segmentors/self_aligned_ddp.py:150-164): times = 1, noise = randn_like(x) (injected
    here), feat = transform(cat[x, noise]), logits = decode_head(feat, time_mlp(log_snr(1))), preds = argmax ->
    embedding -> (sigmoid*2-1)*bit_scale.  x, noise (b,256,h,w) -> (preds (b,256,h,w), logits (b,K,h,w))."""
    log_snr_fn = alpha_cosine_log_snr if noise_schedule == 'cosine' else beta_linear_log_snr
    b = x.shape[0]
    times = torch.ones((b,), dtype=x.dtype)
    temb = time_mlp(log_snr_fn(times), sd)
    feat = F.conv2d(torch.cat([x, noise], dim=1), sd['transform.conv.weight'], sd['transform.conv.bias'])
    logits = head_forward_seg(feat, temb, sd, core)
    return x0_from_logits_seg(logits, sd, bit_scale), logits


def fcn_head_for_sampler(sd, num_convs, dilation=1, prefix='decode_head.'):
    """``head`` argument of ddim_sample_seg / ddpm_sample_seg for FCNHeadWithTime as the segmentor's decode head
    (fcn_head_with_time.py:327-343 forward_test -> forward)."""
    return lambda feat, temb: fcn_head_forward(feat, temb, sd, num_convs, dilation, prefix)


def sample_depth(x, noise, sd, timesteps=3, randsteps=1, bit_scale=0.1, time_difference=1,
                 min_depth=1e-3, max_depth=80.0, core='gridsample', trace=None, scale_up=False, use_eps=True):
    """depth/depth/models/depther/ddp.py:229-247 (+ ddim_step :220-227) for ONE image.
    x (1,256,h,w); noise (r,1,h,w).  -> (1,1,h,w) metric depth (before the encode_decode clamp)."""
    xr = x.repeat(randsteps, 1, 1, 1)
    depth_t = noise.clone()
    depth_pred = None
    for t_now, t_next in sampling_time_pairs(timesteps, time_difference, 0.0):
        times_now = torch.tensor([t_now], dtype=torch.float32)
        times_next = torch.tensor([t_next], dtype=torch.float32)
        feat = F.conv2d(torch.cat([xr, depth_t], dim=1), sd['down.conv.weight'], sd['down.conv.bias'])
        temb = time_mlp(times_now, sd)                                   # raw t, not log-snr (:238)
        depth_pred = head_forward_depth(feat, temb, sd, min_depth, core, None, scale_up, use_eps, max_depth)
        x0 = (depth_pred - min_depth) / (max_depth - min_depth)
        x0 = (x0 * 2 - 1) * bit_scale
        a_now = gamma_cosine(times_now.view(-1, 1, 1, 1))
        a_next = gamma_cosine(times_next.view(-1, 1, 1, 1))
        x0 = x0.clamp(-bit_scale, bit_scale)
        eps = (1 / (1 - a_now).sqrt()) * (depth_t - a_now.sqrt() * x0)
        depth_t = a_next.sqrt() * x0 + (1 - a_next).sqrt() * eps
        if trace is not None:
            trace.append(dict(feat=feat, temb=temb, depth_pred=depth_pred, depth_t=depth_t.clone()))
    return depth_pred.mean(dim=0, keepdim=True)


# --------------------------------------------------------------------------------------------
# BEV  (bev/mmdet3d/models/heads/segm/deformable_head_with_time.py:57-97,179-235;
#       bev/mmdet3d/models/fusion_models/ddp.py:268-301)
# --------------------------------------------------------------------------------------------
def bev_grid_transform(feat, input_scope=((-51.2, 51.2, 0.8), (-51.2, 51.2, 0.8)),
                       output_scope=((-50, 50, 0.5), (-50, 50, 0.5))):
    """BEVGridTransform.forward, bev/.../heads/segm/deformable_head_with_time.py:70-97."""
    coords = []
    for (imin, imax, _), (omin, omax, ostep) in zip(input_scope, output_scope):
        v = torch.arange(omin + ostep / 2, omax, ostep)
        v = (v - imin) / (imax - imin) * 2 - 1
        coords.append(v)
    u, v = torch.meshgrid(coords, indexing='ij')
    grid = torch.stack([v, u], dim=-1)
    grid = torch.stack([grid] * feat.shape[0], dim=0)
    return F.grid_sample(feat, grid.to(feat.dtype), mode='bilinear', align_corners=False)   # (no-op in fp32; the fp64 diagnostic)


def head_forward_bev(feat, temb, sd, core='gridsample', **scopes):
    """bev/.../heads/segm/deformable_head_with_time.py:179-235: grid transform, encoder at the
    output resolution, conv_seg 1x1, sigmoid."""
    ft = bev_grid_transform(feat, **scopes)
    bs, c, h, w = ft.shape
    mem = encoder_forward(ft, temb, sd, core)
    mem = mem.permute(0, 2, 1).reshape(bs, c, h, w).contiguous()
    out = F.conv2d(mem, sd['decode_head.conv_seg.weight'], sd['decode_head.conv_seg.bias'])
    return torch.sigmoid(out)


def ddim_sample_bev(x, noise, sd, timesteps=3, randsteps=1, bit_scale=0.01, time_difference=1,
                    threshold=0.5, num_classes=6, core='gridsample', **scopes):
    """bev/mmdet3d/models/fusion_models/ddp.py:268-301 for ONE sample.  x (1,Cx,h,w); noise
    (r,256,h,w).  -> (1,6,H_out,W_out) mean over steps*r of sigmoid maps."""
    h, w = x.shape[-2:]
    xr = x.repeat(randsteps, 1, 1, 1)
    mask_t = noise.clone()
    outs = []
    for t_now, t_next in sampling_time_pairs(timesteps, time_difference, 0.0):
        times_now = torch.tensor([t_now], dtype=torch.float32)
        times_next = torch.tensor([t_next], dtype=torch.float32)
        feat = F.conv2d(torch.cat([xr, mask_t], dim=1), sd['transform.conv.weight'], sd['transform.conv.bias'])
        log_snr = alpha_cosine_log_snr(times_now)
        log_snr_next = alpha_cosine_log_snr(times_next)
        alpha, sigma = log_snr_to_alpha_sigma(log_snr.view(-1, 1, 1, 1))
        alpha_next, sigma_next = log_snr_to_alpha_sigma(log_snr_next.view(-1, 1, 1, 1))
        temb = time_mlp(log_snr, sd)
        prob = head_forward_bev(feat, temb, sd, core, **scopes)
        pred = (prob > threshold)
        factor = (torch.arange(num_classes) + 1).view(1, num_classes, 1, 1)
        pred = pred * factor
        pred = F.interpolate(pred.float(), size=(h, w), mode='nearest').to(torch.int64)
        e = F.embedding(pred, sd['embedding_table.weight']).mean(dim=1).permute(0, 3, 1, 2)
        x0 = (torch.sigmoid(e) * 2 - 1) * bit_scale
        pred_noise = (mask_t - alpha * x0) / sigma.clamp(min=1e-8)
        mask_t = x0 * alpha_next + pred_noise * sigma_next
        outs.append(prob)
    return torch.cat(outs, dim=0).mean(dim=0, keepdim=True)


def seg_postprocess(scores, img_size, crop_size=None, out_size=None, align_corners=False, flip=None):
    """Post-loop epilogue of the segmentor, the reference's own op sequence:
    resize to the network input size (segmentors/ddp.py:124-128; mmseg.ops.resize == F.interpolate), crop to img_shape
    and resize to ori_shape (encoder_decoder.py:236-248), softmax (:277), flip (:278-285), argmax (:296).
    scores (B,K,h,w) -> int64 (B,out_h,out_w)."""
    o = F.interpolate(scores, size=tuple(img_size), mode='bilinear', align_corners=align_corners)
    if crop_size is not None:
        o = o[:, :, :crop_size[0], :crop_size[1]]
        o = F.interpolate(o, size=tuple(out_size if out_size is not None else crop_size), mode='bilinear',
                          align_corners=align_corners)
    o = F.softmax(o, dim=1)
    if flip == 'horizontal':
        o = o.flip(dims=(3,))
    elif flip == 'vertical':
        o = o.flip(dims=(2,))
    return o.argmax(dim=1)


def seg_aug_test(scores_list, metas, out_size, align_corners=False):
    """Multi-scale / flip test-time augmentation, the reference's own op sequence: per augmentation ``inference``
    (encoder_decoder.py:251-287) = softmax of ``whole_inference`` (:229-248: resize to the network input [segmentors/
    ddp.py:124-128], crop to img_shape, resize to ori_shape) with the flip undone; then ``aug_test`` (:306-331): running sum,
    / n, argmax.  scores_list[i] (B,K,h_i,w_i); metas[i] = dict(img_size, crop_size, flip) -> (int64 (B,oh,ow), mean prob)."""
    acc = None
    for sc, m in zip(scores_list, metas):
        o = F.interpolate(sc, size=tuple(m['img_size']), mode='bilinear', align_corners=align_corners)
        crop = m.get('crop_size') or m['img_size']
        o = o[:, :, :crop[0], :crop[1]]
        o = F.interpolate(o, size=tuple(out_size), mode='bilinear', align_corners=align_corners)
        o = F.softmax(o, dim=1)
        if m.get('flip') == 'horizontal':
            o = o.flip(dims=(3,))
        elif m.get('flip') == 'vertical':
            o = o.flip(dims=(2,))
        acc = o if acc is None else acc + o
    acc = acc / len(scores_list)
    return acc.argmax(dim=1), acc


def seg_slide_inference(window_scores, ys, xs, crop_hw, img_size, keep_size=None, out_size=None, align_corners=False):
    """Sliding-window inference, the reference's own op sequence (encoder_decoder.py:180-227 over segmentors/ddp.py:114-129):
    per window (row-major over the grid ys x xs) the sampler's low-resolution scores are resized to the window size
    (``encode_decode``) and added into ``preds`` through ``F.pad``; ``preds / count_mat``; crop to img_shape and resize to
    ori_shape when rescaling.  window_scores[i] (B,K,h,w) -> (B,K,out_h,out_w) averaged scores (softmax / flip / argmax:
    ``inference`` / ``simple_test``, :273-296, as in seg_postprocess)."""
    H, W = img_size
    ch, cw = crop_hw
    B, K = window_scores[0].shape[:2]
    preds = torch.zeros((B, K, H, W))
    count = torch.zeros((B, 1, H, W))
    i = 0
    for y1 in ys:
        for x1 in xs:
            logit = F.interpolate(window_scores[i], size=(ch, cw), mode='bilinear', align_corners=align_corners)
            preds += F.pad(logit, (int(x1), int(W - x1 - cw), int(y1), int(H - y1 - ch)))
            count[:, :, y1:y1 + ch, x1:x1 + cw] += 1
            i += 1
    assert (count == 0).sum() == 0
    preds = preds / count
    if keep_size is not None:
        preds = preds[:, :, :keep_size[0], :keep_size[1]]
        preds = F.interpolate(preds, size=tuple(out_size if out_size is not None else keep_size), mode='bilinear',
                              align_corners=align_corners)
    return preds


def depth_postprocess(depth_list, flips, out_size, min_depth, max_depth, align_corners=False):
    """Post-loop epilogue of the depth toolbox, the reference's own op sequence: per augmentation ``encode_decode``
    (depth/depth/models/depther/ddp.py:95-109: clamp to the head's depth range, resize to the network input) and the flip-undo
    of ``inference`` (depth/depth/models/depther/encoder_decoder.py:187-194); then ``aug_test`` (:210-229): in-place running
    sum in list order, ``/= n`` - or, for ONE augmentation, ``simple_test`` (:198-209), which does not divide.
    depth_list[i] (B,1,h_i,w_i) = the sampler's output; flips[i] None | 'horizontal' | 'vertical' -> (B,1,out_h,out_w)."""
    acc = None
    for d, fl in zip(depth_list, flips):
        o = torch.clamp(d, min=min_depth, max=max_depth)
        if tuple(o.shape[2:]) != tuple(out_size):
            o = F.interpolate(o, size=tuple(out_size), mode='bilinear', align_corners=align_corners)
        if fl == 'horizontal':
            o = o.flip(dims=(3,))
        elif fl == 'vertical':
            o = o.flip(dims=(2,))
        acc = o.clone() if acc is None else acc + o
    if len(depth_list) > 1:
        acc = acc / len(depth_list)
    return acc


def neck_multi_stage_merging(levels, sd, align_corners=False, prefix=''):
    """MultiStageMerging.forward (necks/multi_stage_merging.py:40-52): resize every level to level 0's grid, concat,
    down = ConvModule(1024,256,1, bias=False, GN(32), no act) (mmcv ConvModule order conv -> norm)."""
    size = levels[0].shape[2:]
    outs = [F.interpolate(t, size=tuple(size), mode='bilinear', align_corners=align_corners) for t in levels]
    out = torch.cat(outs, dim=1)
    out = F.conv2d(out, sd[prefix + 'down.conv.weight'])
    return F.group_norm(out, 32, sd[prefix + 'down.gn.weight'], sd[prefix + 'down.gn.bias'], eps=1e-5)


def fcn_head_forward(feat, temb, sd, num_convs, dilation=1, prefix=''):
    """FCNHeadWithTime.forward in eval mode (decode_heads/fcn_head_with_time.py:285-305; ConvWithTimeModule.forward
    :205-225): conv3x3 -> norm (eval BatchNorm, if any) -> x*(scale+1)+shift, (scale|shift) = Linear(SiLU(temb)) -> ReLU,
    then cls_seg = conv_seg (dropout is the identity in eval).  conv_cat is built but never called (:285-299)."""
    x = feat
    for i in range(num_convs):
        p = f'{prefix}convs.{i}.'
        x = F.conv2d(x, sd[p + 'conv.weight'], sd.get(p + 'conv.bias'), padding=dilation, dilation=dilation)
        if p + 'bn.weight' in sd:
            x = F.batch_norm(x, sd[p + 'bn.running_mean'], sd[p + 'bn.running_var'], sd[p + 'bn.weight'], sd[p + 'bn.bias'],
                             False, 0.0, 1e-5)
        if temb is not None:
            te = F.linear(F.silu(temb), sd[p + 'time_mlp.1.weight'], sd[p + 'time_mlp.1.bias'])
            scale, shift = te[:, :, None, None].chunk(2, dim=1)
            x = x * (scale + 1) + shift
        x = F.relu(x)
    return F.conv2d(x, sd[prefix + 'conv_seg.weight'], sd[prefix + 'conv_seg.bias'])


def neck_fpn(inputs, sd, prefix=''):
    """FPN.forward of the DDP configs (necks/fpn.py:163-191): laterals = GN(conv1x1), top-down nearest upsample + add,
    outs = GN(conv3x3(lateral)); 4 levels, no activation, no extra levels."""
    lat = []
    for l, x in enumerate(inputs):
        p = f'{prefix}lateral_convs.{l}.'
        lat.append(F.group_norm(F.conv2d(x, sd[p + 'conv.weight']), 32, sd[p + 'gn.weight'], sd[p + 'gn.bias'], eps=1e-5))
    for l in range(len(lat) - 1, 0, -1):
        lat[l - 1] = lat[l - 1] + F.interpolate(lat[l], size=lat[l - 1].shape[2:], mode='nearest')
    outs = []
    for l, x in enumerate(lat):
        p = f'{prefix}fpn_convs.{l}.'
        outs.append(F.group_norm(F.conv2d(x, sd[p + 'conv.weight'], padding=1), 32, sd[p + 'gn.weight'], sd[p + 'gn.bias'],
                                 eps=1e-5))
    return tuple(outs)
