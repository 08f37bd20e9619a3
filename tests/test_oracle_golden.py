"""Pin the CPU oracle: known-answer values + every golden vector generated from the reference
(tests/golden/gen_golden.py).  Runs on CPU."""
import pytest
import torch

from oracle import ddp_oracle as O
from golden_util import load_aug_case  # noqa: E402
from golden_util import case_names, load_case, load_fcn_case, load_fpn_case, load_neck_case, load_post_case, max_rel

TOL = 2e-5   # oracle vs reference on the same CPU: fp32 summation-order noise only


def test_schedule_known_answers():
    # SURVEY.md §3.2 (values printed by the reference functions in the survey container)
    t = torch.tensor([1.0, 2 / 3, 1 / 3, 0.0])
    ls = O.alpha_cosine_log_snr(t)
    ref = torch.tensor([-18.90747452, -1.09885418, 1.09776640, 11.51292515])
    assert torch.allclose(ls, ref, rtol=0, atol=2e-6)
    a, s = O.log_snr_to_alpha_sigma(ls)
    assert torch.allclose(a, torch.tensor([7.8396e-05, 0.49995464, 0.86593378, 0.99999499]), atol=1e-6)
    assert torch.allclose(s, torch.tensor([1.0, 0.86605161, 0.50015861, 0.00316226]), atol=1e-6)
    lin = O.beta_linear_log_snr(t)
    assert torch.allclose(lin, torch.tensor([-10.00005436, -4.43273306, -0.71198654, 9.21028996]), atol=2e-6)


def test_time_pairs():
    # segmentors/ddp.py:204-213 with time_difference=1
    assert O.sampling_time_pairs(1) == [(1.0, 0.0)]
    p = O.sampling_time_pairs(3)
    assert [round(a, 6) for a, _ in p] == [1.0, round(2 / 3, 6), round(1 / 3, 6)]
    assert [round(b, 6) for _, b in p] == [round(1 / 3, 6), 0.0, 0.0]


def test_sine_posenc_known_answers():
    # SURVEY.md §8 a9
    pe = O.sine_positional_encoding(4, 6)
    assert torch.allclose(pe[:4, 0, 0], torch.tensor([0.70710665, 0.70710689, 0.62889153, 0.77749306]), atol=1e-6)
    assert torch.allclose(pe[128:132, 3, 5], torch.tensor([-0.5000006, 0.86602509, -0.96236897, 0.27174622]), atol=2e-6)


@pytest.mark.parametrize('name', case_names('seg'))
def test_seg_golden(name):
    cfg, sd, x, noise, step_noise, g = load_case(name)
    kw = dict(timesteps=cfg['timesteps'], randsteps=cfg['randsteps'], bit_scale=cfg['bit_scale'],
              sample_range0=cfg.get('sample_range', (0.0, 0.999))[0], noise_schedule=cfg['noise_schedule'],
              accumulation=cfg['accumulation'], time_difference=cfg.get('time_difference', 1))
    if cfg['diffusion'] == 'ddpm':
        out = O.ddpm_sample_seg(x, noise, step_noise, sd, **kw)
    else:
        trace = [] if cfg['trace'] else None
        out = O.ddim_sample_seg(x, noise, sd, trace=trace, **kw)
        if trace is not None:
            assert max_rel(trace[0]['feat'], g['feat_step0']) < TOL
            # reference layer outputs are seq-first (N, r, 256)
            assert max_rel(trace[0]['layers'][0].transpose(0, 1), g['layer0_step0']) < TOL
            assert max_rel(trace[0]['layers'][-1].transpose(0, 1), g['layer_last_step0']) < TOL
            for s in range(cfg['timesteps']):
                assert max_rel(trace[s]['logits'], g['logits_steps'][s]) < TOL
            for s in range(cfg['timesteps'] - 1):
                assert max_rel(trace[s]['mask_t'], g['mask_t_steps'][s]) < TOL
    assert out.shape == g['out'].shape
    assert max_rel(out, g['out']) < TOL


@pytest.mark.parametrize('name', case_names('seg_city_r2'))
def test_seg_golden_explicit_taps(name):
    """the explicit 4-corner pixel-unit gather (what the HIP kernel implements) == grid_sample."""
    cfg, sd, x, noise, _, g = load_case(name)
    out = O.ddim_sample_seg(x, noise, sd, timesteps=cfg['timesteps'], randsteps=cfg['randsteps'],
                            bit_scale=cfg['bit_scale'], accumulation=cfg['accumulation'], core='taps')
    assert max_rel(out, g['out']) < TOL


@pytest.mark.parametrize('name', case_names('depth'))
def test_depth_golden(name):
    cfg, sd, x, noise, _, g = load_case(name)
    trace = []
    out = O.sample_depth(x, noise, sd, timesteps=cfg['timesteps'], randsteps=cfg['randsteps'],
                         bit_scale=cfg['bit_scale'], min_depth=cfg['min_depth'], max_depth=cfg['max_depth'],
                         time_difference=cfg.get('time_difference', 1), trace=trace, scale_up=cfg.get('scale_up', False),
                         use_eps=cfg.get('use_eps', True))
    assert max_rel(trace[0]['feat'], g['feat_step0']) < TOL
    for s in range(cfg['timesteps']):
        assert max_rel(trace[s]['depth_pred'], g['depth_pred_steps'][s]) < 5 * TOL, s
    assert max_rel(out, g['out']) < 5 * TOL


@pytest.mark.parametrize('name', case_names('bev'))
def test_bev_golden(name):
    cfg, sd, x, noise, _, g = load_case(name)
    out = O.ddim_sample_bev(x, noise, sd, timesteps=cfg['timesteps'], randsteps=cfg['randsteps'],
                            bit_scale=cfg['bit_scale'], input_scope=cfg['input_scope'],
                            output_scope=cfg['output_scope'])
    assert out.shape == g['out'].shape
    assert max_rel(out, g['out']) < TOL


@pytest.mark.parametrize('name', case_names('post'))
def test_post_epilogue_golden(name):
    """resize -> crop -> resize -> softmax -> flip -> argmax, fixture made by the reference's own
    simple_test (encoder_decoder.py:229-304)."""
    cfg, scores, seg, margin = load_post_case(name)
    got = O.seg_postprocess(scores, cfg['img'], cfg['img_shape'], cfg['ori_shape'], cfg['align_corners'], cfg['flip'])[0]
    assert got.shape == seg.shape
    assert torch.equal(got.to(torch.uint8), seg)


@pytest.mark.parametrize('name', case_names('aug'))
def test_aug_test_golden(name):
    """multi-scale / flip aug_test (encoder_decoder.py:306-331), fixture made by the reference's own aug_test."""
    cfg, scores, metas, seg, prob, margin = load_aug_case(name)
    got, p = O.seg_aug_test(scores, metas, cfg['ori_shape'], cfg['align_corners'])
    assert torch.equal(got[0].to(torch.uint8), seg)
    assert max_rel(p[0], prob) < TOL


@pytest.mark.parametrize('name', case_names('slide'))
def test_slide_inference_golden(name):
    """sliding-window inference (encoder_decoder.py:180-227), fixtures made by the reference's own simple_test / inference with
    test_cfg.mode='slide' (backbone and sampler replaced by seeded per-window scores)."""
    import torch.nn.functional as F
    from golden_util import load_slide_case, reference_window_grid
    from ddp_amd.engine import slide_windows
    cfg, scores, seg, prob, margin = load_slide_case(name)
    ys, xs, crop = reference_window_grid(cfg)             # recorded from the reference's own slide_inference
    assert slide_windows(cfg['img'], cfg['crop_size'], cfg['stride']) == (ys, xs, crop)
    assert len(ys) * len(xs) == cfg['n_windows']
    preds = O.seg_slide_inference(scores, ys, xs, crop, cfg['img'], cfg['img_shape'], cfg['ori_shape'], cfg['align_corners'])
    p = F.softmax(preds, dim=1)
    if cfg['flip']:
        p = p.flip(dims=(3,) if cfg['flip'] == 'horizontal' else (2,))
    assert max_rel(p[0], prob) < TOL
    assert torch.equal(p.argmax(1)[0].to(torch.uint8), seg)


@pytest.mark.parametrize('name', case_names('dpost'))
def test_depth_epilogue_golden(name):
    """depth toolbox test entry (clamp / resize / flip-undo / mean over augmentations), fixtures made by calling the reference
    model the way depth/depth/apis/test.py:88 does: ``model(return_loss=False, img=[...], img_metas=[[...]])``."""
    from golden_util import load_dpost_case
    cfg, maps, flips, out = load_dpost_case(name)
    size = cfg['img'] if cfg['rescale'] else (cfg['augs'][0]['h'], cfg['augs'][0]['w'])
    got = O.depth_postprocess(maps, flips, size, cfg['min_depth'], cfg['max_depth'], cfg['align_corners'])
    assert got.shape == out.shape
    assert torch.equal(got, out)                              # same torch ops on the same CPU: bit-exact
    assert float(out.min()) >= cfg['min_depth'] and float(out.max()) <= cfg['max_depth']
    assert float((torch.cat([m.flatten() for m in maps]) > cfg['max_depth']).float().mean()) > 0.01      # the clamp is exercised
    assert float((torch.cat([m.flatten() for m in maps]) < cfg['min_depth']).float().mean()) > 0.01


@pytest.mark.parametrize('name', case_names('neck'))
def test_neck_golden(name):
    """MultiStageMerging, fixture made by the reference class."""
    cfg, levels, sd, out = load_neck_case(name)
    got = O.neck_multi_stage_merging(levels, sd, cfg['align_corners'])
    assert got.shape == out.shape
    assert max_rel(got, out) < TOL


@pytest.mark.parametrize('name', case_names('fcn'))
def test_fcn_head_golden(name):
    """FCNHeadWithTime.forward, fixture made by the reference class (eval mode)."""
    cfg, feat, temb, sd, out = load_fcn_case(name)
    got = O.fcn_head_forward(feat, temb, sd, cfg['num_convs'], cfg['dilation'])
    assert got.shape == out.shape
    assert max_rel(got, out) < TOL


@pytest.mark.parametrize('name', case_names('fpn'))
def test_fpn_golden(name):
    """FPN, fixture made by the reference class."""
    cfg, levels, sd, outs = load_fpn_case(name)
    got = O.neck_fpn(levels, sd)
    for g, o in zip(got, outs):
        assert g.shape == o.shape
        assert max_rel(g, o) < TOL


from golden_util import load_aligned_case, load_loopfcn_case  # noqa: E402


@pytest.mark.parametrize('name', case_names('aligned'))
def test_self_aligned_prepass_golden(name):
    """the self-aligned pre-pass, fixture recorded inside the reference's SelfAlignedDDP.forward_train
    (self_aligned_ddp.py:150-164)."""
    cfg, sd, x, noise, g = load_aligned_case(name)
    preds, logits = O.self_aligned_predict(x, noise, sd, cfg['bit_scale'], cfg['noise_schedule'])
    assert preds.shape == g['preds'].shape and logits.shape == g['logits'].shape
    assert max_rel(logits, g['logits']) < TOL
    assert max_rel(preds, g['preds']) < TOL


@pytest.mark.parametrize('name', case_names('loopfcn'))
def test_sampler_loop_around_fcn_head_golden(name):
    """the reference's ddim_sample / ddpm_sample driving the reference's FCNHeadWithTime."""
    cfg, sd, x, noise, step_noise, g = load_loopfcn_case(name)
    head = O.fcn_head_for_sampler(sd, cfg['num_convs'], cfg['dilation'])
    kw = dict(timesteps=cfg['timesteps'], randsteps=cfg['randsteps'], bit_scale=cfg['bit_scale'], accumulation=cfg['accumulation'],
              head=head)
    if cfg['diffusion'] == 'ddpm':
        out = O.ddpm_sample_seg(x, noise, step_noise, sd, **kw)
    else:
        out = O.ddim_sample_seg(x, noise, sd, **kw)
    assert out.shape == g['out'].shape
    assert max_rel(out, g['out']) < TOL


@pytest.mark.parametrize('name', ['full_c1', 'full_c2', 'full_c2_b5', 'full_c2_trained'])
def test_oracle_matches_reference_at_full_size(name):
    """The full-size fixtures (gen_golden.py --task fullsize: the reference at BASELINE.json's sizes) pin the oracle where the bench
    is quoted, not only on <= 33-px maps: C1 (1x512x512, K = 1) and one C2 image (512x1024, K = 3, 150 classes) - same per-step
    decisions, same final class map, scores to summation-order noise (bit-identical on the generating host).  C3 / C4 / C5 take
    CPU-minutes and are compared on the GPU box against the engine only (tests/test_full_size_parity.py)."""
    from golden_util import class_projection_weights, load_fullsize_case
    cfg, sd, x, noise, g = load_fullsize_case(name)
    b = cfg['b']
    dec = []
    old = torch.get_num_threads()
    torch.set_num_threads(min(8, old))
    try:
        out = O.ddim_sample_seg(x[b:b + 1], noise[b], sd, timesteps=cfg['timesteps'], randsteps=1, bit_scale=0.01,
                                accumulation=cfg['accumulation'], decisions=dec)
    finally:
        torch.set_num_threads(old)
    d = torch.stack(dec)[:, 0]
    differ = d != g['decisions']
    # a decision may only differ at a near-tie of the reference (stored top-2 gap, clipped at 1e-2 of the score scale)
    assert not (differ & (g['gap'].float() > 1e-4 * g['score_scale'].view(-1, 1, 1))).any()
    flat = out[0].reshape(cfg['num_classes'], -1)
    idx = torch.arange(0, flat.shape[1], cfg['stride'])
    if not differ.any():
        assert float((flat[:, idx] - g['out_sub']).abs().max()) <= TOL * float(g['out_absmax'])
        proj = (out[0] * class_projection_weights(cfg['num_classes']).view(-1, 1, 1)).sum(0)
        assert float((proj - g['out_proj']).abs().max()) <= TOL * float(g['out_proj'].abs().max())
        assert float((out[0].argmax(0) == g['final_cls'].long()).float().mean()) >= 0.9999
