"""The registered drop-in classes (plugin surface) on the GPU: same configs / state_dict / methods as the
reference classes, outputs equal to the golden vectors recorded from the reference."""
import pytest
import torch

import ddp_amd
from golden_util import load_case, max_rel
from test_host_logic import ENCODER, POSENC, seg_cfg

pytestmark = pytest.mark.gpu
REL = 2e-4


class FakeBackbone(torch.nn.Module):
    """stands in for Swin+FPN+MultiStageMerging (frozen, upstream of the hot path): returns [x]."""

    def __init__(self, x):
        super().__init__()
        self.x = x

    def forward(self, img):
        return [self.x]


def _seg_model(cfg):
    model = ddp_amd.build_segmentor(seg_cfg(
        timesteps=cfg['timesteps'], randsteps=cfg['randsteps'], bit_scale=cfg['bit_scale'],
        accumulation=cfg['accumulation'], noise_schedule=cfg['noise_schedule'], diffusion=cfg['diffusion'],
        sample_range=tuple(cfg.get('sample_range', (0, 0.999))),
        decode_head=dict(seg_cfg()['decode_head'], num_classes=cfg['num_classes'])))
    return model


@pytest.mark.parametrize('name', ['seg_ade_k3', 'seg_city_r2', 'seg_linear'])
def test_segmentor_ddim_sample(name):
    cfg, sd, x, noise, _, g = load_case(name)
    model = _seg_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    out = model.ddim_sample(x.cuda(), None, noise=noise.unsqueeze(0).cuda())
    assert max_rel(out.cpu(), g['out']) < REL
    # encode_decode: extract_feat -> loop -> bilinear resize to the image size (segmentors/ddp.py:114-129)
    model.backbone = FakeBackbone(x.cuda())
    img = torch.zeros(1, 3, cfg['h'] * 4, cfg['w'] * 4, device='cuda')
    torch.manual_seed(0)
    o1 = model.encode_decode(img, None)
    torch.manual_seed(0)
    o2 = model.encode_decode(img, None)
    assert o1.shape == (1, cfg['num_classes'], cfg['h'] * 4, cfg['w'] * 4)
    assert torch.equal(o1, o2)                       # same seed -> bit-identical, like the reference
    torch.manual_seed(0)
    labels = model.simple_test(img, None)
    assert labels[0].shape == (cfg['h'] * 4, cfg['w'] * 4)
    # fused epilogue == argmax of the resized scores (softmax is monotone), up to interpolation rounding at ties
    assert (torch.from_numpy(labels[0]) != o1[0].argmax(0).cpu()).float().mean() < 1e-3
    # rescale path: crop to img_shape, resize to ori_shape, flip (encoder_decoder.py:236-248,278-285)
    H, W = cfg['h'] * 4, cfg['w'] * 4
    meta = [dict(img_shape=(H - 3, W - 5, 3), ori_shape=(H + 7, W - 9, 3), flip=True, flip_direction='horizontal')]
    torch.manual_seed(0)
    lab2 = model.simple_test(img, meta, rescale=True)[0]
    torch.manual_seed(0)
    ref = torch.nn.functional.softmax(model.whole_inference(img, meta, True), dim=1).flip(dims=(3,)).argmax(1)[0].cpu()
    assert lab2.shape == (H + 7, W - 9)
    assert (torch.from_numpy(lab2) != ref).float().mean() < 1e-3


def test_segmentor_ddpm_sample():
    cfg, sd, x, noise, step_noise, g = load_case('seg_ddpm')
    model = _seg_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    out = model.ddpm_sample(x.cuda(), None, noise=noise.unsqueeze(0).cuda(), step_noise=step_noise.unsqueeze(1).cuda())
    assert max_rel(out.cpu(), g['out']) < REL


def test_decode_head_forward_surface():
    """DeformableHeadWithTime.forward(inputs, times) called the way DDP._decode_head_forward_test does."""
    from oracle import ddp_oracle as O
    cfg, sd, x, noise, _, g = load_case('seg_city_r2')
    model = _seg_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    temb = O.time_mlp(O.alpha_cosine_log_snr(torch.tensor([1.0])), sd).cuda()
    logits = model._decode_head_forward_test([g['feat_step0'].cuda()], temb, img_metas=None)
    assert logits.shape == g['logits_steps'][0].shape
    assert max_rel(logits.cpu(), g['logits_steps'][0]) < REL
    # changing a parameter in place invalidates the cached engine
    with torch.no_grad():
        model.decode_head.conv_seg.bias.add_(1.0)
    logits2 = model._decode_head_forward_test([g['feat_step0'].cuda()], temb, img_metas=None)
    assert max_rel(logits2.cpu(), g['logits_steps'][0] + 1.0) < REL


def test_depther_sample():
    cfg, sd, x, noise, _, g = load_case('depth_k3_r2')
    dcfg = dict(type='DDP', bit_scale=cfg['bit_scale'], timesteps=cfg['timesteps'], randsteps=cfg['randsteps'],
                min_depth=cfg['min_depth'], max_depth=cfg['max_depth'],
                decode_head=dict(type='DeformableHeadWithTime', in_channels=[256], channels=256, in_index=[0],
                                 dropout_ratio=0., scale_up=False, min_depth=1e-3, max_depth=80, use_eps=True,
                                 align_corners=False, num_feature_levels=1, encoder=ENCODER, positional_encoding=POSENC))
    model = ddp_amd.build_depther(dcfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    out = model.sample(x.cuda(), None, noise=noise.unsqueeze(0).cuda())
    assert max_rel(out.cpu(), g['out']) < REL
    model.backbone = FakeBackbone(x.cuda())
    o = model.encode_decode(torch.zeros(1, 3, cfg['h'] * 4, cfg['w'] * 4, device='cuda'), None, rescale=True)
    assert o.shape == (1, 1, cfg['h'] * 4, cfg['w'] * 4) and float(o.min()) >= 1e-3 and float(o.max()) <= 80


def _depth_model(cfg, sd, align_corners=False, min_depth=1e-3, max_depth=80):
    dcfg = dict(type='DDP', bit_scale=cfg.get('bit_scale', 0.1), timesteps=cfg.get('timesteps', 3), randsteps=cfg.get('randsteps', 1),
                min_depth=min_depth, max_depth=max_depth, test_cfg=dict(mode='whole'),
                decode_head=dict(type='DeformableHeadWithTime', in_channels=[256], channels=256, in_index=[0],
                                 dropout_ratio=0., scale_up=False, min_depth=min_depth, max_depth=max_depth, use_eps=True,
                                 align_corners=align_corners, num_feature_levels=1, encoder=ENCODER, positional_encoding=POSENC))
    model = ddp_amd.build_depther(dcfg)
    model.load_state_dict(sd, strict=True)
    return model.cuda().eval()


@pytest.mark.parametrize('name', ['dpost_kitti_flip', 'dpost_simple_hflip', 'dpost_vflip_ac3', 'dpost_norescale'])
def test_depther_test_entry_matches_reference_fixture(name):
    """The depth toolbox's harness call ``model(return_loss=False, **data)`` (depth/depth/apis/test.py:88,204) on the drop-in
    class, end to end through forward -> forward_test -> simple_test / aug_test -> the fused epilogue, against what the
    REFERENCE model returned for the same call (fixtures of gen_golden.py --task dpost; backbone and sampler replaced by the
    same seeded low-resolution maps on both sides)."""
    from golden_util import load_dpost_case
    from ddp_amd.utils import synthetic
    cfg, maps, flips, out = load_dpost_case(name)
    model = _depth_model({}, synthetic.make_state_dict('depth', 1, 6, 256, seed=0), cfg['align_corners'], cfg['min_depth'], cfg['max_depth'])
    model.extract_feat = lambda img: [None]
    calls = []

    def sample(x, img_metas=None, noise=None):
        calls.append(1)
        return maps[(len(calls) - 1) % len(maps)].cuda()
    model.sample = sample
    H, W = cfg['img']
    data = dict(img=[torch.zeros(cfg['batch'], 3, H, W, device='cuda') for _ in flips],
                img_metas=[[dict(img_shape=(H, W, 3), ori_shape=(H, W, 3), pad_shape=(H, W, 3), flip=f is not None,
                                 flip_direction=f or 'horizontal')] * cfg['batch'] for f in flips])
    res = model(return_loss=False, **data) if cfg['rescale'] else model(return_loss=False, rescale=False, **data)
    assert isinstance(res, list) and len(res) == cfg['batch'] and len(calls) == len(flips)
    assert res[0].dtype == out.numpy().dtype and res[0].shape == tuple(out.shape[1:])
    err = float((torch.from_numpy(res[0]) - out[0]).abs().max())
    assert err <= 2e-5 * cfg['max_depth'], err


def test_depther_test_entry_end_to_end_vs_oracle():
    """Same call, nothing stubbed but the backbone: KITTI's two augmentations (plain + horizontally flipped feature), each
    through the full K-step loop with its own noise, then ONE epilogue kernel - against the oracle's loop + the reference's op
    sequence for the epilogue.  Also: the protocol checks of base.py:62-92, a b = 2 batch, and simple_test == one augmentation."""
    from oracle import ddp_oracle as O
    cfg, sd, x, _, _, _ = load_case('depth_k3_r2')
    model = _depth_model(cfg, sd)
    h, w, r = cfg['h'], cfg['w'], cfg['randsteps']
    H, W = 4 * h, 4 * w
    xf = x.flip(dims=(3,))
    feats = iter([x.cuda(), xf.cuda()])
    model.extract_feat = lambda img: [next(feats)]
    meta = dict(img_shape=(H, W, 3), ori_shape=(H, W, 3), pad_shape=(H, W, 3), flip=False, flip_direction='horizontal')
    data = dict(img=[torch.zeros(1, 3, H, W, device='cuda')] * 2, img_metas=[[meta], [dict(meta, flip=True)]])
    torch.manual_seed(5)
    res = model(return_loss=False, **data)
    torch.manual_seed(5)
    n0 = torch.randn((1, r, 1, h, w), device='cuda').cpu()
    n1 = torch.randn((1, r, 1, h, w), device='cuda').cpu()
    kw = dict(timesteps=cfg['timesteps'], randsteps=r, bit_scale=cfg['bit_scale'], min_depth=cfg['min_depth'], max_depth=cfg['max_depth'])
    ref = O.depth_postprocess([O.sample_depth(x, n0[0], sd, **kw), O.sample_depth(xf, n1[0], sd, **kw)], [None, 'horizontal'],
                              (H, W), 1e-3, 80.0)
    assert len(res) == 1 and res[0].shape == (1, H, W)
    assert max_rel(torch.from_numpy(res[0]), ref[0]) < REL
    # one augmentation == simple_test == inference; b = 2 images in one call
    feats = iter([torch.cat([x, xf]).cuda()])
    torch.manual_seed(6)
    two = model(return_loss=False, img=[torch.zeros(2, 3, H, W, device='cuda')], img_metas=[[meta, meta]])
    assert len(two) == 2 and two[0].shape == (1, H, W)
    torch.manual_seed(6)
    nb = torch.randn((2, r, 1, h, w), device='cuda').cpu()
    for i, xi in enumerate((x, xf)):
        refi = O.depth_postprocess([O.sample_depth(xi, nb[i], sd, **kw)], [None], (H, W), 1e-3, 80.0)
        assert max_rel(torch.from_numpy(two[i]), refi[0]) < REL
    # protocol errors of base.py:62-92 / encoder_decoder.py:183-187
    with pytest.raises(TypeError):
        model(return_loss=False, img=torch.zeros(1, 3, H, W, device='cuda'), img_metas=[[meta]])
    with pytest.raises(ValueError):
        model(return_loss=False, img=[torch.zeros(1, 3, H, W, device='cuda')] * 2, img_metas=[[meta]])
    with pytest.raises(NotImplementedError):
        model(return_loss=True, img=torch.zeros(1, 3, H, W, device='cuda'), img_metas=[meta])
    model.test_cfg = dict(mode='slide')
    with pytest.raises(NotImplementedError):
        model(return_loss=False, img=[torch.zeros(1, 3, H, W, device='cuda')], img_metas=[[meta]])


def test_bev_ddim_sample():
    cfg, sd, x, noise, _, g = load_case('bev_fusion')
    head = ddp_amd.BEVDeformableHeadWithTime(
        num_feature_levels=1, encoder=dict(ENCODER, num_layers=cfg['num_layers']), positional_encoding=POSENC,
        classes=list('abcdef'), loss='focal',
        grid_transform=dict(input_scope=cfg['input_scope'], output_scope=cfg['output_scope']))
    model = ddp_amd.BEVDDP(bit_scale=cfg['bit_scale'], timesteps=cfg['timesteps'], randsteps=cfg['randsteps'],
                           feat_channels=cfg['feat_channels'])
    model.load_state_dict({k: v for k, v in sd.items() if not k.startswith('decode_head.')}, strict=True)
    head.load_state_dict({k[len('decode_head.'):]: v for k, v in sd.items() if k.startswith('decode_head.')}, strict=True)
    model, head = model.cuda().eval(), head.cuda().eval()
    out = model.ddim_sample([x.cuda()], head, noise=noise.unsqueeze(0).cuda())
    assert out.shape == g['out'].shape
    assert max_rel(out.cpu(), g['out']) < REL


def test_segmentor_batched_harness():
    """§8 f4: b > 1 images per call, every image with its own crop / rescale / flip; the mmseg call protocol
    ``model([img], [[meta, ...]], return_loss=False, rescale=True)`` through ddp_amd.apis.single_gpu_test."""
    import numpy as np
    from ddp_amd import apis
    cfg, sd, x, _, _, _ = load_case('seg_ade_k3')
    model = _seg_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    H, W = cfg['h'] * 4, cfg['w'] * 4
    xb = torch.cat([x, x.flip(dims=(3,))]).cuda()
    model.backbone = FakeBackbone(xb)
    img = torch.zeros(2, 3, H, W, device='cuda')
    m0 = dict(img_shape=(H, W, 3), ori_shape=(H + 5, W + 3, 3), flip=False)
    m1 = dict(img_shape=(H - 2, W - 4, 3), ori_shape=(H - 7, W + 9, 3), flip=True, flip_direction='horizontal')

    def run(metas):
        torch.manual_seed(3)
        return model.simple_test(img, metas, rescale=True)

    mixed, a, b = run([m0, m1]), run([m0, m0]), run([m1, m1])
    assert mixed[0].shape == (H + 5, W + 3) and mixed[1].shape == (H - 7, W + 9)
    assert np.array_equal(mixed[0], a[0]) and np.array_equal(mixed[1], b[1])
    # ... and every image of the mixed call against the ORACLE (the harness test was self-consistency only): the
    # reference's sampler (b = 1, its own noise: the draw the batched call made for that image) + the reference's op sequence of
    # simple_test for that image's meta (crop to img_shape, resize to ori_shape, softmax, flip undone, argmax - encoder_decoder.py:236-296)
    from oracle import ddp_oracle as O
    torch.manual_seed(3)
    nb = torch.randn((2, cfg['randsteps'], 256, cfg['h'], cfg['w']), device='cuda').cpu()
    for i, (xi, meta) in enumerate(((x, m0), (x.flip(dims=(3,)), m1))):
        sc = O.ddim_sample_seg(xi, nb[i], sd, timesteps=cfg['timesteps'], randsteps=cfg['randsteps'], bit_scale=cfg['bit_scale'],
                               accumulation=cfg['accumulation'])
        ref = O.seg_postprocess(sc, (H, W), meta['img_shape'][:2], meta['ori_shape'][:2], False,
                                meta.get('flip_direction') if meta['flip'] else None)[0]
        agree = float((torch.from_numpy(mixed[i].astype('int64')) == ref).float().mean())
        print(f'batched harness, image {i}: class map agreement with the oracle {agree:.5f}')
        assert agree >= 0.999
    torch.manual_seed(3)
    fwd = model([img], [[m0, m1]], return_loss=False, rescale=True)
    assert all(np.array_equal(p, q) for p, q in zip(fwd, mixed))
    with pytest.raises(ValueError):
        model.simple_test(img, [m0, m1, m0])
    # aug_test (encoder_decoder.py:306-331): two augmentations of the batch, the second one flipped
    mf = dict(m0, flip=True, flip_direction='horizontal')
    aug = model([img, img], [[m0, m0], [mf, mf]], return_loss=False)
    assert len(aug) == 2 and all(a.shape == (H + 5, W + 3) for a in aug)
    assert all(0 <= int(a.min()) and int(a.max()) < cfg['num_classes'] for a in aug)
    with pytest.raises(ValueError):
        model([img, img], [[m0, m0]], return_loss=False)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            return i

        def pre_eval(self, preds, indices):
            return [(int(i), p.shape) for p, i in zip(preds, indices)]

    def collate(idx):
        return dict(img=[torch.zeros(len(idx), 3, H, W)], img_metas=[[m0 if i % 2 == 0 else m1 for i in idx]])

    loader = torch.utils.data.DataLoader(DS(), batch_size=2, shuffle=False, collate_fn=collate)
    res = apis.single_gpu_test(model, loader, pre_eval=True)
    assert res == [(0, (H + 5, W + 3)), (1, (H - 7, W + 9)), (2, (H + 5, W + 3)), (3, (H - 7, W + 9))]


def test_segmentor_size_stream_b1():
    """The reference's own test protocol (segmentation/tools/test.py:214-219, mmseg/apis/test.py:87-89): ONE image per call
    and a different (h, w) almost every call.  Five sizes, alternating, each call checked against the oracle; the engine and
    the packed weights are built once (a new size only re-derives the positional tables: ``DDPEngine.set_geometry``), and
    coming back to a size reproduces the earlier bits."""
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    cfg, sd, _, _, _, _ = load_case('seg_ade_k3')
    model = _seg_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    sizes = [(16, 24), (13, 31), (32, 20), (7, 9), (24, 43)]
    first = {}
    engines = set()
    for rnd in range(2):
        for i, (h, w) in enumerate(sizes if rnd == 0 else sizes[::-1]):
            x, noise = synthetic.make_inputs(1, h, w, cfg['randsteps'], 256, 256, seed=900 + 7 * h + w)
            out = model.ddim_sample(x.cuda(), noise=noise.cuda()).cpu()
            engines.add(id(next(reversed(model._engine_cache.values()))))
            if rnd == 0:
                ref = O.ddim_sample_seg(x, noise[0], sd, timesteps=cfg['timesteps'], randsteps=cfg['randsteps'],
                                        bit_scale=cfg['bit_scale'], accumulation=cfg['accumulation'])
                assert max_rel(out, ref) < REL, (h, w)
                first[(h, w)] = out
            else:
                assert torch.equal(out, first[(h, w)]), (h, w)
    assert len(engines) == 1                                   # one engine served all ten calls
    eng = next(iter(model._engine_cache.values()))
    assert eng.geometry_changes == 8                           # 4 new sizes going up, 4 coming back (the turn repeats a size)
    assert len(model._weights_cache) == 1
    # a batch of 3 at one of the sizes through the same engine == three single-image calls
    h, w = sizes[1]
    xs, ns = zip(*[synthetic.make_inputs(1, h, w, cfg['randsteps'], 256, 256, seed=50 + j) for j in range(3)])
    outb = model.ddim_sample(torch.cat(xs).cuda(), noise=torch.cat(ns).cuda()).cpu()
    for j in range(3):
        assert torch.equal(model.ddim_sample(xs[j].cuda(), noise=ns[j].cuda()).cpu(), outb[j:j + 1])
    # changing a weight invalidates engines and packed weights
    with torch.no_grad():
        model.decode_head.conv_seg.bias.add_(1.0)
    out2 = model.ddim_sample(xs[0].cuda(), noise=ns[0].cuda()).cpu()
    assert not torch.equal(out2, outb[:1])


@pytest.mark.parametrize('case,sampler', [('seg_ade_k3', 'ddim'), ('seg_city_r2', 'ddim'), ('seg_ddpm', 'ddpm'), ('depth_k3_r2', 'ddim'),
                                          ('depth_td2', 'ddim'), ('bev_fusion', 'ddim'), ('bev_camera', 'ddim')])   # round 6: the depth chain (r = 1), the bev u chain
def test_sample_as_hip_graph_is_bit_identical(case, sampler):
    """the K-step loop captured in a hipGraph (no host sync, no host memory inside ``ddp_sample``) and
    replayed - same bits as the plain call, on the captured inputs and on NEW inputs copied into the graph's static buffers;
    a geometry change invalidates the graph loudly."""
    from ddp_amd import _lib
    from ddp_amd.engine import DDPEngine
    from ddp_amd.utils import synthetic
    cfg, sd, x, noise, step_noise, g = load_case(case)
    task = cfg['task']
    kw = dict(h=cfg['h'], w=cfg['w'], batch=1, randsteps=cfg['randsteps'], timesteps=cfg['timesteps'], bit_scale=cfg['bit_scale'])
    if task == 'seg':
        kw.update(num_classes=cfg['num_classes'], accumulation=cfg['accumulation'], sampler=sampler)
    elif task == 'depth':
        kw.update(min_depth=cfg['min_depth'], max_depth=cfg['max_depth'], time_difference=cfg.get('time_difference', 1))
    else:
        kw.update(num_classes=6, feat_channels=cfg['feat_channels'], bev_input_scope=cfg['input_scope'], bev_output_scope=cfg['output_scope'])
    eng = DDPEngine(sd, task, **kw)
    dx, dn = x.cuda(), noise.unsqueeze(0).cuda()
    dsn = step_noise.unsqueeze(1).contiguous().cuda() if step_noise is not None else None
    plain = eng.sample(dx, dn, dsn).clone()
    assert max_rel(plain.cpu(), g['out']) < REL
    graph = eng.capture(dx, dn, dsn)
    assert torch.equal(graph.replay().clone(), plain)
    assert torch.equal(graph.replay().clone(), plain)                  # replays are idempotent
    cm = 1 if task == 'depth' else 256
    x2, n2 = synthetic.make_inputs(1, cfg['h'], cfg['w'], cfg['randsteps'], cfg.get('feat_channels', 256), cm, seed=4242)
    sn2 = torch.randn_like(dsn) if dsn is not None else None
    want = eng.sample(x2.cuda(), n2.cuda(), sn2).clone()
    assert not torch.equal(want, plain)
    assert torch.equal(graph.replay(x2.cuda(), n2.cuda(), sn2).clone(), want)
    assert torch.equal(graph.replay(dx, dn, dsn).clone(), plain)
    # head_forward rewrites the FiLM slot of step 0 (clears the prepared state): replay restores the constants first
    if task == 'seg':
        R = cfg['randsteps']
        eng.head_forward(torch.randn(R, 256, cfg['h'], cfg['w'], device='cuda'), torch.randn(1, 1024, device='cuda'))
        assert torch.equal(graph.replay(dx, dn, dsn).clone(), plain)
    ws_ptr = eng.workspace.data_ptr()
    eng.set_geometry(1, cfg['h'] - 1, cfg['w'])          # a smaller map fits the same workspace
    with pytest.raises(_lib.DdpError, match='captured for geometry'):
        graph.replay()
    # back on the captured geometry with the SAME workspace: valid again
    eng.set_geometry(1, cfg['h'], cfg['w'])
    assert eng.workspace.data_ptr() == ws_ptr
    assert torch.equal(graph.replay(dx, dn, dsn).clone(), plain)
    # growing the geometry reallocates the workspace; back on the captured size the graph still holds the OLD address:
    # refused loudly (it used to read and write freed memory)
    eng.set_geometry(4, 4 * cfg['h'], 4 * cfg['w'])
    eng.set_geometry(1, cfg['h'], cfg['w'])
    with pytest.raises(_lib.DdpError, match='stale graph'):
        graph.replay()
    assert torch.equal(eng.sample(dx, dn, dsn), plain)


def test_segmentor_slide_inference_matches_the_reference_composition():
    """test_cfg.mode='slide' on the drop-in segmentor (encoder_decoder.py:180-227): the windows go through the K-step loop as
    ONE batch and one fused epilogue; compared with the reference's formulation composed in torch from this class's own
    per-window ``encode_decode`` with the same noise (the batched call draws the windows' noise as one tensor)."""
    import numpy as np
    cfg, sd, x, _, _, _ = load_case('seg_city_r2')
    model = _seg_model(dict(cfg, randsteps=1))
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    model.test_cfg = dict(mode='slide', crop_size=(32, 48), stride=(20, 30))
    H, W = 56, 100
    ys, xs = [0, 20, 24], [0, 30, 52]

    class CropBackbone(torch.nn.Module):               # the "feature" of a crop = a pooled view of the crop itself: depends on its content
        def forward(self, img):
            return [torch.nn.functional.avg_pool2d(img, 4).repeat(1, 86, 1, 1)[:, :256].contiguous()]
    model.backbone = CropBackbone()
    g = torch.Generator().manual_seed(3)
    img = torch.randn(1, 3, H, W, generator=g).cuda()
    meta = [dict(img_shape=(H - 2, W - 3, 3), ori_shape=(H + 5, W + 9, 3), flip=True, flip_direction='horizontal')]
    torch.manual_seed(21)
    seg = model.simple_test(img, meta, rescale=True)[0]
    torch.manual_seed(21)
    preds = model.slide_inference(img, meta, True)
    assert seg.shape == (H + 5, W + 9) and tuple(preds.shape) == (1, cfg['num_classes'], H + 5, W + 9)
    # reference composition: the same noise, window by window
    torch.manual_seed(21)
    noise = torch.randn((9, 1, 256, 8, 12), device='cuda')
    acc = torch.zeros(1, cfg['num_classes'], H, W, device='cuda')
    cnt = torch.zeros(1, 1, H, W, device='cuda')
    i = 0
    for y1 in ys:
        for x1 in xs:
            crop = img[:, :, y1:y1 + 32, x1:x1 + 48]
            low = model.ddim_sample(model.extract_feat(crop)[0], noise=noise[i:i + 1])
            acc[:, :, y1:y1 + 32, x1:x1 + 48] += torch.nn.functional.interpolate(low, size=(32, 48), mode='bilinear', align_corners=False)
            cnt[:, :, y1:y1 + 32, x1:x1 + 48] += 1
            i += 1
    want = torch.nn.functional.interpolate((acc / cnt)[:, :, :H - 2, :W - 3], size=(H + 5, W + 9), mode='bilinear', align_corners=False)
    assert max_rel(preds.cpu(), want.cpu()) < 1e-5
    ref = torch.softmax(want, dim=1).flip(dims=(3,)).argmax(1)[0].cpu().numpy()
    assert (seg != ref).mean() < 2e-3
    # the harness call and inference() in slide mode; aug_test composes inference()
    torch.manual_seed(21)
    assert np.array_equal(model([img], [meta], return_loss=False)[0], seg)
    torch.manual_seed(21)
    prob = model.inference(img, meta, True)
    assert torch.allclose(prob.sum(1), torch.ones_like(prob[:, 0]), atol=1e-5)
    assert (prob.argmax(1)[0].cpu().numpy() != seg).mean() < 1e-3
    out = model([img, img], [meta, [dict(meta[0], flip=False)]], return_loss=False)
    assert out[0].shape == (H + 5, W + 9)


def _seg_model_for_epilogue(num_classes, align_corners, test_cfg=None):
    from ddp_amd.utils import synthetic
    cfg = seg_cfg(decode_head=dict(seg_cfg()['decode_head'], num_classes=num_classes, align_corners=align_corners), test_cfg=test_cfg)
    model = ddp_amd.build_segmentor(cfg)
    model.load_state_dict(synthetic.make_state_dict('seg', num_classes, 6, 256, seed=0), strict=True)
    model = model.cuda().eval()
    model.extract_feat = lambda img: [torch.zeros(img.shape[0], 256, 1, 1, device='cuda')]
    return model


def _same_class_map(got, seg, margin):
    diff = torch.from_numpy(got.astype('int64')) != seg.long()
    return float(diff.float().mean()) < 1e-3 and not bool((diff & (margin > 1e-5)).any())


def test_segmentor_harness_call_matches_reference_fixtures():
    """The segmentation harness call ``model(return_loss=False, **data)`` (segmentation/mmseg/apis/test.py:87-89) on the drop-in class
    against what the REFERENCE model returned for it, for the three post-loop protocols: single scale (`post_*`, simple_test),
    multi-scale / flip (`aug_*`, aug_test) and sliding window (`slide_*`, test_cfg.mode='slide') - backbone and sampler replaced
    by the same seeded low-resolution scores on both sides (the plugin's epilogue path compared with the
    reference, not with its own torch composition)."""
    from golden_util import case_names, load_aug_case, load_post_case, load_slide_case
    for name in case_names('post'):
        cfg, scores, seg, margin = load_post_case(name)
        model = _seg_model_for_epilogue(cfg['num_classes'], cfg['align_corners'], dict(mode='whole'))
        model.ddim_sample = lambda x, img_metas=None, _s=scores: _s.cuda()
        meta = dict(img_shape=tuple(cfg['img_shape']) + (3,), ori_shape=tuple(cfg['ori_shape']) + (3,), pad_shape=tuple(cfg['img']) + (3,),
                    flip=cfg['flip'] is not None, flip_direction=cfg['flip'] or 'horizontal')
        res = model(return_loss=False, img=[torch.zeros((1, 3) + tuple(cfg['img']), device='cuda')], img_metas=[[meta]])
        assert len(res) == 1 and res[0].shape == tuple(cfg['ori_shape']) and _same_class_map(res[0], seg, margin), name
    for name in case_names('aug'):
        cfg, scores, metas, seg, prob, margin = load_aug_case(name)
        model = _seg_model_for_epilogue(cfg['num_classes'], cfg['align_corners'], dict(mode='whole'))
        calls = []

        def sample(x, img_metas=None, _c=calls, _s=scores):
            _c.append(1)
            return _s[(len(_c) - 1) % len(_s)].cuda()
        model.ddim_sample = sample
        imgs = [torch.zeros((1, 3) + tuple(a['img']), device='cuda') for a in cfg['augs']]
        mm = [[dict(img_shape=tuple(a['img_shape']) + (3,), ori_shape=tuple(cfg['ori_shape']) + (3,), pad_shape=tuple(a['img']) + (3,),
                    flip=a['flip'] is not None, flip_direction=a['flip'] or 'horizontal')] for a in cfg['augs']]
        res = model(return_loss=False, img=imgs, img_metas=mm)
        assert len(calls) == len(imgs) and res[0].shape == tuple(cfg['ori_shape']) and _same_class_map(res[0], seg, margin), name
    for name in case_names('slide'):
        cfg, scores, seg, prob, margin = load_slide_case(name)
        model = _seg_model_for_epilogue(cfg['num_classes'], cfg['align_corners'],
                                        dict(mode='slide', crop_size=tuple(cfg['crop_size']), stride=tuple(cfg['stride'])))
        model.extract_feat = lambda img: [torch.zeros(img.shape[0], 256, cfg['h'], cfg['w'], device='cuda')]
        model.ddim_sample = lambda x, img_metas=None, _s=scores: torch.cat(_s).cuda()       # the windows arrive as ONE batch
        meta = dict(img_shape=tuple(cfg['img_shape']) + (3,), ori_shape=tuple(cfg['ori_shape']) + (3,), pad_shape=tuple(cfg['img']) + (3,),
                    flip=cfg['flip'] is not None, flip_direction=cfg['flip'] or 'horizontal')
        res = model(return_loss=False, img=[torch.zeros((1, 3) + tuple(cfg['img']), device='cuda')], img_metas=[[meta]])
        assert res[0].shape == tuple(cfg['ori_shape']) and _same_class_map(res[0], seg, margin), name
        p = model.inference(torch.zeros((1, 3) + tuple(cfg['img']), device='cuda'), [meta], True)
        assert max_rel(p[0].cpu(), prob) < 2e-6, name


def test_engine_set_geometry_is_transactional():
    """a geometry switch that fails (here: more tokens than the library accepts) must leave the engine on its old,
    still prepared geometry - the same bad request fails again instead of hitting the early-return, and the next good call
    reproduces the earlier bits."""
    from ddp_amd import _lib
    from ddp_amd.engine import DDPEngine
    cfg, sd, x, noise, _, g = load_case('seg_ade_k3')
    eng = DDPEngine(sd, 'seg', h=cfg['h'], w=cfg['w'], batch=1, randsteps=cfg['randsteps'], timesteps=cfg['timesteps'],
                    num_classes=cfg['num_classes'], bit_scale=cfg['bit_scale'], accumulation=cfg['accumulation'])
    first = eng.sample(x.cuda(), noise.unsqueeze(0).cuda()).clone()
    assert max_rel(first.cpu(), g['out']) < REL
    ws = eng.workspace
    for _ in range(2):
        with pytest.raises(_lib.DdpError):
            eng.set_geometry(1, 1 << 15, 1 << 15)           # 2^30 tokens: refused by ddp_query_workspace
        assert eng.geometry() == (1, cfg['h'], cfg['w']) and eng.workspace is ws and eng.geometry_changes == 0
    assert torch.equal(eng.sample(x.cuda(), noise.unsqueeze(0).cuda()), first)
    # a constructor head_hw equal to the map follows it; the switch itself still works after the failures
    eng.set_geometry(1, 9, 11)
    assert (eng.cfg.head_h, eng.cfg.head_w) == (9, 11) and eng.geometry_changes == 1
    eng.set_geometry(1, cfg['h'], cfg['w'])
    assert torch.equal(eng.sample(x.cuda(), noise.unsqueeze(0).cuda()), first)


def test_aug_epilogue_small_then_large_class_count_in_one_process():
    """k_seg_aug_postprocess's dynamic-LDS limit is set once per device; it must cover a later, larger class count
    (19 classes = 9.7 KB first, then 150 = 76.8 KB and 256 = 128 KB, above the 64 KB default)."""
    from ddp_amd.engine import seg_aug_postprocess
    from ddp_amd.utils import synthetic
    from oracle import ddp_oracle as O
    for K in (19, 150, 256):
        sc = [synthetic.make_scores(1, K, 6, 9, 300 + K), synthetic.make_scores(1, K, 6, 9, 301 + K)]
        metas = [dict(img_size=(24, 36), crop_size=(24, 36), flip=None), dict(img_size=(24, 36), crop_size=(24, 36), flip='horizontal')]
        seg, p = seg_aug_postprocess([t.cuda() for t in sc], metas, (24, 36), False, return_prob=True)
        ref, rp = O.seg_aug_test(sc, metas, (24, 36), False)
        assert max_rel(p.cpu(), rp) < 2e-6, K
        assert (seg.cpu().long() != ref).float().mean() < 1e-2, K


def test_segmentor_aug_test_matches_inference_mean():
    """``aug_test`` (fused multi-scale / flip epilogue) against the reference's formulation composed from this class's own
    ``inference`` (torch ops, encoder_decoder.py:251-287): same noise, class maps equal up to near-ties."""
    import numpy as np
    cfg, sd, x, _, _, _ = load_case('seg_ade_k3')
    model = _seg_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    H, W = cfg['h'] * 4, cfg['w'] * 4
    model.backbone = FakeBackbone(x.cuda())
    img = torch.zeros(1, 3, H, W, device='cuda')
    ori = (H + 9, W - 5, 3)
    metas = [[dict(img_shape=(H, W, 3), ori_shape=ori, flip=False)],
             [dict(img_shape=(H - 3, W - 2, 3), ori_shape=ori, flip=True, flip_direction='horizontal')],
             [dict(img_shape=(H, W - 1, 3), ori_shape=ori, flip=True, flip_direction='vertical')]]
    torch.manual_seed(11)
    got = model.aug_test([img] * 3, metas)[0]
    torch.manual_seed(11)
    prob = sum(model.inference(img, m, True) for m in metas) / 3
    ref = prob.argmax(1)[0].cpu().numpy()
    top2 = prob.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])[0].cpu().numpy()
    diff = got != ref
    assert got.shape == ori[:2] and diff.mean() < 1e-3 and not (diff & (margin > 1e-5)).any()
    assert np.array_equal(model([img] * 3, metas, return_loss=False)[0].shape, ori[:2])
