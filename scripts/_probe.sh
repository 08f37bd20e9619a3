timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
python - <<'PY'
import torch, time, ddp_amd
from ddp_amd.utils import synthetic
from oracle import ddp_oracle as O
inc=[96,192,384,768]
sd = synthetic.make_fpn_state_dict(inc, 9)
fpn = ddp_amd.FPN(in_channels=inc, out_channels=256, act_cfg=None, norm_cfg=dict(type='GN', num_groups=32), num_outs=4)
fpn.load_state_dict(sd); fpn = fpn.cuda().eval()
lv = [t.cuda() for t in synthetic.make_backbone_levels(8, inc, 128, 256, 9)]
for _ in range(2): o = fpn(lv)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(5): o = fpn(lv)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print('FPN 8 x (128x256 .. 16x32): %.2f ms' % (dt*1e3))
sdg = {k: v.cuda() for k, v in sd.items()}
for _ in range(2): r = O.neck_fpn(lv, sdg)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(3): r = O.neck_fpn(lv, sdg)
torch.cuda.synchronize(); dt2=(time.perf_counter()-t0)/3
print('torch-ROCm eager same ops: %.2f ms' % (dt2*1e3), 'max rel', max(float((a-b).abs().max()/a.abs().max()) for a,b in zip(r,o)))
PY
