timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
python - <<'PY'
import torch, time, ddp_amd
from ddp_amd.utils import synthetic
from oracle import ddp_oracle as O
sd = synthetic.make_fcn_state_dict(2, 150, True, True, 9)
head = ddp_amd.FCNHeadWithTime(num_convs=2, concat_input=True, in_channels=256, channels=256, num_classes=150, in_index=0, norm_cfg=dict(type='SyncBN'))
head.load_state_dict(sd); head = head.cuda().eval()
feat, temb = synthetic.make_fcn_inputs(8, 128, 256, 9)
f, t = feat.cuda(), temb.expand(8, 1024).cuda()
for _ in range(2): o = head([f], t)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(5): o = head([f], t)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print('FCNHeadWithTime 2 convs, 8x256x128x256 -> 150 classes: %.2f ms' % (dt*1e3))
import torch.nn.functional as F
sdg = {k: v.cuda() for k, v in sd.items()}
for _ in range(2): r = O.fcn_head_forward(f, temb.cuda(), sdg, 2)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(3): r = O.fcn_head_forward(f, temb.cuda(), sdg, 2)
torch.cuda.synchronize(); dt2=(time.perf_counter()-t0)/3
print('torch-ROCm eager same ops: %.2f ms' % (dt2*1e3), 'max rel', float((r-o).abs().max()/r.abs().max()))
PY
