timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
python - <<'PY'
import torch, time
import ddp_amd
from ddp_amd.utils import synthetic
sd = synthetic.make_neck_state_dict(5)
neck = ddp_amd.MultiStageMerging([256]*4, 256, kernel_size=1, norm_cfg=dict(type='GN', num_groups=32), act_cfg=None)
neck.load_state_dict(sd); neck = neck.cuda().eval()
lv = [t.cuda() for t in synthetic.make_levels(8, 128, 256, 5)]
for _ in range(3): o = neck(lv)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): o = neck(lv)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
print('neck MSM 8 x (128x256 + 3 coarser levels): %.3f ms' % (dt*1e3))
import torch.nn.functional as F
w = sd['down.conv.weight'].cuda(); gw=sd['down.gn.weight'].cuda(); gb=sd['down.gn.bias'].cuda()
def ref():
    outs=[F.interpolate(t,size=(128,256),mode='bilinear',align_corners=False) for t in lv]
    return F.group_norm(F.conv2d(torch.cat(outs,1), w), 32, gw, gb)
for _ in range(2): r = ref()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): r = ref()
torch.cuda.synchronize(); dt2=(time.perf_counter()-t)/5
print('torch-ROCm eager same ops: %.3f ms' % (dt2*1e3), 'max rel', float((r-o[0]).abs().max()/r.abs().max()))
PY
