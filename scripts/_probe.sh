timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python - <<'PY'
import torch, time
from ddp_amd.engine import seg_postprocess
from ddp_amd.utils import synthetic
s = synthetic.make_scores(8, 150, 128, 256, seed=1).cuda()
for _ in range(3): o = seg_postprocess(s, (512, 1024))
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): o = seg_postprocess(s, (512, 1024))
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
print('postprocess 8x150x128x256 -> 8x512x1024: %.3f ms' % (dt*1e3), 'read GB/s', s.numel()*4/dt/1e9)
meta_crop=(509,1019); ori=(683,1024)
for _ in range(3): o = seg_postprocess(s, (512, 1024), meta_crop, ori)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): o = seg_postprocess(s, (512, 1024), meta_crop, ori)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
print('two-stage -> 8x683x1024: %.3f ms' % (dt*1e3))
# torch reference path on the GPU for comparison
import torch.nn.functional as F
for _ in range(2): r = F.softmax(F.interpolate(s, size=(512,1024), mode='bilinear', align_corners=False), dim=1).argmax(1)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): r = F.softmax(F.interpolate(s, size=(512,1024), mode='bilinear', align_corners=False), dim=1).argmax(1)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
print('torch-ROCm eager resize+softmax+argmax: %.3f ms' % (dt*1e3), 'agree', float((r==o_same).float().mean()) if False else '')
PY
