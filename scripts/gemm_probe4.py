"""where do blocks land? co-residency map of the first dispatch round."""
import sys, ctypes, torch, collections
sys.path.insert(0, '.')
from ddp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
lib.ddp_debug_set_stamps.argtypes = [ctypes.c_void_p]
K, blocks = 256, 2048
M, N = blocks * 128, 256
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
grid = blocks
stamps = torch.zeros(grid * 5, dtype=torch.int64, device=dev)
def run():
    _lib.check(lib.ddp_linear(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 0, st))
for _ in range(2): run()
torch.cuda.synchronize()
lib.ddp_debug_set_stamps(stamps.data_ptr()); run(); torch.cuda.synchronize(); lib.ddp_debug_set_stamps(None)
s = stamps.cpu()
t = s[:grid * 4].reshape(grid, 4)
hw = s[grid * 4:]
xcc = (hw >> 32) & 0xf
hwid = hw & 0xffffffff
# gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]...
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
key = [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in range(grid)]
t0 = int(t[:, 0].min())
order = sorted(range(grid), key=lambda i: int(t[i, 0]))
first = order[:512]
bycu = collections.defaultdict(list)
for i in first: bycu[key[i]].append(i)
print('distinct CUs in first 512 starters:', len(bycu))
print('sample co-resident pairs (block ids):', [v for v in list(bycu.values())[:12]])
print('first 24 blocks -> (xcc,se,sh,cu):', [key[i] for i in range(24)])
d = collections.Counter(abs(v[0] - v[1]) for v in bycu.values() if len(v) == 2)
print('pair id differences:', d.most_common(8))
print('start times of blocks 0..15 (cycles rel):', [int(t[i, 0]) - t0 for i in range(16)])
print('start of blocks 256..263:', [int(t[i, 0]) - t0 for i in range(256, 264)], ' 512..519:', [int(t[i, 0]) - t0 for i in range(512, 520)])
