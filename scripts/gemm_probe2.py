"""micro-probe: time vs number of row tiles (occupancy / round structure) for N=256, K=256 and K=1024."""
import sys, torch
sys.path.insert(0, '.')
from ddp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
for K in (256, 1024):
    for blocks in (64, 128, 256, 384, 512, 768, 1024, 2048, 4096):
        M = blocks * 128
        N = 256
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        def run():
            _lib.check(lib.ddp_linear(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 0, st))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f'K={K} blocks={blocks:5d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TF', flush=True)
