"""per-block cycle stamps of one GEMM launch: start / after prologue / after main loop / after epilogue."""
import sys, ctypes, torch
sys.path.insert(0, '.')
from ddp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
lib.ddp_debug_set_stamps.argtypes = [ctypes.c_void_p]
import os
for K, blocks in ((256, 256), (256, 2048)) if os.environ.get('DDP_GEMM_STAGGER','0')=='0' else ((256, 2048),):
    M, N = blocks * 128, 256
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    grid = (blocks + 7) // 8 * 8
    stamps = torch.zeros(grid * 4, dtype=torch.int64, device=dev)
    def run():
        _lib.check(lib.ddp_linear(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 0, st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    lib.ddp_debug_set_stamps(stamps.data_ptr())
    run(); torch.cuda.synchronize()
    lib.ddp_debug_set_stamps(None)
    s = stamps.cpu().reshape(grid, 4).double()
    s = s[s[:, 3] > 0]
    t0 = s[:, 0].min()
    clk = 2400.0  # shader cycles
    pro, main, epi = (s[:, 1] - s[:, 0]), (s[:, 2] - s[:, 1]), (s[:, 3] - s[:, 2])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print(f'launch {e0.elapsed_time(e1)*100:.1f} us', end=' ')
    print(f'K={K} blocks={blocks}: total span {(s[:,3].max()-t0)/clk:.1f} us | prologue {pro.mean()/clk:.2f} (max {pro.max()/clk:.2f}) '
          f'main {main.mean()/clk:.2f} (min {main.min()/clk:.2f} max {main.max()/clk:.2f}) epilogue {epi.mean()/clk:.2f} (max {epi.max()/clk:.2f}) '
          f'| start spread: first-round max start {(s[:512,0].max()-t0)/clk:.2f} us', flush=True)
