"""micro-probe: ddp_linear (fp32 MFMA GEMM + bias) TFLOP/s for several K / N at M = 262144."""
import sys, time, torch
sys.path.insert(0, '.')
from ddp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
M = 262144
st = torch.cuda.current_stream().cuda_stream
for (N, K, gelu) in [(256, 256, 0), (256, 1024, 0), (256, 4096, 0), (1024, 256, 0), (1024, 256, 1), (96, 256, 0), (160, 256, 0)]:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    def run():
        _lib.check(lib.ddp_linear(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, gelu, st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'N={N:5d} K={K:5d} gelu={gelu}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.1f} TF', flush=True)
    del a, w, out
