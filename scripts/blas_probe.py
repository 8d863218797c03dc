"""Speed of the vendor GEMM library (through torch: hipBLASLt / rocBLAS) at the product shapes of the training steps, for
comparison with this library's kernels.  usage: python scripts/blas_probe.py"""
import torch, os
dev = torch.device("cuda:0")
def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("preferred blas:", torch.backends.cuda.preferred_blas_library())
for (M, N, K) in [(98304, 1600, 400), (98304, 1600, 200), (98304, 400, 1600), (98304, 200, 800), (98304, 432, 432), (196608, 4096, 1024), (196608, 4096, 512), (196608, 1024, 4096), (98304, 2048, 512), (98304, 512, 2048), (12288, 1600, 400)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.1).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    t = timed(lambda: torch.nn.functional.linear(x, w, b))
    print(f"linear M={M} N={N} K={K}: {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TF/s", flush=True)
# weight-gradient shape: C[N,K] = dy[M,N]^T x[M,K]
for (M, N, K) in [(98304, 1600, 400), (98304, 800, 200), (196608, 4096, 1024), (196608, 2048, 512)]:
    dy = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, K, device=dev).bfloat16()
    t = timed(lambda: torch.mm(dy.t(), x))
    print(f"tn     M={M} N={N} K={K}: {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TF/s", flush=True)
