import sys, os
sys.path.insert(0, '.')
import torch
from timeautodiff_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
M = 98304
big = torch.empty(256 * 1024 * 1024, device=dev)      # flush buffer: keeps A out of the caches between launches
for (N, K) in [(200, 200), (128, 128), (200, 408), (200, 800), (400, 200), (432, 432), (1600, 200)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    Wp = ops.w16_image(N, K, dev); Wp[:N, :K] = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.mm16_nt(A, Wp, None, C))
    def cold():
        big.add_(1.0)
        ops.mm16_nt(A, Wp, None, C)
    t_flush = timeit(lambda: big.add_(1.0), reps=5)
    tc = timeit(cold, reps=5) - t_flush
    print(f"N={N} K={K}: warm {t:6.1f} us   cold {tc:6.1f} us")
