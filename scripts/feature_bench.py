"""Feature kernels of a training step in isolation (sincos features, embedding lookup, positional encoding), at the
bench shape, with the result's row pitch aligned (16-byte-store form) and odd (element-per-thread form).
Usage: python scripts/feature_bench.py [rows]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from timeautodiff_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
dev = torch.device("cuda:0")
nn_, E, H = 27, 128, 200
x = torch.rand(rows, 1 + nn_, device=dev)
x[:, 0] = torch.randint(0, 8, (rows,), device=dev).float()
tables = torch.randn(8, E, device=dev)
tv = torch.randint(0, 100, (rows,), device=dev).float()
w, b = torch.randn(H // 2, device=dev), torch.randn(H // 2, device=dev)
sc = torch.exp(-torch.arange(H // 2, device=dev).float() * (np.log(10000.0) / (H // 2)))
sh = torch.zeros(H // 2, device=dev)


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dtype in (torch.bfloat16, torch.float32):
    for name, width, call in (("sincos", 16 * nn_, lambda o: ops.sincos_features(x[:, 1:], o)),
                              ("embed_fwd", E, lambda o: ops.embed_fwd(x[:, :1], tables, [0], [8], E, o)),
                              ("posenc", H, lambda o: ops.posenc_fwd(tv, 1, rows, H, 0.01, w, b, sc, sh, o))):
        res = []
        for pad in (8, 9):
            buf = torch.zeros(rows, width + pad, device=dev, dtype=dtype)
            res.append(timed(lambda: call(buf[:, :width])))
        print(f"{name:10s} {str(dtype):15s} rows {rows}: wide {res[0]:7.1f} us   narrow {res[1]:7.1f} us", flush=True)
