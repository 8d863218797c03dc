"""Cycle stamps of the weight-gradient launch's slab loop (diagnostic build of mm16.hip with -DMM16_STAMPS, selected with
TAD_HIP_LIB): per slab, waves 0 and 4 (the two waves of one SIMD) of every workgroup.
usage: TAD_HIP_LIB=exp/libtad_stamps.so python scripts/tn_stamps.py [lstm0|lstm1|gru0]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from timeautodiff_amd import _lib, ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "lstm0"
M, H, T = 98304, 200, 24
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
li, lh = (H, 4, (0, 1, 2, 3)), (H, 4, (0, 1, 2, 3))
gi, gh = (H, 3, (0, 1, 2, -1)), (H, 3, (0, 1, -1, 2))
spec = {"lstm0": [(0, 1600, 200, 0, li), (0, 800, 200, -1, lh), (800, 800, 200, 1, lh)],
        "lstm1": [(0, 1600, 400, 0, li), (0, 800, 200, -1, lh), (800, 800, 200, 1, lh)],
        "gru0": [(0, 1600, 128, 0, gi), (0, 800, 200, -1, gh), (800, 800, 200, 1, gh)]}[which]
A = torch.randn(M, 1600, device=dev, generator=g).bfloat16()
jobs = []
for a0, na, nb, shift, gm in spec:
    B = torch.randn(M, nb, device=dev, generator=g).bfloat16()
    jobs.append(dict(b=B, c=torch.zeros(na // 4 * gm[1], nb, device=dev), a0=a0, na=na, shift=shift, gate_map=gm))
lib = _lib.load()
lib.tad_dbg_tn_stamps.argtypes = [C.c_void_p, C.c_int]
for _ in range(5):
    ops.mm16_tn(A, jobs, T=T)
torch.cuda.synchronize()
lib.tad_dbg_tn_stamps(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.mm16_tn(A, jobs, T=T)
e1.record()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
lib.tad_dbg_tn_stamps(buf, 0)
names = ["wait f0", "mma f0 issue", "wait f1", "mma f1 issue", "fetch issue", "refill issue", "wait vmcnt"]
print(f"{which}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (stamped build)")
for base, who in ((0, "wave 0"), (16, "wave 4")):
    n = buf[base + 15]
    print(who, "slabs", n)
    for i, nm in enumerate(names):
        print(f"   {nm:14s} {buf[base + i] / max(n, 1):8.1f} cycles per slab")
