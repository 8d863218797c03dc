"""One forward sequence launch at H = 512 (bf16) through tad_rnn_fwd_ws, timed; with a library built with -DCL_STAMPS
(TAD_HIP_LIB=...) also the cycle counts of the phases of a (step, half) of the cluster form.
usage: python scripts/cl_one.py [cell 0|1] [B] [T] [stash 0|1] [ndir]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from timeautodiff_amd import _lib, ops  # noqa: E402

cell = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 48
stash = int(sys.argv[4]) if len(sys.argv) > 4 else 1
nd = int(sys.argv[5]) if len(sys.argv) > 5 else 2
H, G = 512, (4 if cell == 0 else 3)
dev = torch.device("cuda:0")
torch.manual_seed(0)
BT, H4 = B * T, 4 * H
xproj = (torch.randn(BT, nd * H4, device=dev) * 0.5).to(torch.bfloat16)
whh = (torch.rand(nd, G * H, H, device=dev) * 2 - 1) / H ** 0.5
bhh = (torch.rand(nd, G * H, device=dev) * 2 - 1) / H ** 0.5
out = torch.zeros(BT, nd * H, device=dev)
out16 = torch.zeros(BT, nd * H, dtype=torch.bfloat16, device=dev)
gates = torch.zeros(BT, nd * H4, dtype=torch.bfloat16, device=dev)
cst = torch.zeros(BT, nd * H, device=dev)
nbytes = ops.rnn_packed_bytes(1, H, G)
pf = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(nd)]
dirs = []
for d in range(nd):
    ops.rnn_pack_weights(1, H, G, whh[d], pf[d], None)
    dirs.append(dict(xproj=xproj[:, d * H4:(d + 1) * H4], w_packed=pf[d], b_hh=bhh[d] if cell == 1 else None,
                     out=out[:, d * H:(d + 1) * H], out16=out16[:, d * H:(d + 1) * H],
                     gates=gates[:, d * H4:(d + 1) * H4] if stash else None,
                     cstate=cst[:, d * H:(d + 1) * H] if (stash and cell == 0) else None, reverse=bool(d & 1)))
nws = ops.rnn_workspace_bytes(cell, 1, 1, B, T, H, nd)
ws = torch.zeros(max(nws, 16), dtype=torch.uint8, device=dev) if nws > 0 else None
lib = _lib.load()
has_st = hasattr(lib, "tad_dbg_cl_stamps")
for rep in range(3):
    ops.rnn_fwd(cell, 1, B, T, H, dirs, workspace=ws)
torch.cuda.synchronize()
if has_st:
    lib.tad_dbg_cl_stamps(None, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for rep in range(n):
    ops.rnn_fwd(cell, 1, B, T, H, dirs, workspace=ws)
e1.record()
torch.cuda.synchronize()
if ws is not None:
    ops.rnn_ws_status(ws)
us = e0.elapsed_time(e1) * 1e3 / n
print(f"cell {cell} B {B} T {T} stash {stash} ndir {nd}: {us:.1f} us per launch  kernel {lib.tad_last_kernel().decode()}")
if has_st:
    buf = (C.c_ulonglong * 8)()
    lib.tad_dbg_cl_stamps(buf, 0)
    names = ["xproj issue", "poll + barrier", "fetch -> LDS + barrier", "seed + MFMA", "gates + stores issued", "barrier", "publish + drain",
             "barrier"]
    ntile = -(-B // 128)
    wgs = min(ntile * nd, 32) * 8
    per = n * T * 2 * (ntile * nd * 8)            # (step, half) visits summed over workgroups
    tot = sum(buf)
    for k in range(8):
        print(f"  {names[k]:26s} {buf[k] / per:9.0f} cyc per (step, half)   {100.0 * buf[k] / max(tot, 1):5.1f} %")
    print(f"  total {tot / per:9.0f} cyc per (step, half) at 100 MHz x ? (s_memtime ticks)")
