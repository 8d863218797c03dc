#!/usr/bin/env python3
"""Device timing of the sequence kernels at the bench shape.  usage: rnn_bench.py [bf16|fp32] [B]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from timeautodiff_amd import _lib, ops  # noqa: E402


def main():
    prec = ops.BF16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else ops.F32
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    T, H, nd = 24, 200, 2
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for cell, G, S in ((0, 4, 5), (1, 3, 4)):
        xproj = torch.randn(B * T, nd * G * H, device=dev)
        out = torch.zeros(B * T, nd * H, device=dev)
        stash = torch.zeros(nd, B * T, S * H, device=dev)
        dout = torch.randn(B * T, nd * H, device=dev)
        dg = torch.zeros(B * T, nd * G * H, device=dev)
        dgh = torch.zeros(B * T, nd * G * H, device=dev)
        w = torch.randn(G * H, H, device=dev) * 0.05
        nb = ops.rnn_packed_bytes(prec, H, G)
        pf = torch.empty(nb, dtype=torch.uint8, device=dev)
        pb = torch.empty(nb, dtype=torch.uint8, device=dev)
        ops.rnn_pack_weights(prec, H, G, w, pf, pb)
        bh = torch.zeros(G * H, device=dev)
        fd = [dict(xproj=xproj[:, d * G * H:(d + 1) * G * H], w_packed=pf, b_hh=bh if cell else None, out=out[:, d * H:(d + 1) * H],
                   stash=stash[d], reverse=bool(d)) for d in range(nd)]
        fi = [dict(f, stash=None) for f in fd]
        bd = [dict(dout=dout[:, d * H:(d + 1) * H], stash=stash[d], out=out[:, d * H:(d + 1) * H], w_packed_t=pb,
                   dgates=dg[:, d * G * H:(d + 1) * G * H], dgates_h=dgh[:, d * G * H:(d + 1) * G * H] if cell else None,
                   reverse=bool(d)) for d in range(nd)]

        def timeit(fn, reps=5):
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        tf = timeit(lambda: ops.rnn_fwd(cell, prec, B, T, H, fd))
        ti = timeit(lambda: ops.rnn_fwd(cell, prec, B, T, H, fi))
        tb = timeit(lambda: ops.rnn_bwd(cell, prec, B, T, H, bd))
        print(f"cell={cell} B={B}: fwd(train) {tf:.1f} us  fwd(infer) {ti:.1f} us  bwd {tb:.1f} us   per step: {tf / T:.2f} / {ti / T:.2f} / {tb / T:.2f} us")
        if hasattr(lib, "tad_dbg_rnn_stamps"):
            buf = (ctypes.c_ulonglong * 8)()
            for name, f in (("train", fd), ("infer", fi)):
                lib.tad_dbg_rnn_stamps(None, 1)
                ops.rnn_fwd(cell, prec, B, T, H, f)
                torch.cuda.synchronize()
                lib.tad_dbg_rnn_stamps(buf, 0)
                n = max(buf[7], 1)
                names = ["seed+wait", "mfma stream", "prefetch issue", "gates+stores", "barrier"]
                print(f"   fwd {name}: cycles per step (wave 0): " + ", ".join(f"{nm} {buf[i] / n:.0f}" for i, nm in enumerate(names)))


if __name__ == "__main__":
    main()
