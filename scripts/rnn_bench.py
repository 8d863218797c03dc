#!/usr/bin/env python3
"""Device timing of the sequence kernels at the bench shape.  usage: rnn_bench.py [bf16|fp32] [B] [store: bf16|fp32]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from timeautodiff_amd import ops  # noqa: E402


def main():
    prec = ops.BF16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else ops.F32
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    sdt = torch.bfloat16 if (prec == ops.BF16 and (len(sys.argv) < 4 or sys.argv[3] == "bf16")) else torch.float32
    T, H, nd = int(os.environ.get("RNN_T", "24")), int(os.environ.get("RNN_H", "200")), 2      # (environment: other shapes)
    dev = torch.device("cuda:0")
    for cell, G in ((0, 4), (1, 3)):
        xproj = torch.randn(B * T, nd * 4 * H, device=dev).to(sdt)
        out = torch.zeros(B * T, nd * H, device=dev)
        gates = torch.zeros(B * T, nd * 4 * H, dtype=sdt, device=dev)
        cst = torch.zeros(B * T, nd * H, device=dev)
        dout = torch.randn(B * T, nd * H, device=dev)
        dg = torch.zeros(B * T, nd * 4 * H, dtype=sdt, device=dev)
        w = torch.randn(G * H, H, device=dev) * 0.05
        nb = ops.rnn_packed_bytes(prec, H, G)
        pf = torch.empty(nb, dtype=torch.uint8, device=dev)
        pb = torch.empty(nb, dtype=torch.uint8, device=dev)
        ops.rnn_pack_weights(prec, H, G, w, pf, pb)
        bh = torch.zeros(G * H, device=dev)
        sl = lambda d: slice(d * 4 * H, (d + 1) * 4 * H)
        hl = lambda d: slice(d * H, (d + 1) * H)
        fd = [dict(xproj=xproj[:, sl(d)], w_packed=pf, b_hh=bh if cell else None, out=out[:, hl(d)], gates=gates[:, sl(d)],
                   cstate=None if cell else cst[:, hl(d)], reverse=bool(d)) for d in range(nd)]
        fi = [dict(f, gates=None, cstate=None) for f in fd]
        bd = [dict(dout=dout[:, hl(d)], gates=gates[:, sl(d)], cstate=None if cell else cst[:, hl(d)],
                   out=out[:, hl(d)] if cell else None, w_packed_t=pb, dgates=dg[:, sl(d)], reverse=bool(d)) for d in range(nd)]

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
        print(f"cell={cell} B={B} store={sdt}: fwd(train) {tf:.1f} us  fwd(infer) {ti:.1f} us  bwd {tb:.1f} us   "
              f"per step: {tf / T:.2f} / {ti / T:.2f} / {tb / T:.2f} us")
        from timeautodiff_amd import _lib
        lib = _lib.load()
        if hasattr(lib, "tad_dbg_rnn_stamps"):          # diagnostic build (-DEXP_STAMPS) only
            import ctypes
            for nm, dirs_ in (("train", fd), ("infer", fi), ("bwd", bd)):
                buf = (ctypes.c_ulonglong * 8)()
                lib.tad_dbg_rnn_stamps(None, 1)
                if nm == "bwd":
                    ops.rnn_bwd(cell, prec, B, T, H, dirs_)
                else:
                    ops.rnn_fwd(cell, prec, B, T, H, dirs_)
                torch.cuda.synchronize()
                lib.tad_dbg_rnn_stamps(buf, 0)
                n = max(buf[7], 1)
                names = (["operand fetch (drained)", "gate math + stores", "barrier 1", "mma", "barrier 2"] if nm == "bwd"
                         else ["seed", "mma", "xproj prefetch issue", "gate math + stores", "barrier"])
                print(f"  {nm} cycles per step (wave 0): " + ", ".join(f"{k} {buf[i] / n:.0f}" for i, k in enumerate(names)))


if __name__ == "__main__":
    main()
