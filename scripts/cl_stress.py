"""Stress run of the cluster-form forward sequence kernel (H = 512): many launches of one problem, eager and replayed from a
HIP graph, idle and beside memory-bound work, every result compared bit for bit with the first.  Prints where a launch that
differs first differs (direction, tile of 128 rows, quarter, time step, owning workgroups) and the workspace's status.

  python scripts/cl_stress.py [launches per mode, default 1500] [cell 0 = LSTM | 1 = GRU | 2 = both] [B, default 700] [T, default 7]
(B = 4200: 66 (tile, direction) items on 32 groups, so that groups walk several items and reuse their exchange buffers)
"""
import sys
import time
import torch

sys.path.insert(0, ".")
from timeautodiff_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
CELLS = {0: [0], 1: [1], 2: [0, 1]}[int(sys.argv[2]) if len(sys.argv) > 2 else 2]
dev = torch.device("cuda:0")
H = 512


def problem(cell, B, T, nd, seed):
    G = 4 if cell == 0 else 3
    g = torch.Generator(device="cpu").manual_seed(seed)
    BT, H4 = B * T, 4 * H
    xproj = (torch.randn(BT, nd * H4, generator=g) * 0.7).to(dev).to(torch.bfloat16)
    whh = ((torch.rand(nd, G * H, H, generator=g) * 2 - 1) / H ** 0.5).to(dev)
    bhh = ((torch.rand(nd, G * H, generator=g) * 2 - 1) / H ** 0.5).to(dev)
    nbytes = ops.rnn_packed_bytes(1, H, G)
    pf = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(nd)]
    for d in range(nd):
        ops.rnn_pack_weights(1, H, G, whh[d], pf[d], None)

    def run(workspace, bufs=None):
        if bufs is None:
            bufs = (torch.zeros(BT, nd * H, device=dev), torch.zeros(BT, nd * H, dtype=torch.bfloat16, device=dev),
                    torch.zeros(BT, nd * H4, dtype=torch.bfloat16, device=dev), torch.zeros(BT, nd * H, device=dev))
        out, out16, gates, cst = bufs
        dirs = [dict(xproj=xproj[:, d * H4:(d + 1) * H4], w_packed=pf[d], b_hh=bhh[d] if cell == 1 else None,
                     out=out[:, d * H:(d + 1) * H], out16=out16[:, d * H:(d + 1) * H], gates=gates[:, d * H4:(d + 1) * H4],
                     cstate=cst[:, d * H:(d + 1) * H] if cell == 0 else None, reverse=bool(d & 1)) for d in range(nd)]
        ops.rnn_fwd(cell, 1, B, T, H, dirs, workspace=workspace)
        return bufs
    return run


def where(got, ref, T, nd):
    """First time step (in processing order) at which `out` differs, per (direction, tile, quarter), with the members whose units differ."""
    bad = (got != ref).nonzero()
    if bad.numel() == 0:
        return "out equal"
    rows, cols = bad[:, 0], bad[:, 1]
    b, t = rows // T, rows % T
    d, unit = cols // H, cols % H
    step = torch.where(d % 2 == 1, T - 1 - t, t)
    key = (d * 1000 + b // 128) * 4 + (b % 128) // 32
    lines = []
    for k in key.unique().tolist():
        m = key == k
        s0 = int(step[m].min())
        m0 = m & (step == s0)
        members = sorted(set((unit[m0] // 64).tolist()))
        lines.append(f"dir {k // 4 // 1000} tile {k // 4 % 1000} quarter {k % 4}: first step {s0}, members {members}, "
                     f"{int(m0.sum())} values there, rows {sorted(set((b[m0] % 32).tolist()))[:6]}.., {int(m.sum())} in all")
    return f"{bad.shape[0]} values of out differ\n    " + "\n    ".join(lines[:12])


def status(ws):
    try:
        ops.rnn_ws_status(ws)
        return "status clean"
    except Exception as e:  # noqa: BLE001
        return f"status: {e}"


def memset_node_check():
    """A captured hipMemsetAsync(buf, 0, 16 KB) replayed between other work of the process: what the words hold afterwards."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    buf = torch.full((4096,), 7, dtype=torch.int32, device=dev)
    x = torch.randn(4900, 1024, device=dev)
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, 16384, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    seen = {}
    for rep in range(40):
        n = int((x > 0.5 + 0.01 * rep).nonzero().shape[0])
        same = torch.equal(x, x)
        buf.fill_(7)
        g.replay()
        torch.cuda.synchronize()
        vals = tuple(sorted(set(buf.tolist())))
        seen[vals] = seen.get(vals, 0) + 1
        if rep < 3:
            print(f"  memset node replay {rep}: words hold {vals[:4]} (nonzero count just before: {n}, {same})", flush=True)
    print(f"memset node, 40 replays: values of the words -> replays: {dict(list(seen.items())[:6])}", flush=True)


def main():
    memset_node_check()
    B, T, nd = (int(sys.argv[3]) if len(sys.argv) > 3 else 700), (int(sys.argv[4]) if len(sys.argv) > 4 else 7), 2
    for cell in CELLS:
        run = problem(cell, B, T, nd, 77 + cell)
        nws = ops.rnn_workspace_bytes(cell, 1, 1, B, T, H, nd)
        ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
        ref = run(ws)
        torch.cuda.synchronize()
        print(f"cell {cell}: {status(ws)}", flush=True)
        bufs = run(ws)
        side = torch.cuda.Stream()
        big = torch.empty(64 << 20, device=dev)
        big2 = torch.empty_like(big)
        main_s = torch.cuda.current_stream()
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(main_s)
        with torch.cuda.stream(cap):
            gb = run(ws)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=cap):
                run(ws, gb)
        torch.cuda.synchronize()
        for mode in ("eager idle", "eager loaded", "graph idle", "graph loaded", "graph zeroed outputs"):
            nbad, t0 = 0, time.time()
            for rep in range(N):
                if "loaded" in mode and rep % 4 == 0:
                    side.wait_stream(main_s)
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            ops.copy_2d(big.view(-1, 4096), big2.view(-1, 4096))
                if mode.startswith("graph"):
                    if "zeroed" in mode:
                        for t_ in gb:
                            t_.zero_()
                    g.replay()
                    got = gb
                else:
                    got = run(ws, bufs)
                eq = [torch.equal(a, b_) for a, b_ in zip(got, ref)]
                ctr = ws[:16384].view(torch.int32).view(-1, 32)[:, 0]
                if rep < 2 or (not all(eq) and nbad < 4):
                    print(f"  {mode} launch {rep}: counters after the launch: {sorted(set(ctr[:4 * min(32, 2 * -(-B // 128))].tolist()))}", flush=True)
                if not all(eq):
                    nbad += 1
                    if nbad <= 4:
                        print(f"  {mode} launch {rep}: equal {eq}; {where(got[0], ref[0], T, nd)}; {status(ws)}", flush=True)
                        ws[16384:16388].zero_()
            main_s.wait_stream(side)
            torch.cuda.synchronize()
            print(f"cell {cell} {mode}: {nbad} of {N} launches differ ({time.time() - t0:.1f} s)", flush=True)


main()
