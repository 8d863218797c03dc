#!/usr/bin/env python3
"""Benchmark of the TimeAutoDiff hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): energy_data-shaped windows (28 features = 1 categorical of
cardinality 8 + 27 numericals, T = 24, latent K = 6, denoiser H = 200, 2 layers, emb 128),
bf16 matrix-core operands with fp32 accumulation and fp32 master weights.
The "sampler" and "vae" objects of the JSON line carry the two companion rates (sampled sequences/s through
all 100 reverse steps; VAE training samples/s at the reference's hidden size 512, "vae_h200" at hidden 200).
On one GPU the line also carries "extras": denoiser training at the other four workloads' shapes, the VAE at the
Metro / AirQuality shapes, an fp32 line of the headline config (each with its own roofline object), and the
reference-batch regime (B = 100) through DIFF.train_diffusion / VAE.train_autoencoder / DIFF.sample.
`python bench.py --gpus N` started as ONE process launches its N ranks itself (fresh children, before any GPU call).
A "step" is one full optimizer step of the denoiser (draws, gather, add_noise, forward, loss,
backward, gradient all-reduce, Adam) on a per-GPU batch of 4096 windows (weak scaling: the
global batch is 4096 x N).  The sampler line generates 1024 sequences per GPU through all 100
reverse steps.  Inputs are synthetic and resident in HBM before the timed region.

One JSON line is printed by rank 0: value = whole-job denoiser training samples/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SCHEMA = dict(n_bins=0, n_cats=1, n_nums=27, cards=[8])
F, T, K, H, E, LAYERS, STEPS_DIFF = 28, 24, 6, 200, 128, 2, 100
B_LOCAL = 4096
B_SAMPLE = 1024
N_DATA = 16 * B_LOCAL
PEAK_BF16 = 2.5e15       # dense bf16 MFMA FLOP/s (MI355X)
PEAK_FP32 = 1.57e14      # dense fp32 matrix FLOP/s (MI355X, v_mfma_f32_*_f32)
PEAK_HBM = 8.0e12        # HBM3E bytes/s

# The five workloads of the reference's notebooks: column schema, window length, denoiser hidden size, the VAE's
# reference batch, the per-GPU batch of the large-batch regime, the mask pattern of the VAE target.
WORKLOADS = {
    "energy_T24_H200": dict(schema=dict(n_bins=0, n_cats=1, n_nums=27, cards=[8]), T=24, H=200, vae_B=100, big_B=4096, mask="ones"),
    "stock_T24_H200": dict(schema=dict(n_bins=0, n_cats=0, n_nums=6, cards=[]), T=24, H=200, vae_B=100, big_B=4096, mask="ones"),
    "metro_T48_H512": dict(schema=dict(n_bins=0, n_cats=3, n_nums=5, cards=[12, 1, 11]), T=48, H=512, vae_B=40, big_B=4096, mask="bern"),
    "etth1_T96_H200": dict(schema=dict(n_bins=0, n_cats=0, n_nums=7, cards=[]), T=96, H=200, vae_B=100, big_B=1024, mask="ones"),
    "airq_T48_H512": dict(schema=dict(n_bins=0, n_cats=0, n_nums=13, cards=[]), T=48, H=512, vae_B=40, big_B=4096, mask="bern"),
}
VAE_HIDDEN = 512         # hidden_size of every reference VAE (Run_TV-MCG.ipynb:125 and the other notebooks)


class _Parser:
    def __init__(self, schema=None):
        self.schema = dict(schema if schema is not None else SCHEMA)

    def datatype_info(self):
        return dict(self.schema)


def synth(n, seed, dev, schema=None, T_=None):
    """Synthetic windows of a workload, generated on the device: (latent [n,T,K], time_info [n,T,8], cond [n,T,F])
    with category codes stored as floats in the leading columns (bins, then cats), numericals ~ U(0,1)."""
    schema = schema if schema is not None else SCHEMA
    T_ = T_ if T_ is not None else T
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cols = [torch.randint(0, 2, (n, T_, 1), device=dev, generator=g).float() for _ in range(schema["n_bins"])]
    cols += [torch.randint(0, c, (n, T_, 1), device=dev, generator=g).float() for c in schema["cards"][:schema["n_cats"]]]
    cols.append(torch.rand(n, T_, schema["n_nums"], device=dev, generator=g))
    cond = torch.cat(cols, -1).contiguous()
    ang = torch.rand(n, T_, 4, device=dev, generator=g) * (2 * np.pi)
    time_info = torch.cat([ang.sin(), ang.cos()], -1).contiguous()
    latent = torch.randn(n, T_, K, device=dev, generator=g)
    return latent, time_info, cond


def spawn_ranks(n):
    """`bench.py --gpus N` started as ONE process: start the N ranks as fresh children (torch.distributed.run) before
    this process has made any GPU call, relay their output (rank 0 prints the JSON line) and exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def fwd_macs_per_sample():
    """Algorithmic MACs of one denoiser forward per window (every matrix product of BiRNN_score.forward,
    DIFF.py:197-229, at the bench shape)."""
    nd, nn = 1, 27
    emb = 2 * (16 * nn) ** 2 + (E * nd + 16 * nn) * E + E * E
    per_t = (K * H + 2 * (H // 2) + (8 * H + H * H) + emb + 6 * H * (E + H) + 18 * H * H + 2 * H * H + 4 * H * H
             + 40 * H * H + 2 * H * K)
    return T * per_t


def cpu_baseline(seconds_budget=20.0):
    """The numpy oracle (a port of the reference's algorithm) timed on this host: full denoiser
    training steps (forward, loss, backward, Adam) on {Bc}-window batches of the same shape."""
    from oracle import denoiser as oden, ops as oops
    from oracle.denoiser import Schema
    from oracle.optim import Adam
    from timeautodiff_amd import DIFF
    torch.manual_seed(0)
    sd = {k: v.numpy().copy() for k, v in DIFF.init_denoiser_state(K, H, LAYERS, STEPS_DIFF, 8, E, 0, 1, 27, [8]).items()}
    sch = Schema(0, 1, 27, [8])
    rng = np.random.default_rng(0)
    Bc = 100
    cond = np.concatenate([rng.integers(0, 8, (Bc, T, 1)).astype(np.float32), rng.random((Bc, T, 27), dtype=np.float32)], -1)
    ang = rng.random((Bc, T, 4), dtype=np.float32) * np.float32(2 * np.pi)
    ti = np.concatenate([np.sin(ang), np.cos(ang)], -1)
    x0 = rng.standard_normal((Bc, T, K)).astype(np.float32)
    _, alphas = oops.schedule(100)
    opt = Adam(sd)
    n, t0 = 0, time.perf_counter()
    while True:
        t = np.sort(rng.random((Bc, T, 1), dtype=np.float32), 1)
        i = rng.integers(0, 100, Bc)
        noise = rng.standard_normal((Bc, T, K)).astype(np.float32)
        _, _, grads = oden.loss_and_grads(sd, x0, t, i, noise, ti, cond, sch, alphas)
        opt.step(grads)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 64:
            break
    # the threads numpy's BLAS actually runs its products on (the oracle is numpy; everything else in it is one thread)
    threads = 0
    try:
        from threadpoolctl import threadpool_info
        threads = max([int(lib.get("num_threads", 0)) for lib in threadpool_info() if lib.get("user_api") == "blas"] or [0])
    except Exception:                       # noqa: BLE001 -- reporting only
        threads = 0
    if threads <= 0:
        threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or torch.get_num_threads()
    return dict(value=round(n * Bc / el, 2), unit="samples/s", cores=threads, kind="port",
                sample=f"{n} training steps of the numpy oracle on {Bc}-window batches (T={T}, H={H}, F={F}), {el:.1f} s")


def trace_step(step_fn, ms_step, ops, fp=None, peak_flops=PEAK_BF16, NTR=3):
    """Per-kernel device timing of ``step_fn`` (HIP events on the launch stream around every C-ABI call, aggregated
    by the device kernel the call launched) -> (roofline object of the kernel with the largest summed duration,
    table of the twelve largest kernels)."""
    with ops.LaunchTrace() as tr:
        for _ in range(NTR):
            step_fn()
    by_kernel = tr.summary_by_kernel()
    kstat = {}
    for kern, calls in by_kernel.items():
        ent = kstat.setdefault(kern, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0, issued=0.0))
        for key, (cnt, ms, fl, by, iss) in calls.items():
            ent["calls"] += cnt
            ent["ms"] += ms
            ent["flops"] += fl * cnt            # fl, by, iss: means over the calls of the key, each call with its own cost
            ent["bytes"] += by * cnt
            ent["issued"] += iss * cnt
    dom = max(kstat, key=lambda k: kstat[k]["ms"])
    d = kstat[dom]
    avg_s = d["ms"] / d["calls"] * 1e-3
    fl_l, by_l = d["flops"] / d["calls"], d["bytes"] / d["calls"]          # algorithmic work of an average launch
    # the roofline that binds: below the ridge (peak FLOP/s / peak B/s) a kernel is HBM-bound
    hbm_bound = by_l > 0 and fl_l / by_l < peak_flops / PEAK_HBM
    if hbm_bound:
        roof = dict(bound="hbm", achieved=round(by_l / avg_s / 1e9, 1), peak=PEAK_HBM / 1e9, unit="GB/s",
                    frac=round(by_l / avg_s / PEAK_HBM, 4), traffic=None)
    else:
        roof = dict(bound="mfma", achieved=round(fl_l / avg_s / 1e12, 2), peak=peak_flops / 1e12, unit="TFLOP/s",
                    frac=round(fl_l / avg_s / peak_flops, 4), traffic=None)
    # both fractions, whichever bound the kernel's algorithmic intensity puts it under: the weight-gradient launch sits
    # near the ridge (245-316 FLOP per algorithmic byte against 312) and changes sides with the shape
    roof.update(frac_of_mfma_peak=round(fl_l / avg_s / peak_flops, 4), frac_of_hbm_peak=round(by_l / avg_s / PEAK_HBM, 4))
    roof.update(kernel=dom, avg_launch_us=round(avg_s * 1e6, 2), launches_per_step=d["calls"] // NTR,
                algorithmic_bytes_per_launch=round(by_l), algorithmic_flops_per_launch=round(fl_l),
                issued_flops_per_launch=round(d["issued"] / d["calls"]),
                tflops=round(fl_l / avg_s / 1e12, 2), issued_tflops=round(d["issued"] / d["calls"] / avg_s / 1e12, 2),
                share_of_step=round(d["ms"] / NTR / ms_step, 3),
                traced_flops_per_step=sum(v["flops"] for v in kstat.values()) / NTR)
    # The weight-gradient products of a backward pass run on a second stream beside the main one, so the durations
    # above are those of kernels that SHARE the chip (their sum exceeds the step time).  For the quality of the
    # dominant kernel by itself, the same step is traced once more with every launch on one stream.
    if fp is not None and fp.side is not None:
        side, fp.side = fp.side, None
        try:
            with ops.LaunchTrace() as tr1:
                for _ in range(NTR):
                    step_fn()
            calls1 = tr1.summary_by_kernel().get(dom, {})
        finally:
            fp.side = side
        n1 = sum(c[0] for c in calls1.values())
        ms1 = sum(c[1] for c in calls1.values())
        if n1:
            a1 = ms1 / n1 * 1e-3
            roof["alone_on_the_chip"] = dict(
                avg_launch_us=round(a1 * 1e6, 2),
                achieved=round((by_l if hbm_bound else fl_l) / a1 / (1e9 if hbm_bound else 1e12), 2),
                frac=round((by_l / PEAK_HBM if hbm_bound else fl_l / peak_flops) / a1, 4),
                note="same step with the gradient stream off: no other kernel runs beside it")
    kernels = {k: dict(launches_per_step=v["calls"] // NTR, ms_per_step=round(v["ms"] / NTR, 3),
                       avg_us=round(v["ms"] / v["calls"] * 1e3, 1),
                       tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
                       hbm_GBps=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1))
               for k, v in sorted(kstat.items(), key=lambda kv: -kv[1]["ms"])[:12]}
    return roof, kernels


def attach_profiles(roof, kernels):
    """HBM traffic and matrix-pipe occupancy per launch cannot be collected from inside this process (PMC counters
    need rocprofv3 around it): the figures measured by scripts/pmc_traffic.sh / scripts/pmc_mfma.sh on this same step
    (committed under profiles/) are attached to the kernels they belong to."""
    import glob
    try:
        tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
        tk = json.load(open(tf))["kernels"].get(roof["kernel"])
        if tk:
            # a counter run of its own (PMC passes cannot run inside this process): stamped with the average launch
            # duration of THAT run, so that figures from an older build of the kernel show as such
            roof["traffic"] = tk["traffic_bytes_per_launch"]
            roof["traffic_source"] = dict(file=os.path.relpath(tf, ROOT), avg_launch_us_in_that_run=tk.get("avg_launch_us"),
                                          launches_in_that_run=tk.get("launches"))
    except (IndexError, OSError, KeyError, ValueError):
        pass
    try:
        mf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma.json")))[-1]
        mk = json.load(open(mf)).get("train_step", {})
        for k, v in kernels.items():
            if k in mk:
                v["mfma_busy_frac"] = mk[k]["mfma_busy_frac"]
        if roof["kernel"] in mk:
            roof["mfma_busy_frac"] = mk[roof["kernel"]]["mfma_busy_frac"]
            roof["mfma_busy_source"] = os.path.relpath(mf, ROOT)
    except (IndexError, OSError, KeyError, ValueError):
        pass


def note(msg):
    """Progress line on stderr (stdout carries the one JSON line only)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


class Ctx:
    """What every section of the bench needs: device, rank layout, barrier, max-over-ranks timing."""

    def __init__(self, dev, world, rank):
        self.dev, self.world, self.rank = dev, world, rank
        self.pg = True if world > 1 else None

    def barrier(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, reps):
        """Seconds for ``reps`` calls of fn between two barriers, maximum over the ranks."""
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        self.barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt


def bench_denoiser(cx, schema, T_, H_, B_, dtype, steps, warmup, prime=8, trace=True, seed=1234):
    """Full optimizer steps of BiRNN_score at a workload's shape, B_ windows per GPU (weak scaling), device draws."""
    from timeautodiff_amd import ops
    from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer
    dev, world = cx.dev, cx.world
    F_ = schema["n_bins"] + schema["n_cats"] + schema["n_nums"]
    torch.manual_seed(seed)             # same model and same global draws on every rank
    latent, time_info, cond = synth(16 * B_, seed, dev, schema, T_)
    model = BiRNN_score(K, F_, H_, LAYERS, STEPS_DIFF, 8, E, schema["n_bins"], schema["n_cats"], schema["n_nums"], schema["cards"])
    trainer = DiffusionTrainer(model, latent, time_info, cond, batch_size=B_ * world, lr=1e-3, dtype=dtype,
                               process_group=cx.pg, draws="device", dev=dev)
    # one-off initialisation outside the measurement: the first steps of a process load the code objects, size the
    # workspace and grow torch's caching allocator (host-bound for about ten steps), whatever --warmup says
    for _ in range(prime):
        trainer.step()
    # everything allocated so far is permanent: keep Python's cyclic collector from walking it inside the timed
    # region (one full collection over torch's objects stalls the host for tens of milliseconds)
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(warmup):
        trainer.step()
    dt = cx.timed(trainer.step, steps)
    ms_step = 1e3 * dt / steps
    out = dict(samples_per_s=round(B_ * world * steps / dt, 1), ms_per_step=round(ms_step, 3), per_gpu_batch=B_, T=T_, H=H_, F=F_,
               dtype=dtype, steps=steps)
    roof = kernels = None
    if trace:
        roof, kernels = trace_step(trainer.step, ms_step, ops, fp=trainer.eng.fp,
                                   peak_flops=PEAK_BF16 if dtype == "bf16" else PEAK_FP32)
    return out, roof, kernels, (model, trainer, time_info, cond)


def bench_vae(cx, schema, T_, hidden, B_, dtype, steps, mask_kind="ones", with_cond=False, trace=True):
    """Full optimizer steps of DeapStack (the reference's train_autoencoder iteration without its per-iteration
    validation pass) on B_ windows per GPU."""
    from timeautodiff_amd import ops
    from timeautodiff_amd.VAE import DeapStack, VAETrainer
    dev, world = cx.dev, cx.world
    F_ = schema["n_bins"] + schema["n_cats"] + schema["n_nums"]
    NV = 4 * B_
    _, _, resp = synth(NV + 256, 77, dev, schema, T_)
    if mask_kind == "bern":
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        mask = (torch.rand(NV + 256, T_, F_, device=dev, generator=g) < 0.9).float()      # 10 % of the entries held out
    else:
        mask = torch.ones(NV + 256, T_, F_, device=dev)
    cond_all = resp * (1 - mask) if with_cond else None
    dd = dict(response_train=resp[:NV], target_mask_train=mask[:NV], cond_train=cond_all[:NV] if with_cond else None,
              response_val=resp[NV:], target_mask_val=mask[NV:], cond_val=cond_all[NV:] if with_cond else None)
    torch.manual_seed(4321)
    ae = DeapStack(1, schema["n_bins"], schema["n_cats"], schema["n_nums"], schema["cards"], F_, hidden, LAYERS, E, 8, K)
    vt = VAETrainer(ae, dd, batch_size=B_ * world, lr=2e-4, weight_decay=1e-6, dtype=dtype, process_group=cx.pg, dev=dev)
    import random
    random.seed(99)                       # same batch indices on every rank
    for _ in range(3):
        vt.train_step(0.1)
    dv = cx.timed(lambda: vt.train_step(0.1), steps)
    ms_step = 1e3 * dv / steps
    out = dict(metric="vae_train_samples_per_sec", value=round(B_ * world * steps / dv, 1), unit="samples/s", steps=steps,
               ms_per_step=round(ms_step, 3), per_gpu_batch=B_, dtype=dtype,
               model=f"DeapStack F={F_} hidden {hidden}, {LAYERS}-layer GRU x2, emb {E}, latent {K}, T={T_}, mask {mask_kind}")
    if trace:
        roof, _ = trace_step(lambda: vt.train_step(0.1), ms_step, ops, fp=vt.eng.fp, peak_flops=PEAK_BF16 if dtype == "bf16" else PEAK_FP32)
        out["roofline"] = roof
    return out


def bench_reference_batch(cx, dtype):
    """The reference's own batch sizes through the reference's entry points (energy shape): DIFF.train_diffusion at
    B = 100, VAE.train_autoencoder at B = 100 (hidden 512, validation pass every iteration, like the reference),
    DIFF.sample on a whole split (1577 windows).  At these sizes a step is bound by launch latency and host work,
    not by the device; the cost of building the model is removed by timing two run lengths and taking the slope."""
    from timeautodiff_amd import DIFF, VAE
    dev = cx.dev
    N = 1577                                          # training windows of the energy_data split
    latent, time_info, cond = synth(N + 177, 4242, dev)
    dd = dict(parser=_Parser(), cond_train=cond[:N], time_info_train=time_info[:N])

    def run_diff(n, graph):
        torch.manual_seed(1)
        import random
        random.seed(1)
        t0 = time.perf_counter()
        m = DIFF.train_diffusion(latent[:N], None, dd, H, LAYERS, STEPS_DIFF, n, dtype=dtype, draws="reference", graph=graph)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, m
    ms_d = {}
    for graph in (False, True):            # kernel by kernel (the default), then as one captured HIP graph per step
        run_diff(10, graph)
        ta, _ = run_diff(20, graph)
        tb, model = run_diff(220, graph)
        ms_d[graph] = 1e3 * (tb - ta) / 200
    ms_diff = ms_d[False]
    ones = torch.ones(N + 177, T, F, device=dev)
    vd = dict(parser=_Parser(), response_train=cond[:N], target_mask_train=ones[:N], cond_train=None,
              response_val=cond[N:], target_mask_val=ones[N:], cond_val=None)

    def run_vae(n, graph):
        torch.manual_seed(2)
        import random
        random.seed(2)
        t0 = time.perf_counter()
        VAE.train_autoencoder(None, 1, VAE_HIDDEN, LAYERS, 2e-4, 1e-6, n, 100, None, 1e-5, 0.1, E, 8, K, dev, vd, dtype=dtype, graph=graph)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    ms_v = {}
    for graph in (False, True):
        run_vae(5, graph)
        va = run_vae(10, graph)
        vb = run_vae(110, graph)
        ms_v[graph] = 1e3 * (vb - va) / 100
    ms_vae = ms_v[True]
    tgrid = torch.linspace(0, 1, T, device=dev).view(1, -1, 1).repeat(N, 1, 1)
    DIFF.sample(tgrid, N, T, K, STEPS_DIFF, model, time_info[:N], cond[:N], None, noise_source="device")
    ds = cx.timed(lambda: DIFF.sample(tgrid, N, T, K, STEPS_DIFF, model, time_info[:N], cond[:N], None, noise_source="device"), 2) / 2
    return dict(dtype=dtype, workload="energy_T24_H200",
                train_diffusion=dict(batch=100, ms_per_step=round(ms_diff, 3), samples_per_s=round(100 / ms_diff * 1e3, 1), draws="reference",
                                     ms_per_step_graph=round(ms_d[True], 3), launches="kernel by kernel (the default; graph: one HIP graph per step)"),
                train_autoencoder=dict(batch=100, hidden=VAE_HIDDEN, ms_per_iteration=round(ms_vae, 3),
                                       samples_per_s=round(100 / ms_vae * 1e3, 1), includes="validation pass over 177 windows",
                                       ms_per_iteration_kernel_by_kernel=round(ms_v[False], 3), replay="one HIP graph per step and per validation pass"),
                sample=dict(batch=N, ms_per_call=round(1e3 * ds, 2), seqs_per_s=round(N / ds, 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other workloads, the fp32 line and the reference-batch regime")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a.gpus)             # never returns; no GPU call has been made in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == a.gpus or (a.gpus == 1 and world == 1), f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    cx = Ctx(dev, world, rank)

    from timeautodiff_amd import DIFF

    # ---- headline: denoiser training at the energy shape, 4096 windows per GPU
    note("denoiser training, energy shape")
    head, roof, kernels, (model, trainer, time_info, cond) = bench_denoiser(cx, SCHEMA, T, H, B_LOCAL, a.dtype, a.steps, a.warmup)
    attach_profiles(roof, kernels)
    ms_step, train_rate = head["ms_per_step"], head["samples_per_s"]

    # ---- sampler: 1k sequences per GPU, 100 reverse steps
    sampler = None
    if not a.no_sampler:
        note("sampler")
        Bs = B_SAMPLE * world
        tgrid = torch.linspace(0, 1, T, device=dev).view(1, -1, 1).repeat(Bs, 1, 1)
        ti_s, cond_s = time_info[:Bs], cond[:Bs]
        run = lambda: DIFF.sample(tgrid, Bs, T, K, STEPS_DIFF, model, ti_s, cond_s, None, noise_source="device", process_group=cx.pg)  # noqa: E731
        run()
        reps = 3
        ds = cx.timed(run, reps)
        sampler = dict(metric="sampled_seqs_per_sec", value=round(Bs * reps / ds, 1), unit="seqs/s", batch_per_gpu=B_SAMPLE,
                       reverse_steps=STEPS_DIFF, ms_per_call=round(1e3 * ds / reps, 2))

    # ---- VAE: full optimizer steps of the mixed-type autoencoder at the reference's hidden size (and at hidden 200,
    # the figure earlier rounds reported)
    vae = vae_h200 = None
    if not a.no_vae:
        vsteps = max(3, a.steps // 2)
        note("VAE training, hidden 512 and 200")
        vae = bench_vae(cx, SCHEMA, T, VAE_HIDDEN, B_LOCAL, a.dtype, vsteps)
        vae_h200 = bench_vae(cx, SCHEMA, T, H, B_LOCAL, a.dtype, vsteps, trace=False)

    # ---- the other workloads, the fp32 line, the reference-batch regime (one GPU only: they are not scaling lines)
    extras = None
    if world == 1 and not a.no_extras:
        extras = dict(workloads={}, vae_workloads={})
        for name, wl in WORKLOADS.items():
            if name == "energy_T24_H200":
                continue
            note(f"denoiser training, {name}")
            o, r, _, keep = bench_denoiser(cx, wl["schema"], wl["T"], wl["H"], wl["big_B"], a.dtype, 5, 2, prime=4, seed=2000)
            o["roofline"] = r
            # the sampler at this workload's shape: 1024 sequences through all 100 reverse steps
            m_w, _, ti_w, cond_w = keep
            tg = torch.linspace(0, 1, wl["T"], device=dev).view(1, -1, 1).repeat(B_SAMPLE, 1, 1)
            run_w = lambda: DIFF.sample(tg, B_SAMPLE, wl["T"], K, STEPS_DIFF, m_w, ti_w[:B_SAMPLE], cond_w[:B_SAMPLE], None, noise_source="device")  # noqa: E731
            run_w()
            ds_w = cx.timed(run_w, 2) / 2
            o["sampler"] = dict(seqs_per_s=round(B_SAMPLE / ds_w, 1), ms_per_call=round(1e3 * ds_w, 2), batch=B_SAMPLE)
            del keep, m_w, ti_w, cond_w
            extras["workloads"][name] = o
            torch.cuda.empty_cache()
        for name in ("metro_T48_H512", "airq_T48_H512"):
            wl = WORKLOADS[name]
            note(f"VAE training, {name}")
            extras["vae_workloads"][name] = bench_vae(cx, wl["schema"], wl["T"], VAE_HIDDEN, wl["big_B"], a.dtype, 4, mask_kind=wl["mask"],
                                                      with_cond=True)
            torch.cuda.empty_cache()
        note("denoiser training, fp32")
        o, r, _, keep = bench_denoiser(cx, SCHEMA, T, H, B_LOCAL, "fp32", 5, 2, prime=4)
        del keep
        o["roofline"] = r
        extras["fp32"] = o
        torch.cuda.empty_cache()
        note("reference-batch regime")
        extras["reference_batch"] = bench_reference_batch(cx, a.dtype)

    if rank == 0:
        macs = fwd_macs_per_sample()
        roof_traced = roof["traced_flops_per_step"]
        out = dict(metric="denoiser_train_samples_per_sec", value=round(train_rate, 1), unit="samples/s", n_gpus=world,
                   steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype=a.dtype, data="synthetic",
                   config=dict(workload="energy_data-shaped 24-step windows (F=28: 1 cat card 8 + 27 num, K=6), "
                                        "BiRNN_score H=200 2 layers emb 128, full optimizer step, per-GPU batch 4096",
                               global_batch=B_LOCAL * world, per_gpu_batch=B_LOCAL, T=T, H=H, parallelism=f"dp{world}"),
                   model_flops_frac_of_bf16_peak=round(3 * 2 * macs * train_rate / (PEAK_BF16 * world), 4),
                   # the algorithmic FLOPs the traced launches of one step add up to, against 3 x the forward's (a step
                   # computes no gradient wrt the model's inputs, so the sum sits a few percent below 3 x)
                   flops_check=dict(traced_per_step=roof.pop("traced_flops_per_step"), model_per_step=3.0 * 2.0 * macs * B_LOCAL,
                                    ratio=round(roof_traced / (3.0 * 2.0 * macs * B_LOCAL), 4)),
                   roofline=roof, kernels=kernels, sampler=sampler, vae=vae, vae_h200=vae_h200, extras=extras)
        if not a.no_cpu_baseline and world == 1:
            note("CPU baseline (numpy oracle)")
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
