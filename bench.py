#!/usr/bin/env python3
"""Benchmark of the TimeAutoDiff hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): energy_data-shaped windows (28 features = 1 categorical of
cardinality 8 + 27 numericals, T = 24, latent K = 6, denoiser H = 200, 2 layers, emb 128),
bf16 matrix-core operands with fp32 accumulation and fp32 master weights.
The "sampler" and "vae" objects of the JSON line carry the two companion rates (sampled sequences/s through
all 100 reverse steps; VAE training samples/s).
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
PEAK_HBM = 8.0e12        # HBM3E bytes/s


class _Parser:
    def datatype_info(self):
        return dict(SCHEMA)


def synth(n, seed, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cat = torch.randint(0, SCHEMA["cards"][0], (n, T, 1), device=dev, generator=g).float()
    nums = torch.rand(n, T, SCHEMA["n_nums"], device=dev, generator=g)
    cond = torch.cat([cat, nums], -1).contiguous()
    ang = torch.rand(n, T, 4, device=dev, generator=g) * (2 * np.pi)
    time_info = torch.cat([ang.sin(), ang.cos()], -1).contiguous()
    latent = torch.randn(n, T, K, device=dev, generator=g)
    return latent, time_info, cond


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
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or torch.get_num_threads()
    return dict(value=round(n * Bc / el, 2), unit="samples/s", cores=threads, kind="port",
                sample=f"{n} training steps of the numpy oracle on {Bc}-window batches (T={T}, H={H}, F={F}), {el:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    a = ap.parse_args()

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

    from timeautodiff_amd import DIFF, ops
    from timeautodiff_amd.DIFF import BiRNN_score, DiffusionTrainer

    torch.manual_seed(1234)             # same model and same global draws on every rank
    latent, time_info, cond = synth(N_DATA, 1234, dev)
    model = BiRNN_score(K, F, H, LAYERS, STEPS_DIFF, 8, E, 0, 1, 27, [8])
    pg = True if world > 1 else None
    trainer = DiffusionTrainer(model, latent, time_info, cond, batch_size=B_LOCAL * world, lr=1e-3, dtype=a.dtype,
                               process_group=pg, draws="device", dev=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one-off initialisation outside the measurement: the first steps of a process load the code objects, size the
    # workspace and grow torch's caching allocator (host-bound for about ten steps), whatever --warmup says
    for _ in range(8):
        trainer.step()
    # everything allocated so far is permanent: keep Python's cyclic collector from walking it inside the timed
    # region (one full collection over torch's objects stalls the host for tens of milliseconds)
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(a.warmup):
        trainer.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_step = 1e3 * dt / a.steps
    train_rate = B_LOCAL * world * a.steps / dt

    # ---- per-kernel device timing of the same step (separate pass after the timed region; HIP events
    # on the launch stream around every C-ABI call, aggregated by the device kernel the call launched)
    NTR = 3
    with ops.LaunchTrace() as tr:
        for _ in range(NTR):
            trainer.step()
    by_kernel = tr.summary_by_kernel()
    kstat = {}
    for kern, calls in by_kernel.items():
        ent = kstat.setdefault(kern, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
        for key, (cnt, ms, fl, by) in calls.items():
            ent["calls"] += cnt
            ent["ms"] += ms
            ent["flops"] += fl * cnt
            ent["bytes"] += by * cnt
    dom = max(kstat, key=lambda k: kstat[k]["ms"])
    d = kstat[dom]
    avg_s = d["ms"] / d["calls"] * 1e-3
    fl_l, by_l = d["flops"] / d["calls"], d["bytes"] / d["calls"]          # algorithmic work of an average launch
    # the roofline that binds: below the ridge (peak FLOP/s / peak B/s) a kernel is HBM-bound
    hbm_bound = by_l > 0 and fl_l / by_l < PEAK_BF16 / PEAK_HBM
    if hbm_bound:
        roof = dict(bound="hbm", achieved=round(by_l / avg_s / 1e9, 1), peak=PEAK_HBM / 1e9, unit="GB/s",
                    frac=round(by_l / avg_s / PEAK_HBM, 4), traffic=None)
    else:
        roof = dict(bound="mfma", achieved=round(fl_l / avg_s / 1e12, 2), peak=PEAK_BF16 / 1e12, unit="TFLOP/s",
                    frac=round(fl_l / avg_s / PEAK_BF16, 4), traffic=None)
    # HBM traffic per launch of that kernel: PMC counters cannot be collected from inside this process, so the
    # figure is the one measured by scripts/pmc_traffic.sh on this same step (committed under profiles/)
    try:
        import glob
        tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
        tk = json.load(open(tf))["kernels"].get(dom)
        if tk:
            roof["traffic"] = tk["traffic_bytes_per_launch"]
            roof["traffic_source"] = os.path.relpath(tf, ROOT)
    except (IndexError, OSError, KeyError, ValueError):
        pass
    roof.update(kernel=dom, avg_launch_us=round(avg_s * 1e6, 2), launches_per_step=d["calls"] // NTR,
                algorithmic_bytes_per_launch=round(by_l), algorithmic_flops_per_launch=round(fl_l),
                tflops=round(fl_l / avg_s / 1e12, 2), share_of_step=round(d["ms"] / NTR / ms_step, 3))
    # The weight-gradient products of a backward pass run on a second stream beside the main one, so the durations
    # above are those of kernels that SHARE the chip (their sum exceeds the step time).  For the quality of the
    # dominant kernel by itself, the same step is traced once more with every launch on one stream.
    fp = trainer.eng.fp
    if fp.side is not None:
        side, fp.side = fp.side, None
        try:
            with ops.LaunchTrace() as tr1:
                for _ in range(NTR):
                    trainer.step()
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
                frac=round((by_l / PEAK_HBM if hbm_bound else fl_l / PEAK_BF16) / a1, 4),
                note="same step with the gradient stream off (TAD_GRAD_STREAM=0): no other kernel runs beside it")
    kernels = {k: dict(launches_per_step=v["calls"] // NTR, ms_per_step=round(v["ms"] / NTR, 3),
                       avg_us=round(v["ms"] / v["calls"] * 1e3, 1),
                       tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
                       hbm_GBps=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1))
               for k, v in sorted(kstat.items(), key=lambda kv: -kv[1]["ms"])[:12]}

    # ---- sampler: 1k sequences per GPU, 100 reverse steps
    sampler = None
    if not a.no_sampler:
        Bs = B_SAMPLE * world
        tgrid = torch.linspace(0, 1, T, device=dev).view(1, -1, 1).repeat(Bs, 1, 1)
        ti_s, cond_s = time_info[:Bs], cond[:Bs]
        for _ in range(1):
            DIFF.sample(tgrid, Bs, T, K, STEPS_DIFF, model, ti_s, cond_s, None, noise_source="device", process_group=pg)
        barrier()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            DIFF.sample(tgrid, Bs, T, K, STEPS_DIFF, model, ti_s, cond_s, None, noise_source="device", process_group=pg)
        barrier()
        ds = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([ds], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ds = float(tt.item())
        sampler = dict(metric="sampled_seqs_per_sec", value=round(Bs * reps / ds, 1), unit="seqs/s", batch_per_gpu=B_SAMPLE,
                       reverse_steps=STEPS_DIFF, ms_per_call=round(1e3 * ds / reps, 2))

    # ---- VAE: full optimizer steps of the mixed-type autoencoder on the same windows (third line)
    vae = None
    if not a.no_vae:
        from timeautodiff_amd.VAE import DeapStack, VAETrainer
        NV = 4 * B_LOCAL
        ones = torch.ones(NV, T, F, device=dev)
        dd = dict(response_train=cond[:NV], target_mask_train=ones, cond_train=None,
                  response_val=cond[NV:NV + 256], target_mask_val=ones[:256], cond_val=None)
        torch.manual_seed(4321)
        ae = DeapStack(1, 0, 1, 27, [8], F, H, LAYERS, E, 8, K)
        vt = VAETrainer(ae, dd, batch_size=B_LOCAL * world, lr=2e-4, weight_decay=1e-6, dtype=a.dtype, process_group=pg, dev=dev)
        import random
        random.seed(99)                       # same batch indices on every rank
        for _ in range(2):
            vt.train_step(0.1)
        barrier()
        t2 = time.perf_counter()
        vsteps = max(3, a.steps // 2)
        for _ in range(vsteps):
            vt.train_step(0.1)
        barrier()
        dv = time.perf_counter() - t2
        if world > 1:
            tt = torch.tensor([dv], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dv = float(tt.item())
        vae = dict(metric="vae_train_samples_per_sec", value=round(B_LOCAL * world * vsteps / dv, 1), unit="samples/s",
                   steps=vsteps, ms_per_step=round(1e3 * dv / vsteps, 3), per_gpu_batch=B_LOCAL,
                   model="DeapStack F=28 hidden 200, 2-layer GRU x2, emb 128, latent 6")

    if rank == 0:
        macs = fwd_macs_per_sample()
        out = dict(metric="denoiser_train_samples_per_sec", value=round(train_rate, 1), unit="samples/s", n_gpus=world,
                   steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype=a.dtype, data="synthetic",
                   config=dict(workload="energy_data-shaped 24-step windows (F=28: 1 cat card 8 + 27 num, K=6), "
                                        "BiRNN_score H=200 2 layers emb 128, full optimizer step, per-GPU batch 4096",
                               global_batch=B_LOCAL * world, per_gpu_batch=B_LOCAL, T=T, H=H, parallelism=f"dp{world}"),
                   model_flops_frac_of_bf16_peak=round(3 * 2 * macs * train_rate / (PEAK_BF16 * world), 4),
                   roofline=roof, kernels=kernels, sampler=sampler, vae=vae)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
