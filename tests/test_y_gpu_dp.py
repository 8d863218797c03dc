"""Data parallelism rehearsed on ONE GPU: two (or eight) fresh processes share the card, hold a gloo group over device tensors
and run the trainers' real entry points (DiffusionTrainer.step, VAETrainer.train_step / eval_loss, DIFF.sample) with
``process_group``; the result must equal the one-process run on the same draws.  The global batch (7, validation 5,
sampling 5) is not divisible by the number of ranks.  fp32; tolerances: 2e-5 relative on losses and gradients (the
ranks sum their halves in a different order than one process sums the whole), 1e-4 on weights after three steps."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("world,grad_stream", [(2, "0"), (2, "1"), (8, "0")])
def test_ranks_on_one_gpu_equal_one_process(tmp_path, world, grad_stream):
    """world 2: the ranks take 3 / 4 of the 7 training windows; with TAD_GRAD_STREAM=1 the weight gradients and the first
    bucket's all-reduce go through the second stream.  world 8: one rank's training shard and three ranks' validation
    and sampling shards are EMPTY (7 and 5 items over 8 ranks) -- every rank still takes part in every collective."""
    worker = os.path.join(HERE, "dp_gpu_worker.py")
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    env = dict(os.environ, TAD_GRAD_STREAM=grad_stream)
    subprocess.run([sys.executable, worker, "0", "1", "0", one], check=True, timeout=300, env=env)
    port = _port()
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), two], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = np.load(one), np.load(two)
    # denoiser: loss of every step, the reduced gradient of the first step (and its norm, the cross-rank equality
    # check a training loop would use), the weights after three steps
    assert np.max(np.abs(a["den_loss"] - b["den_loss"]) / np.abs(a["den_loss"])) < 2e-5
    assert _rel(b["den_grad"], a["den_grad"]) < 2e-5
    assert abs(float(b["den_gnorm"][0]) - float(a["den_gnorm"][0])) < 2e-5 * float(a["den_gnorm"][0])
    assert abs(float(a["den_gnorm"][0]) - float(np.linalg.norm(a["den_grad"].astype(np.float64)))) < 1e-5 * float(a["den_gnorm"][0])
    assert _rel(b["den_w"], a["den_w"]) < 1e-4
    # sampler: every rank returns the whole batch, identical to the one-process trajectory; local_only returns the slice
    assert b["sample"].shape == a["sample"].shape and _rel(b["sample"], a["sample"]) < 1e-6
    assert int(b["sample_local_rows"][0]) == (5 * 1) // world and int(a["sample_local_rows"][0]) == 5       # rank 0's slice
    # VAE: total loss, all loss parts (bce, mse, ce per column, kld sums), validation loss, gradient, weights
    assert np.max(np.abs(a["vae_loss"] - b["vae_loss"]) / np.abs(a["vae_loss"])) < 2e-5
    assert np.max(np.abs(a["vae_parts"] - b["vae_parts"]) / np.maximum(np.abs(a["vae_parts"]), 1e-6)) < 2e-5
    assert np.max(np.abs(a["vae_eval"] - b["vae_eval"]) / np.abs(a["vae_eval"])) < 2e-5
    assert _rel(b["vae_grad"], a["vae_grad"]) < 2e-5
    assert _rel(b["vae_w"], a["vae_w"]) < 1e-4


@pytest.mark.parametrize("grad_stream", ["0", "1"])
def test_one_rank_over_rccl_equals_no_group(tmp_path, grad_stream):
    """The product's backend on the one GPU a test box has: a group of ONE rank over RCCL ("nccl").  Every collective of the
    trainers and of the sampler (bucketed asynchronous all-reduce of the gradients, the loss and validation sums, the padded
    all-gather of the sampled sequences) goes through RCCL's streams and its Work.wait(), which differ from gloo's; with
    nobody to exchange with, the results must equal the run without a group."""
    worker = os.path.join(HERE, "dp_gpu_worker.py")
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "rccl.npz")
    env = dict(os.environ, TAD_GRAD_STREAM=grad_stream)
    subprocess.run([sys.executable, worker, "0", "1", "0", one], check=True, timeout=300, env=env)
    subprocess.run([sys.executable, worker, "0", "1", str(_port()), two, "nccl"], check=True, timeout=300, env=env)
    a, b = np.load(one), np.load(two)
    for k in ("den_loss", "den_grad", "den_w", "sample", "vae_loss", "vae_parts", "vae_eval", "vae_grad", "vae_w"):
        assert a[k].shape == b[k].shape and _rel(b[k], a[k]) < 1e-6, k
    assert int(b["sample_local_rows"][0]) == 5
