"""CPU, world_size 2 over gloo: the batch-sharded sampler run (broadcast of conditioning from
rank 0, per-rank shard, all-gather of mels) equals the single-process run on the full batch.
The denoiser is the package's explicit torch backend on the tiny config; the sampler is the
compiled plan (libdvits_hip.so host tables) executed with torch ops — the same code path as on
GPUs except for the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import UNET_CASES, unet_case
from diff_vits_amd import shard, synth
from diff_vits_amd.sampler import dpm_solver
from diff_vits_amd.unet1d.unet_1d_condition import UNet1DConditionModel


def test_shard_range_partitions():
    for G in (1, 7, 8, 64):
        for W in (1, 2, 3, 8):
            spans = [shard.shard_range(G, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _model():
    kw, sd, *_ = unet_case("tiny")
    m = UNet1DConditionModel(backend="torch", **kw).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m


def _inputs(G, T=24, L=10):
    x, cond, enc, mask = synth.make_inputs(G, 8, T, L, cond_channels=16, enc_dim=32, seed=5, ragged_mask=True)
    return tuple(map(torch.from_numpy, (x, cond, enc, mask)))


def _run_local(model):
    ns = dpm_solver.NoiseScheduleVP("discrete", betas=torch.from_numpy(synth.make_betas()))

    def run(x, cond, enc, mask):
        native = dpm_solver.NativeUNetModel(model, cond, enc, mask)     # CPU tensors -> python plan path
        fn = dpm_solver.model_wrapper(native, ns, model_type="x_start")
        return dpm_solver.DPM_Solver(fn, ns).sample(x, steps=4, order=2)
    return run


def _worker(rank, world, port, G, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        model = _model()
        x, cond, enc, mask = _inputs(G)
        lo, hi = shard.shard_range(G, world, rank)
        if rank != 0:                      # only rank 0 owns the conditioning; others receive it
            enc = torch.zeros_like(enc)
            mask = torch.zeros_like(mask)
        with torch.no_grad():
            out = shard.sharded_sample(_run_local(model), x[lo:hi].contiguous(), cond[lo:hi].contiguous(), enc, mask)
        if rank == 0:
            np.save(out_path, out.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("G", [4, 3])      # 3: ragged shards (2 + 1), padded all-gather
def test_two_rank_sharded_run_equals_single_process(tmp_path, G):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "sharded.npy")
    mp.spawn(_worker, args=(world, port, G, out_path), nprocs=world, join=True)
    model = _model()
    x, cond, enc, mask = _inputs(G)
    with torch.no_grad():
        ref = _run_local(model)(x, cond, enc, mask)
    got = np.load(out_path)
    assert got.shape == tuple(ref.shape)
    assert np.allclose(got, ref.numpy(), rtol=0, atol=2e-6)
