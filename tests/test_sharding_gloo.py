"""CPU, world_size 2 over gloo: the N>1 path of bench.py / config 5 — contiguous clip
sharding with no data-path collective, the optional final gather, the 4-byte global-max
all-reduce that a sharded power_to_db(top_db) needs, and the max-over-ranks timing."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_clips, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from mlx_audio_primitives_amd import sharding
    from oracle import audio_oracle as ao

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        batch = rng.standard_normal((n_clips, 4000)).astype(np.float32)   # same on every rank
        local = sharding.shard_clips(torch.from_numpy(batch))
        lo, hi = sharding.shard_range(n_clips, rank, world)
        assert local.shape[0] == hi - lo
        # per-rank feature extraction (CPU oracle stands in for the per-GPU kernels here)
        mel_local = torch.from_numpy(ao.melspectrogram(local.numpy(), sr=16000, n_fft=400,
                                                       hop_length=160, n_mels=40))
        gathered = sharding.gather_clips(mel_local, n_clips)
        # sharded power_to_db with the global clip == unsharded power_to_db
        db_local = 10.0 * torch.log10(torch.clamp(mel_local, min=1e-10))
        gmax = sharding.global_max(db_local)
        db_local = torch.maximum(db_local, gmax - 80.0)
        db_all = sharding.gather_clips(db_local, n_clips)
        slow = sharding.max_over_ranks(1.0 + rank)
        q.put((rank, gathered.numpy(), db_all.numpy(), slow))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [7, 8])
def test_two_rank_shard_gather_and_global_clip(n_clips):
    from oracle import audio_oracle as ao

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    batch = np.random.default_rng(0).standard_normal((n_clips, 4000)).astype(np.float32)
    mel = ao.melspectrogram(batch, sr=16000, n_fft=400, hop_length=160, n_mels=40)
    db = ao.power_to_db(mel)
    for rank, gathered, db_all, slow in results:
        np.testing.assert_array_equal(gathered, mel)            # batch order restored exactly
        np.testing.assert_allclose(db_all, db, rtol=1e-6, atol=1e-5)
        assert slow == 2.0


def test_shard_range_properties():
    from mlx_audio_primitives_amd.sharding import shard_range

    for n in (0, 1, 7, 8, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(4096, 3, 8) == (1536, 2048)                # config 5: 512 clips per GPU
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)
