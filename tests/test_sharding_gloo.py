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


def _key(x: float) -> int:
    """csrc ap_fkey: order-preserving unsigned key of a float, as the int32 the tensor stores."""
    u = int(np.array([x], np.float32).view(np.uint32)[0])
    k = (u | 0x80000000) if not (u & 0x80000000) else (~u & 0xFFFFFFFF)
    return k - (1 << 32) if k >= (1 << 31) else k


def _key_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from mlx_audio_primitives_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = []
        # (rank 0 value, rank 1 value): positive keys are NEGATIVE int32s, so a signed MAX would
        # pick the wrong rank in the mixed-sign case
        for vals in ((3.0, 100.0), (250.5, 1e-7), (-2.0, -1.0), (-5.0, 0.25), (float("-inf"), 7.0)):
            key = torch.tensor([_key(vals[rank])], dtype=torch.int32)
            sharding.global_max_key(key)
            res.append(int(key[0]))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_global_max_key_two_ranks():
    """The 4-byte MAX all-reduce mfcc(group=...) puts between the mel kernel and the dB + DCT kernel."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_key_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_key(v) for v in (100.0, 250.5, -1.0, 0.25, 7.0)]
    for rank, res in results:
        assert res == want, (rank, res, want)
    # no process group: a no-op
    from mlx_audio_primitives_amd import sharding
    k = torch.tensor([_key(4.0)], dtype=torch.int32)
    assert int(sharding.global_max_key(k)[0]) == _key(4.0)


def _run_bench(*argv, timeout=600):
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_launches_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no torch.distributed.run around it): the parent spawns the
    two ranks, they rendezvous (gloo + a no-op step here; RCCL + the mel kernel on GPUs), rank 0's
    single JSON line comes back through the parent, exit code 0."""
    import json
    r = _run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]     # gloo logs its rendezvous on stdout
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["scaling"] == "weak" and rec["unit"] == "frames/s" and rec["higher_is_better"] is True
    assert rec["config"]["parallelism"] == "clip-sharded x2"
    assert rec["value"] > 0 and "stub" in rec["data"]


def test_bench_cfg5_workload_and_separate_gather():
    """BASELINE config 5 as a bench workload (512 clips x 30 s @16 kHz per GPU, Whisper mel parameters): two gloo
    ranks with the no-op step; the final all-gather of the outputs is timed on its own and never enters `value`."""
    import json
    r = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-step", "--workload", "cfg5")
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["config"]["clips_per_gpu"] == 512 and rec["config"]["frames_per_clip"] == 3001
    assert "n_fft=400 hop=160 n_mels=80" in rec["config"]["workload"]
    assert rec["gather"]["ms"] > 0 and "not in `value`" in rec["gather"]["note"]
    assert abs(rec["value"] - 2 * 512 * 3001 * 2 / (rec["ms_per_step"] * 2 * 1e-3)) / rec["value"] < 1e-6


def test_bench_launcher_reports_a_failed_rank():
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-step", "--stub-fail-rank", "1",
                   timeout=300)
    assert r.returncode != 0
    assert "rank 1 exited with 3" in r.stderr


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
