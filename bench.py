"""bench.py — mel-spectrogram frames/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused melspectrogram hot path over one batch of
synthetic clips already resident in HBM.  Workload at every N (weak scaling: each
rank owns its own batch, no data-path collective — SURVEY.md §8e):

    headline: B=256 clips x 10 s @ 22.05 kHz per GPU, n_fft=2048 hop=512 n_mels=128
              hann, center, constant pad, power 2, Slaney  -> T=431 frames/clip

Prints ONE JSON line (rank 0) with `roofline` and `cpu_baseline` objects.
For N>1 launch with torch.distributed.run (one rank per GPU, RCCL only for the
barrier / max-over-ranks reduction of the timing).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (B per GPU, L, sr, n_fft, hop, n_mels)
    "headline": (256, 220500, 22050, 2048, 512, 128),
    "whisper": (256, 160000, 16000, 400, 160, 80),
}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
N_ROTATE = 3                    # distinct input batches, rotated, so no step re-reads a
                                # batch that is still in the 256 MiB Infinity Cache


def synth_batch(B, L, sr, seed, device):
    """Device-side version of benchmarks/utils.py:92-115: chirp + 0.1*N(0,1), per-clip noise."""
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.linspace(0, L / sr, L, device=device, dtype=torch.float32)
    chirp = torch.sin(2 * np.pi * (100 + 2000 * t / 2) * t)
    noise = torch.randn((B, L), device=device, generator=g, dtype=torch.float32) * 0.1
    return (chirp[None, :] + noise).contiguous()


def cpu_baseline(sr, n_fft, hop, n_mels, L, budget_s=12.0):
    """The oracle's melspectrogram arithmetic (float32 scipy.fft + BLAS) on the host
    cores, on a bounded sample of the same workload."""
    from oracle import audio_oracle as ao

    # the 1-GPU box's CPU share is 16 cores even though os.cpu_count() reports the host
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    n_clips = 8
    y = np.stack([ao.bench_signal(L, sr, seed=42 + i) for i in range(n_clips)])
    ao.melspectrogram_cpu_baseline(y[:1], sr, n_fft, hop, n_mels, workers=cores)  # warm-up
    frames = 0
    t0 = time.perf_counter()
    reps = 0
    while True:
        out = ao.melspectrogram_cpu_baseline(y, sr, n_fft, hop, n_mels, workers=cores)
        frames += out.shape[0] * out.shape[2]
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 2000:
            break
    return {
        "value": frames / el, "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"{reps} x {n_clips} clips x {L} samples ({frames} frames, {el:.1f} s); "
                  "oracle arithmetic with float32 scipy.fft.rfft(workers=cores) + BLAS matmul",
    }


def main():
    ap_ = argparse.ArgumentParser()
    ap_.add_argument("--gpus", type=int, default=1)
    ap_.add_argument("--steps", type=int, default=50)
    ap_.add_argument("--warmup", type=int, default=5)
    ap_.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap_.add_argument("--batch", type=int, default=None, help="clips per GPU (default: workload's)")
    ap_.add_argument("--no-cpu-baseline", action="store_true")
    args = ap_.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(
            f"--gpus {args.gpus} needs one rank per GPU: launch with "
            f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=device)     # RCCL on ROCm

    import mlx_audio_primitives_amd as ap

    B, L, sr, n_fft, hop, n_mels = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    batches = [synth_batch(B, L, sr, 42 + 1000 * rank + i, device) for i in range(N_ROTATE)]
    T = 1 + L // hop

    def step(i):
        return ap.melspectrogram(batches[i % N_ROTATE], sr=sr, n_fft=n_fft, hop_length=hop,
                                 n_mels=n_mels)

    out = step(0)
    assert out.shape == (B, n_mels, T)
    for i in range(args.warmup):
        out = step(i)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()          # torch's current stream == the stream the kernels are enqueued on
    for i in range(args.steps):
        out = step(i)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)

    from mlx_audio_primitives_amd import sharding
    wall = sharding.max_over_ranks(wall, device=device)        # slowest rank decides
    dev_ms = sharding.max_over_ranks(dev_ms, device=device)

    if rank == 0:
        frames_per_step = world * B * T
        value = frames_per_step * args.steps / wall
        # dominant (only) kernel of a step: the fused mel kernel, one launch per step per GPU
        launch_ms = dev_ms / args.steps
        alg_bytes = (4 * hop + 4 * n_mels) * B * T           # 4H + 4M per frame
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload)
            except Exception:
                traffic = None
        rec = {
            "metric": "mel-spectrogram frames/sec (n_fft=%d, n_mels=%d)" % (n_fft, n_mels),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {B} clips/GPU x {L} samples @ {sr} Hz, "
                                   f"n_fft={n_fft} hop={hop} n_mels={n_mels} hann center constant "
                                   f"power=2 -> {T} frames/clip",
                       "clips_per_gpu": B, "frames_per_clip": T, "parallelism": f"clip-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "fused mel (pad+frame+window+rfft+|.|^2+mel)",
                         "kernel_ms": launch_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(sr, n_fft, hop, n_mels, L)
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
