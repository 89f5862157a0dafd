"""bench.py — mel-spectrogram frames/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused melspectrogram hot path over one batch of
synthetic clips already resident in HBM.  Workload at every N (weak scaling: each
rank owns its own batch, no data-path collective — SURVEY.md §8e):

    headline: B=256 clips x 10 s @ 22.05 kHz per GPU, n_fft=2048 hop=512 n_mels=128
              hann, center, constant pad, power 2, Slaney  -> T=431 frames/clip

Prints ONE JSON line (rank 0) with `roofline` and `cpu_baseline` objects.  Before the W warm-up steps
the same step runs untimed for --ramp-seconds (default 1 s): a fresh box idles at ~600 MHz and the
power management needs a few hundred ms of load to settle on the sustained clock (2.2-2.35 GHz at
~1 350 W for this kernel, tools/diag_clock.py); without it K = 20 steps of 0.2 ms time the ramp
(0.27-0.28 ms/step) instead of the kernel.  The timed region is still exactly K steps.

N > 1: one process per GPU.  Either the caller starts the ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) or — plain `python bench.py --gpus N` — this process
spawns the N ranks itself as fresh child processes BEFORE anything here touches the GPU, relays
rank 0's JSON line and exits non-zero if any rank failed.  RCCL carries only the barrier and the
MAX-over-ranks of the timing.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (B per GPU, L, sr, n_fft, hop, n_mels)
    "headline": (256, 220500, 22050, 2048, 512, 128),
    "whisper": (256, 160000, 16000, 400, 160, 80),
    # BASELINE config 5: 4096 x 30 s @16 kHz, Whisper mel parameters, 512 clips per GPU (983 MB in, 492 MB out);
    # with N > 1 the optional all-gather of the outputs is timed separately and never enters `value`
    "cfg5": (512, 480000, 16000, 400, 160, 80),
}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: f32 vector == f32-input MFMA peak
N_ROTATE = 3                    # distinct input batches, rotated, so no step re-reads a
                                # batch that is still in the 256 MiB Infinity Cache


def parse_args(argv=None):
    ap_ = argparse.ArgumentParser()
    ap_.add_argument("--gpus", type=int, default=1)
    ap_.add_argument("--steps", type=int, default=50)
    ap_.add_argument("--warmup", type=int, default=5)
    ap_.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap_.add_argument("--batch", type=int, default=None, help="clips per GPU (default: workload's)")
    ap_.add_argument("--no-cpu-baseline", action="store_true")
    ap_.add_argument("--configs", default="all", choices=["all", "none"],
                     help="N = 1 only, AFTER the headline's timed region: steady-state time, algorithmic GB/s and "
                          "fraction of the HBM peak of every BASELINE config (tools/bench_configs.py) and the "
                          "reference's published single-clip rows, embedded as `configs` in the JSON line")
    ap_.add_argument("--no-power", action="store_true", help="skip the 2 s board-power / clock sample after the timed region")
    ap_.add_argument("--ramp-seconds", type=float, default=1.0,
                     help="untimed pre-conditioning before the W warm-up steps: the same step run back to back "
                          "for this long so the GPU leaves its low-power state (a fresh box starts at ~600 MHz "
                          "and needs some hundreds of ms of load to reach its sustained clock; 25 launches of "
                          "0.2 ms do not get it there and measure the ramp, not the kernel)")
    ap_.add_argument("--stub-step", action="store_true",
                     help="(tests) CPU ranks over gloo with a no-op step: exercises the launcher, the "
                          "barrier / max-over-ranks timing and the JSON contract without a GPU; the "
                          "line it prints is labelled and is not a measurement")
    ap_.add_argument("--stub-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    return ap_.parse_args(argv)


# ----------------------------------------------------------------------------------------------
# launcher: plain `python bench.py --gpus N` (no WORLD_SIZE) -> N fresh child ranks
# ----------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str], timeout_s: float = 3000.0) -> int:
    """Start n ranks of this script (one per GPU), wait for all, relay rank 0's stdout.

    The parent never initialises the GPU (no HIP call, no torch.cuda.is_available()): the
    children are ordinary fresh processes, nothing is exec'd over a process that holds a GPU
    context.  Returns the exit code (0 only if every rank exited 0)."""
    port = _free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + argv, env=env,
            stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=None, text=True))
    deadline = time.monotonic() + timeout_s
    rc = 0
    chunks: list[str] = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    try:
        # poll EVERY rank: a rank > 0 that dies after the RCCL rendezvous would otherwise leave rank 0 waiting in
        # its barrier until the deadline.  On the first failure the remaining (fresh child) ranks are killed.
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes) or all(c is not None for c in codes):
                break
            if time.monotonic() > deadline:
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:                    # a failed or hung rank must not leave the others behind
            if p.poll() is None:
                p.kill()
                p.wait()
    reader.join(timeout=5.0)
    out0 = "".join(c for c in chunks if c)
    for rank, p in enumerate(procs):
        if p.returncode != 0:
            print(f"bench.py: rank {rank} exited with {p.returncode}", file=sys.stderr)
            rc = rc or (p.returncode if p.returncode and p.returncode > 0 else 1)
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


# ----------------------------------------------------------------------------------------------
# workload
# ----------------------------------------------------------------------------------------------
def synth_batch(B, L, sr, seed, device):
    """Device-side version of benchmarks/utils.py:92-115: chirp + 0.1*N(0,1), per-clip noise."""
    import numpy as np
    import torch

    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.linspace(0, L / sr, L, device=device, dtype=torch.float32)
    chirp = torch.sin(2 * np.pi * (100 + 2000 * t / 2) * t)
    noise = torch.randn((B, L), device=device, generator=g, dtype=torch.float32) * 0.1
    return (chirp[None, :] + noise).contiguous()


def _cpu_worker(args):
    """One process of the N-process CPU variant: the oracle's arithmetic on its own clips, one FFT thread."""
    sr, n_fft, hop, n_mels, L, n_clips, seed, budget_s = args
    from oracle import audio_oracle as ao
    import numpy as np

    y = np.stack([ao.bench_signal(L, sr, seed=seed + i) for i in range(n_clips)])
    ao.melspectrogram_cpu_baseline(y[:1], sr, n_fft, hop, n_mels, workers=1)
    frames, reps, t0 = 0, 0, time.perf_counter()
    while True:
        out = ao.melspectrogram_cpu_baseline(y, sr, n_fft, hop, n_mels, workers=1)
        frames += out.shape[0] * out.shape[2]
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 2000:
            return frames, el


def _cpu_share():
    """Threads this job may really use: the affinity set, cut to the cgroup's CPU quota where one is set
    (a 1-GPU box shares a 256-thread host and gets 16 cores: the affinity mask still shows all 256, and
    oversubscribing them 16-fold is what handicaps a CPU number).  AP_BENCH_CORES overrides.
    Returns (cores, how)."""
    affinity = len(os.sched_getaffinity(0))
    if os.environ.get("AP_BENCH_CORES"):
        return max(1, int(os.environ["AP_BENCH_CORES"])), "AP_BENCH_CORES"
    quota = None
    try:                                              # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:                                          # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None and quota >= 1:
        return max(1, min(affinity, int(quota))), f"cgroup cpu quota {quota:g}"
    if affinity > 64:                                 # a whole host is visible but no quota is readable: the pool's
        return 16, "16 = the 1-GPU box's CPU share (no cgroup quota readable; affinity shows the whole host)"
    return affinity, "sched_getaffinity"


def cpu_baseline(sr, n_fft, hop, n_mels, L, budget_s=8.0):
    """CPU columns beside the GPU number, on a bounded sample of the same workload (rank 0, N=1), protocol of
    benchmarks/utils.py:30-63 (median of 10 after 3 warm-ups) beside the mean over the budget:
    `value` = the oracle's melspectrogram arithmetic (float32 scipy.fft + BLAS), ONE process with
    `cores` FFT / BLAS threads ("port": librosa itself is not installed in this image);
    `n_process` = the same arithmetic as one single-threaded process per core of this process's affinity
    set, every process on its own clips (clips are independent: the CPU is not handicapped by one
    process's serial sections);  `torch_stft_matmul` = torch.stft + |.|^2 + matmul on the same cores."""
    import numpy as np
    import torch

    from oracle import audio_oracle as ao

    affinity = len(os.sched_getaffinity(0))
    cores, cores_how = _cpu_share()
    n_clips = 8
    y = np.stack([ao.bench_signal(L, sr, seed=42 + i) for i in range(n_clips)])
    for _ in range(3):
        out = ao.melspectrogram_cpu_baseline(y, sr, n_fft, hop, n_mels, workers=cores)  # warm-ups
    fpc = out.shape[0] * out.shape[2]
    runs = []
    for _ in range(10):
        t0 = time.perf_counter()
        ao.melspectrogram_cpu_baseline(y, sr, n_fft, hop, n_mels, workers=cores)
        runs.append(time.perf_counter() - t0)
    frames, reps, t0 = 0, 0, time.perf_counter()
    while True:
        ao.melspectrogram_cpu_baseline(y, sr, n_fft, hop, n_mels, workers=cores)
        frames += fpc
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 2000:
            break
    rec = {
        "value": frames / el, "unit": "frames/s", "cores": cores, "kind": "port",
        "median_of_10": fpc / float(np.median(runs)),
        "os_cpu_count": os.cpu_count(), "sched_affinity": affinity, "cores_from": cores_how,
        "sample": f"{reps} x {n_clips} clips x {L} samples ({frames} frames, {el:.1f} s) + median of 10 calls after 3; "
                  "oracle arithmetic with float32 scipy.fft.rfft(workers=cores) + BLAS matmul, one process",
    }
    # N-process variant: one single-threaded process per core, 2 clips each
    try:
        import multiprocessing as mp

        nproc = max(1, cores)
        ctx = mp.get_context("spawn")                 # fresh interpreters: nothing of this process's GPU state is inherited
        with ctx.Pool(nproc) as pool:
            res = pool.map(_cpu_worker, [(sr, n_fft, hop, n_mels, L, 2, 1000 + 10 * i, budget_s / 2) for i in range(nproc)])
        rec["n_process"] = {"value": sum(f / e for f, e in res), "unit": "frames/s", "processes": nproc,
                            "sample": f"{nproc} single-threaded processes x 2 clips x {L} samples, {budget_s / 2:.0f} s each, rates summed"}
    except Exception as e:  # pragma: no cover
        rec["n_process"] = {"error": repr(e)}
    # second column: torch.stft + matmul on the CPU (torch is the reference's own cross-check oracle,
    # tests/test_torchaudio_crossval.py)
    try:
        torch.set_num_threads(cores)
        yt = torch.from_numpy(y)
        win = torch.from_numpy(ao.padded_window("hann", n_fft, n_fft))
        fb = torch.from_numpy(ao.mel_filterbank(sr, n_fft, n_mels))

        def tstep():
            S = torch.stft(yt, n_fft, hop_length=hop, window=win, center=True, pad_mode="constant",
                           return_complex=True)
            return fb @ (S.real ** 2 + S.imag ** 2)

        for _ in range(3):
            tstep()
        truns = []
        for _ in range(10):
            t0 = time.perf_counter()
            tstep()
            truns.append(time.perf_counter() - t0)
        tf, treps, t0 = 0, 0, time.perf_counter()
        while True:
            o = tstep()
            tf += o.shape[0] * o.shape[2]
            treps += 1
            tel = time.perf_counter() - t0
            if tel > budget_s / 2 or treps >= 2000:
                break
        rec["torch_stft_matmul"] = {"value": tf / tel, "unit": "frames/s", "cores": cores,
                                    "median_of_10": fpc / float(np.median(truns)),
                                    "sample": f"{treps} x {n_clips} clips ({tf} frames, {tel:.1f} s)"}
    except Exception as e:  # pragma: no cover - the first column is the contract
        rec["torch_stft_matmul"] = {"error": repr(e)}
    return rec


def sample_power(step, sync, seconds=2.0):
    """Board power and shader clock while `step` runs back to back (rocm-smi samples from a thread): the
    headline kernel is bound by the power cap, not by HBM or issue slots, so the line
    carries the evidence.  Untimed, after the timed region.  None where rocm-smi is not usable."""
    import re

    watts, mhz, done = [], [], [False]

    def poll():
        while not done[0]:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "-c"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            m = re.search(r"Power \(W\):\s*([\d.]+)", out)
            if m:
                watts.append(float(m.group(1)))
            m = re.search(r"sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)", out)
            if m:
                mhz.append(float(m.group(1)))

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for i in range(50):
            step(n + i)
        sync()
        n += 50
    el = time.perf_counter() - t0
    done[0] = True
    th.join(timeout=15.0)
    if not watts:
        return None
    w = sorted(watts[len(watts) // 3:])
    rec = {"watts": w[len(w) // 2], "watts_samples": len(watts), "ms_per_step_during_sample": el / n * 1e3,
           "board_power_cap_W": 1400}
    if mhz:
        c = sorted(mhz[len(mhz) // 3:])
        rec["sclk_GHz"] = c[len(c) // 2] / 1e3
    return rec


def _traffic_record(workload):
    """HBM bytes per launch from the rocprofv3 PMC passes kept under profiles/ (FETCH_SIZE doubled
    for 16-B/lane streaming reads as MI355X_MICROARCH.md prescribes, + WRITE_SIZE).  Counters cannot
    be collected from inside this process: the figure is the profile's, labelled as such."""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        rec = json.load(open(tpath))
    except Exception:
        return None, None
    src = rec.get("_source", {}).get(workload) or rec.get("_source", {}).get("default")
    return rec.get(workload), src or "profiles/traffic_latest.json (rocprofv3 --pmc passes, not measured in this run)"


def main(argv=None):
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (nothing has touched the GPU yet)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")

    import numpy as np  # noqa: F401
    import torch

    stub = args.stub_step
    if stub:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)     # RCCL on ROCm

    from mlx_audio_primitives_amd import sharding
    B, L, sr, n_fft, hop, n_mels = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    T = 1 + L // hop

    if stub:
        if rank == args.stub_fail_rank:
            raise SystemExit(3)            # (tests) a rank that dies before the first barrier

        def step(i):
            return None

        def sync():
            pass
    else:
        import mlx_audio_primitives_amd as ap

        batches = [synth_batch(B, L, sr, 42 + 1000 * rank + i, device) for i in range(N_ROTATE)]

        def step(i):
            return ap.melspectrogram(batches[i % N_ROTATE], sr=sr, n_fft=n_fft, hop_length=hop,
                                     n_mels=n_mels)

        sync = torch.cuda.synchronize
        out = step(0)
        assert out.shape == (B, n_mels, T)
    if not stub and args.ramp_seconds > 0:
        t_r = time.perf_counter()
        while time.perf_counter() - t_r < args.ramp_seconds:       # untimed: clock / power ramp-up
            for i in range(25):
                step(i)
            sync()
    for i in range(args.warmup):
        step(i)

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    if not stub:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    if not stub:
        ev0.record()      # torch's current stream == the stream the kernels are enqueued on
    for i in range(args.steps):
        step(i)
    if not stub:
        ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) if not stub else wall * 1e3

    wall = sharding.max_over_ranks(wall, device=device)        # slowest rank decides
    dev_ms = sharding.max_over_ranks(dev_ms, device=device)

    # optional final gather of the outputs: timed on its own, never part of `value`
    gather_ms = None
    if dist is not None and args.workload == "cfg5":
        outs = step(0) if not stub else torch.zeros((4, n_mels, 16))      # (stub: the collective itself, on gloo)
        sync()
        buf = [torch.empty_like(outs) for _ in range(world)]
        dist.all_gather(buf, outs)                   # warm-up (RCCL set-up)
        barrier()
        tg = time.perf_counter()
        dist.all_gather(buf, outs)
        barrier()
        gather_ms = sharding.max_over_ranks((time.perf_counter() - tg) * 1e3, device=device)
        del buf
    power = None
    if rank == 0 and world == 1 and not stub and not args.no_power:
        power = sample_power(step, sync)

    if rank == 0:
        frames_per_step = world * B * T
        value = frames_per_step * args.steps / wall
        # dominant (only) kernel of a step: the fused mel kernel, one launch per step per GPU
        launch_ms = dev_ms / args.steps
        alg_bytes = (4 * hop + 4 * n_mels) * B * T           # 4H + 4M per frame
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        frames_per_s_gpu = B * T / (launch_ms * 1e-3)
        traffic, traffic_source = _traffic_record(args.workload)
        F = n_fft // 2 + 1
        # the other denominators SURVEY.md §8d asks to see beside HBM (frames/s one GPU could do if
        # that resource were the only limit):
        mfma_dense = F32_PEAK_TFLOPS * 1e12 / (2.0 * n_mels * F)      # dense f32 mel_basis @ |S|^2 on MFMA
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
                         "traffic_source": traffic_source,
                         "kernel": "fused mel (pad+frame+window+rfft+|.|^2+mel)",
                         "kernel_ms": launch_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "frames_per_s_per_gpu": frames_per_s_gpu,
                         "other_bounds_frames_per_s": {
                             "hbm": HBM_PEAK_GBS * 1e9 / (4 * hop + 4 * n_mels),
                             "mfma_dense_f32": mfma_dense,
                             "note": "mfma_dense_f32 = 157.3 TF / (2 M F) flop per frame: what a DENSE "
                                     "f32-MFMA contraction would cap the fused pipeline at; this build "
                                     "contracts the banded filterbank on the VALU instead (DESIGN.md 4.1)"},
                         "frac_of_mfma_dense": frames_per_s_gpu / mfma_dense},
        }
        if gather_ms is not None:
            rec["gather"] = {"ms": gather_ms, "bytes_per_rank": 4 * B * n_mels * T,
                             "note": "all_gather of the (B, M, T) outputs over RCCL, timed separately; not in `value`"}
        if power is not None and rec.get("roofline"):
            # the kernel's own ceiling: instruction issue of the fp32 vector pipes inside the 1 400 W board cap
            # (profiles/r02_power_probe.log: a plain fp32 VALU stream alone draws 1 230-1 350 W at 0.93-1.03e12
            # wave-instructions/s).  valu_issue_frac = this kernel's ~730 plain-equivalent wave-instructions per
            # frame x frames/s / that measured ceiling.
            rec["roofline"]["secondary"] = {
                "bound": "valu_power", "watts": power.get("watts"), "sclk_GHz": power.get("sclk_GHz"),
                "board_power_cap_W": 1400,
                "valu_issue_frac": (730.0 * frames_per_s_gpu / 1.0e12) if args.workload == "headline" else None,
                "source": "rocm-smi samples during 2 s of the same step after the timed region; ceiling from "
                          "profiles/r02_power_probe.log", "samples": power.get("watts_samples")}
        if stub:
            rec["data"] = "stub (launcher test: no kernel ran, not a measurement)"
            rec["roofline"] = None
        if world == 1 and not stub and args.configs == "all":
            # every other BASELINE config + the reference's published rows, driver-timed in the same run
            try:
                del batches
                torch.cuda.empty_cache()
                from tools import bench_configs
                rec["configs"] = bench_configs.run_configs(ramp_s=0.5)
                rec["configs"]["published_rows"] = bench_configs.published_rows()
            except Exception as e:  # pragma: no cover - the headline line is the contract
                rec["configs"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and not stub:
            rec["cpu_baseline"] = cpu_baseline(sr, n_fft, hop, n_mels, L)
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
