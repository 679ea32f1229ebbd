#!/usr/bin/env python
"""Benchmark of the RNN-T loss hot path on MI355X (driver contract: one JSON line on rank 0).

A "step" is one pass of the hot path over one synthetic minibatch, following the protocol of the
reference's pytorch_binding/benchmark.py:9-50,62-70: fp32 N(0,1) logits of shape (N,T,U,V) already
resident in HBM, labels in [1,V), full lengths; timed region = log_softmax + rnnt_loss forward
(the forward computes the gradients).  Default workload = BASELINE.json configs[3] per rank:
N=16, T=1500, U=300, V=50, gather=True -- the row the reference published as 78.88 ms on an RTX
2070 Super (README.md:51).  With --gpus N every rank owns such a slice (weak scaling, global batch
16N = configs[3] at N=8) and the per-step scalar loss is summed across ranks with one RCCL
all-reduce.

Timing: [--preload-ms of streaming copies, see its help] -> W untimed warm-up steps -> barrier + synchronize ->
exactly K timed steps -> barrier + synchronize; ms_per_step = that wall time / K, MAX over ranks.  The preload is not
the benchmark step and is reported in the JSON line (`preload_ms`).  The same W/K protocol is run once more IN FRONT
of that, from an idle GPU and without the preload, and reported next to it as `ms_per_step_cold`, and three more times
behind it (`ms_per_step_repeats`, `ms_per_step_min`: the spread of the very protocol the headline ran, same W and K);
`lattice_kernel` names the lattice kernel the headline's steps launched.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--preload-ms P]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def _point_rccl_debug_at_a_file():
    """First contact with a multi-GPU node should document itself: RCCL's own account of the transports it
    chose goes to a scratch file per rank -- INIT lines only, nothing per collective -- unless the caller already asked for
    RCCL's debug output somewhere else.  HERE, before torch is imported: RCCL reads NCCL_DEBUG* when the library
    initialises, and setting them from main() was measured to be too late (no file, no lines)."""
    # (a level below INFO -- this image exports NCCL_DEBUG=VERSION -- has no channel lines to lose: raised; a caller who
    #  asked for INFO or TRACE, or named a file, keeps what they set up)
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() not in ("VERSION", "WARN") or "NCCL_DEBUG_FILE" in os.environ \
            or "--dry" in sys.argv:
        return None
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and "--rccl-group" not in sys.argv:
        return None
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"bench_rccl_{os.getpid()}_{os.environ.get('RANK', '0')}.log")
    os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT", NCCL_DEBUG_FILE=path)
    return path


RCCL_LOG = _point_rccl_debug_at_a_file()

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 achievable

CONFIGS = {
    # name: (N per GPU, T, U, V, gather, fastemit_lambda, in-place log-softmax)
    "c2": (16, 150, 40, 28, False, 0.0, False),
    "c3": (32, 150, 20, 5000, True, 0.0, False),
    "c4": (16, 1500, 300, 50, True, 0.0, False),
    "c5": (8, 1500, 300, 10000, True, 0.01, True),
}
# README.md:39,46,51 (RTX 2070 Super, ms per batch incl. log_softmax) -> utterances/s
PUBLISHED_UTT_S = {"c2": 16 / 1.79e-3, "c3": 32 / 12.35e-3, "c4": 16 / 78.88e-3, "c5": None}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-utts", type=int, default=0, help="utterances in the CPU sample (0 = auto)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="process-group backend; nccl IS RCCL on ROCm (gloo only with --dry)")
    p.add_argument("--dry", action="store_true",
                   help="CPU rehearsal of the launcher + process group + reduction + JSON line: no kernels run, "
                        "nothing is measured (tests/test_host_cpu.py)")
    p.add_argument("--rccl-group", action="store_true",
                   help="with --gpus 1 and no launcher: still create a real one-rank RCCL process group, so that the "
                        "step carries the per-step all-reduce leg of the multi-GPU path (profiles/r03_rccl_leg.txt)")
    p.add_argument("--preload-ms", type=float, default=40.0,
                   help="milliseconds of back-to-back streaming kernels (a copy of a slice of the logits into scratch, "
                        "not the benchmark step) enqueued in front of the W warm-up steps with no gap, so that the "
                        "timed steps run in the GPU's sustained-load state, as every step of a training loop does, "
                        "and not on the 4-14 ms transient that follows load onset after idle "
                        "(profiles/r03_step_series_cold_start.txt); reported as `preload_ms`, 0 disables")
    p.add_argument("--global-batch", type=int, default=0,
                   help="strong scaling: fix the GLOBAL number of utterances and shard it over the ranks "
                        "(default 0 = weak scaling, the config's per-GPU batch on every rank)")
    return p.parse_args()


def required_bytes(cfg, global_batch, rank, world, floor=False):
    """Device memory one rank's run of `cfg` needs.  floor=True: what the timed region itself allocates -- the logits, the
    log-probs (unless the log-softmax runs in place), the loss entry's workspace and gathered gradients (16 + 8 bytes per
    lattice cell + rings).  Otherwise: that plus what the secondary timings of rank 0 hold at their peak (the copy
    yardstick's scratch, a clone of the logits that requires grad, its gradient and one dense intermediate of the
    reference-style chain) -- not for the in-place configuration, which runs none of them."""
    N, T, U, V, _, _, inplace = cfg
    if global_batch:
        from warp_rnnt_amd.distributed import shard_bounds
        lo, hi = shard_bounds(global_batch, rank, world)
        N = max(hi - lo, 1)
    dense = 4 * N * T * U * V
    cells = N * T * U
    base = dense * (1 if inplace else 2) + 32 * cells + (64 << 20)
    if floor or inplace or rank != 0:
        return base
    return base + 3 * dense


def make_batch(cfg, rank, dev):
    N, T, U, V, *_ = cfg
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + 16 * rank + N)
    xs = torch.randn((N, T, U, V), dtype=torch.float32, device=dev, generator=g)
    ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev, generator=g)
    xn = torch.full((N,), T, dtype=torch.int32, device=dev)
    yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    return xs, ys, xn, yn


def cpu_baseline(cfg, utts):
    """The fp32 C restatement (oracle/, a port of the reference's arithmetic: the reference has no
    CPU path of its own and awni/transducer is not available offline) on the host cores, on a
    bounded sample of the same workload: log_softmax + loss + grads for `utts` utterances."""
    import numpy as np
    import oracle
    N, T, U, V, gather, lam, _ = cfg
    utts = max(1, min(utts, N))
    rng = np.random.RandomState(0)
    xs = rng.randn(utts, T, U, V).astype(np.float32)
    ys = rng.randint(1, V, (utts, U - 1)).astype(np.int32)
    xn = np.full((utts,), T, dtype=np.int32)
    yn = np.full((utts,), U - 1, dtype=np.int32)
    oracle.log_softmax_f32(xs[:1])  # build + warm

    def once():
        lp = oracle.log_softmax_f32(xs)
        if gather:
            lp2 = oracle.gather_f32(lp, ys, 0)
            oracle.rnnt_loss_f32(lp2, ys, xn, yn, blank=-1, fastemit_lambda=lam, scan_mode=1)
        else:
            oracle.rnnt_loss_f32(lp, ys, xn, yn, blank=0, fastemit_lambda=lam, scan_mode=1)

    reps, t0 = 0, time.perf_counter()
    while True:                      # about 10 s of CPU work, at least 2 passes, at most 200
        once()
        reps += 1
        dt = time.perf_counter() - t0
        if (dt > 10.0 and reps >= 2) or reps >= 200 or dt > 30.0:
            break
    utts_total = utts * reps
    # single-thread figure on a smaller sample (2 utterances, one pass)
    nthreads = oracle.num_threads()
    oracle.set_threads(1)
    xs1, ys1, xn1, yn1 = xs[:2], ys[:2], xn[:2], yn[:2]
    t1 = time.perf_counter()
    lp1 = oracle.log_softmax_f32(xs1)
    if gather:
        oracle.rnnt_loss_f32(oracle.gather_f32(lp1, ys1, 0), ys1, xn1, yn1, blank=-1, fastemit_lambda=lam, scan_mode=1)
    else:
        oracle.rnnt_loss_f32(lp1, ys1, xn1, yn1, blank=0, fastemit_lambda=lam, scan_mode=1)
    one_thread = len(xs1) / (time.perf_counter() - t1)
    oracle.set_threads(nthreads)
    # BASELINE.json configs[0]: N=1, T=150, U=40, V=28 through the awni-style NumPy loops (ref_transduce.py
    # itself is not available offline; oracle/transduce_np.py restates it), one host core
    from oracle import transduce_np
    rng = np.random.RandomState(1)
    lp_c1 = transduce_np.log_softmax(rng.randn(1, 150, 40, 28))
    ys_c1 = rng.randint(1, 28, (1, 39))
    c1_ms = float("inf")
    for _ in range(3):
        t2 = time.perf_counter()
        transduce_np.transduce_batch(lp_c1, ys_c1, np.array([150]), np.array([39]), blank=0)
        c1_ms = min(c1_ms, (time.perf_counter() - t2) * 1e3)
    return {"value": round(utts_total / dt, 3), "unit": "utterances/s", "cores": nthreads,
            "kind": "port", "value_1_thread": round(one_thread, 3),
            "awni_style_numpy_c1_ms": round(c1_ms, 2),
            "sample": f"{reps} passes over {utts} utterances of T={T},U={U},V={V} "
                      f"(log_softmax + gather + loss + grads), {dt:.2f} s wall, OpenMP over rows/utterances"}


def parity_of_timed_batch(lp, ys, xn, yn, lam, utts=2):
    """Gradients of the first `utts` utterances of the timed batch against the fp32 oracle (the reference's operation order)
    on the same log-probs: max and 99.9th percentile of |hip - oracle| over the (blank, label) gradient pairs, and the
    numbers that explain the max on long lattices -- how many live slots are further than 1e-4 apart, and the largest
    distance in ulp of the plane values (max(|alpha|, |beta|, |alpha + beta|)) it sits on: a gradient is exp of a
    difference of numbers of that magnitude, so one ulp of theirs (4.9e-4 at |log-likelihood| = 6e3) moves it by as much
    (oracle.grad_error_report; tests/test_gpu_baseline_sizes.py asserts the bars).  Outside the timed region;
    per-utterance results do not depend on the batch."""
    import numpy as np
    import oracle
    from warp_rnnt_amd import ops
    k = min(utts, lp.shape[0])
    sub = lp[:k].contiguous()
    c, g = ops.loss(sub, ys[:k].contiguous(), xn[:k].contiguous(), yn[:k].contiguous(), ops.IN_LOG_PROBS_DENSE,
                    ops.GRADS_GATHERED, 0, lam)
    torch.cuda.synchronize()
    lp2 = oracle.gather_f32(sub.cpu().numpy(), ys[:k].cpu().numpy(), 0)
    xn_h, yn_h = xn[:k].cpu().numpy(), yn[:k].cpu().numpy()
    ref = oracle.rnnt_loss_f32(lp2, None, xn_h, yn_h, blank=-1, fastemit_lambda=lam, scan_mode=1)
    rep = oracle.grad_error_report(g.cpu().numpy(), ref, xn_h, yn_h)
    return {"max_abs_grad_vs_oracle": rep["max_abs"], "max_abs_grad_vs_oracle_p999": rep["p999"],
            "cells_above_1e-4": rep["cells_above"], "cells_above_1e-4_frac": rep["frac_above"],
            "max_ulp_of_plane": round(rep["max_ulp_of_plane"], 3),
            "min_plane_magnitude_of_cells_above_1e-4": rep["min_plane_magnitude_above"],
            "ulp_of_max_abs_cost": float(np.spacing(np.float32(np.abs(ref["costs"]).max()))),
            "max_rel_cost_vs_oracle": float(np.abs(c.cpu().numpy() / ref["costs"] - 1).max()),
            "parity_sample": f"{k} utterances of the timed batch, gathered gradient pairs, fp32 oracle (oracle/rnnt_oracle.c)"}


def gather_roofline(lp, ys, N, T, U, V, reps):
    """The gather kernel of the loss entry alone (k_to_diagonal<true>: dense log-probs -> diagonal-major pairs),
    HIP events on the launch stream.  Two prices: the ALGORITHMIC bytes of SURVEY.md 8(d) (16 B per cell: 8 read, 8
    written) and the bytes of the 128-byte lines those reads live in -- the memory system fetches whole lines (a
    one-dword-per-line probe over the same tensor takes as long as reading all of it,
    profiles/r03_ubench_gather_variants.txt), so the second is the floor of this access pattern on this part."""
    import numpy as np
    from warp_rnnt_amd import _lib
    L = _lib.load()
    dev = lp.device
    ws = torch.empty((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def call():
        st = L.rnnt_amd_debug_gather_only(stream, ws.data_ptr(), lp.data_ptr(), ys.data_ptr(), N, T, U, V, 0)
        assert st == 0
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    cells = N * T * U
    alg = 16.0 * cells
    out = {"bound": "hbm", "kernel": "k_to_diagonal<true> (dense log-probs -> diagonal-major (blank,label) pairs)",
           "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": alg,
           "kernel_ms": round(ms, 4), "traffic": None}
    # SURVEY.md 8(d)'s DRAM-granularity floor of the access pattern: min(4V, 128) + 8 bytes per cell
    floor = (min(4.0 * V, 128.0) + 8.0) * cells
    out["survey_floor_bytes"] = floor
    out["survey_floor_frac"] = round(floor / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if cells <= 8_000_000:
        # distinct 128-byte lines that hold a blank or a label log-prob of some cell (exact, on the host)
        lab = np.zeros((N, U), dtype=np.int64)
        lab[:, :U - 1] = ys.cpu().numpy()
        row = np.arange(cells, dtype=np.int64).reshape(N, T, U) * (V * 4)
        lines = np.unique(np.concatenate([(row // 128).ravel(), ((row + lab[:, None, :] * 4) // 128).ravel()])).size
        line_bytes = lines * 128.0 + 8.0 * cells
        out["line_bytes"] = line_bytes
        out["line_rate"] = round(line_bytes / (ms * 1e-3) / 1e9, 1)
        out["line_frac"] = round(line_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            rec = json.load(f).get("c4_gather") if (N, T, U, V) == (16, 1500, 300, 50) else None
            if rec:
                out["traffic"] = rec["traffic_bytes"]
                out["traffic_source"] = ("profiles/hbm_traffic.json: " + rec["source"] +
                                         " (a committed measurement, not taken in this run)")
    except OSError:
        pass
    return out


RCCL_VIA = ("P2P/IPC", "P2P/direct pointer", "P2P/CUMEM", "P2P", "SHM", "NET")


def parse_rccl_debug(text):
    """What RCCL said about itself in its NCCL_DEBUG=INFO output (bench.py points NCCL_DEBUG_FILE at a scratch file
    before the process group exists and reads it after the probe all-reduce).  Returns {"version": "2.x.y+hip..." or None,
    "channels_via": {"P2P/IPC": n, "SHM": n, "NET/Socket": n, ...} -- how many channel connections this rank set up over each
    transport (on one xGMI node every one of them should be P2P/...: SHM or NET means the peers do not see each other's
    memory) -- and "transport": the one word summary: "P2P" / "SHM" / "NET" / "mixed" / None when no channel line was seen
    (a one-rank group has no channels; an RCCL build that words its lines differently: None, never an error)."""
    import re
    version = None
    m = re.search(r"(?:NCCL|RCCL) version[: ]+\s*([0-9][^\s]*)", text or "")
    if m:
        version = m.group(1)
    via = {}
    for m in re.finditer(r"\bvia\s+(P2P(?:/[A-Za-z ]+?)?|SHM|NET/[A-Za-z0-9_]+)(?=[/\s]|$)", text or ""):
        kind = m.group(1).strip()
        via[kind] = via.get(kind, 0) + 1
    families = {k.split("/")[0] for k in via}
    transport = None if not families else (families.pop() if len(families) == 1 else "mixed")
    return {"version": version, "channels_via": via, "transport": transport}


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: re-execute under torch.distributed.run, one
    rank per GPU of this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    return subprocess.call(cmd, env=env)


def dry_run(a, world, rank):
    """No GPU work: the launcher, the process group, the per-step scalar reduction and the JSON contract."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group(a.backend if a.backend == "gloo" else "gloo", rank=rank, world_size=world)
    N = CONFIGS[a.config][0]
    n_local = N
    if a.global_batch:
        from warp_rnnt_amd.distributed import shard_bounds
        lo, hi = shard_bounds(a.global_batch, rank, world)
        if hi - lo < 1:
            sys.exit("--global-batch must give every rank at least one utterance")
        n_local = hi - lo
    total = torch.tensor([float(rank + 1)])
    t0 = time.perf_counter()
    for _ in range(a.warmup + a.steps):
        total = torch.tensor([float(rank + 1)])
        dist.all_reduce(total)
    dist.barrier()
    dt = time.perf_counter() - t0
    # the same collectives the measured path uses for its bookkeeping: MAX / spread of the per-rank time, utterances
    ts = [torch.zeros((1,), dtype=torch.float64) for _ in range(world)]
    dist.all_gather(ts, torch.tensor([dt], dtype=torch.float64))
    owned = torch.tensor([float(n_local)])
    dist.all_reduce(owned)
    ranks = dist.get_world_size()
    # the first-contact record of the measured path, rehearsed: every rank's own account, gathered as objects
    mine = {"rank": rank, "device": "cpu (dry run)", "device_name": None, "pci_bus": None,
            "allreduce_scalar_us": round(dt / max(a.warmup + a.steps, 1) * 1e6, 1), "rccl": parse_rccl_debug("")}
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    us = [g["allreduce_scalar_us"] for g in gathered]
    if rank == 0:
        n_global = a.global_batch if a.global_batch else N * world
        emit(json.dumps({"metric": "dry run (launcher / process group / reduction only, nothing measured)",
                          "value": None, "unit": "utterances/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
                          "scaling": "strong" if a.global_batch else "weak",
                          "dry": True, "backend": "gloo", "rccl_ranks": ranks,
                          "reduced_scalar": float(total.item()),
                          "utterances_owned_by_all_ranks": int(owned.item()),
                          "ms_per_step_rank_min_max": [round(min(float(x) for x in ts) * 1e3, 3),
                                                       round(max(float(x) for x in ts) * 1e3, 3)],
                          "allreduce_scalar_us_rank_min_max": [min(us), max(us)],
                          "node": {"ranks": gathered, "rccl_version": None, "transport": None,
                                   "distinct_devices": len({g["rank"] for g in gathered})},
                          "config": {"workload": f"{a.config}: N={n_local}/rank on rank 0 (global {n_global})"}}))
    dist.destroy_process_group()


_STDOUT_FD = None


def claim_stdout():
    """The driver reads ONE JSON line from stdout.  RCCL prints its version banner to C stdout when the first
    communicator is created (buffered, so it lands after Python's own output): everything that writes to file
    descriptor 1 from here on goes to stderr, and the JSON line is written to the original descriptor at the end."""
    global _STDOUT_FD
    sys.stdout.flush()
    _STDOUT_FD = os.dup(1)
    os.dup2(2, 1)


def emit(line):
    if _STDOUT_FD is None:
        print(line, flush=True)
    else:
        os.write(_STDOUT_FD, (line + "\n").encode())


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"--gpus {a.gpus} but the launcher started {world} rank(s)")
    if a.dry:
        return dry_run(a, world, rank)
    if a.backend != "nccl":
        sys.exit("--backend gloo is a CPU rehearsal: use it with --dry")
    if local >= torch.cuda.device_count():
        sys.exit(f"--gpus {a.gpus}: rank {rank} wants cuda:{local} but this node has {torch.cuda.device_count()} GPU(s)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # Pre-flight, BEFORE anything is allocated: the device has room for this rank's share of the configuration (one clear
    # line instead of an out-of-memory traceback from the middle of the batch generator, per rank, eight times over) ...
    need = required_bytes(CONFIGS[a.config], a.global_batch, rank, world)
    floor = required_bytes(CONFIGS[a.config], a.global_batch, rank, world, floor=True)
    free, total = torch.cuda.mem_get_info(dev)
    if floor > free:
        sys.exit(f"bench.py pre-flight: rank {rank} (cuda:{local}) needs at least {floor / 2**30:.1f} GiB for --config {a.config}"
                 f"{' --global-batch ' + str(a.global_batch) if a.global_batch else ''} and has {free / 2**30:.1f} of "
                 f"{total / 2**30:.1f} GiB free -- nothing was allocated")
    skip_secondary = need > free
    if skip_secondary:
        # (a smaller or shared GPU: the timed region fits, the clones of the secondary timings may not -- leave those out
        #  instead of refusing a run that would work)
        print(f"bench.py pre-flight: rank {rank} has {free / 2**30:.1f} GiB free, the secondary timings want about "
              f"{need / 2**30:.1f}: they are left out of the line", file=sys.stderr)
    dist = None
    rccl_ranks = 1
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or a.rccl_group:
        # under torch.distributed.run (or with --rccl-group) even a 1-rank group is a real RCCL communicator
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        rccl_log = RCCL_LOG          # (set before torch was imported: _point_rccl_debug_at_a_file)
        try:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
            # ... and the communicator works: one scalar through RCCL before the batch exists (a rank that cannot reach
            # the others says so here, not after 144 GB of logits have been generated)
            probe = torch.ones((1,), device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"all-reduce over {world} rank(s) summed to {probe.item()}")
            del probe
        except Exception as e:      # noqa: BLE001  (whatever RCCL raises: report and stop, before any large allocation)
            sys.exit(f"bench.py pre-flight: rank {rank} could not set up RCCL over {world} rank(s): {type(e).__name__}: {e}")

    from warp_rnnt_amd import _build
    _build.ensure_built()              # no-op when the prebuilt library travelled with the tree
    import warp_rnnt
    from warp_rnnt_amd import ops

    cfg = CONFIGS[a.config]
    if a.global_batch:
        from warp_rnnt_amd.distributed import shard_bounds
        lo, hi = shard_bounds(a.global_batch, rank, world)
        if hi - lo < 1:
            sys.exit("--global-batch must give every rank at least one utterance")
        cfg = (hi - lo,) + cfg[1:]
    N, T, U, V, gather, lam, inplace = cfg
    xs, ys, xn, yn = make_batch(cfg, rank, dev)
    cells = N * T * U

    # HIP events around the two halves of a step, inside the timed region, on an evenly spaced sample of the steps
    # (every step up to 5 steps, 5-9 of them beyond): a recorded event is a barrier packet in the queue (3-4 us of
    # GPU time each, measured: c2 0.040 ms/step with a sample, 0.051 with three events in every step), which is 1 %
    # of a 0.9 ms step and a quarter of a 40 us one.
    ev_stride = max(1, a.steps // 5)
    ev_steps = list(range(0, a.steps, ev_stride))
    ev_a = {i: torch.cuda.Event(enable_timing=True) for i in ev_steps}
    ev_b = {i: torch.cuda.Event(enable_timing=True) for i in ev_steps}
    ev_c = {i: torch.cuda.Event(enable_timing=True) for i in ev_steps}

    def step(i=None, last=False):
        # timed region of benchmark.py:62-70: log_softmax + loss(+grads) forward
        ev = i is not None and i in ev_a
        if ev:
            ev_a[i].record()
        lp = ops.log_softmax(xs, out=xs if inplace else None)
        if ev:
            ev_b[i].record()
        costs = warp_rnnt.rnnt_loss(lp, ys, xn, yn, gather=gather, fastemit_lambda=lam)
        if ev:
            ev_c[i].record()
        if dist is None and not last:
            return None          # single GPU: the step is benchmark.py's, nothing is summed across ranks
        total = costs.sum()
        if dist is not None:
            # the path's only exchange: one fp32 over xGMI.  Asynchronous: RCCL's stream waits for `total`,
            # the compute stream does not wait for RCCL -- the global loss is only consumed (logged) later;
            # every handle is waited on before the closing fence, inside the timed region.
            pending.append(dist.all_reduce(total, async_op=True))
        return total

    pending = []

    def fence():
        for h in pending:
            h.wait()
        pending.clear()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def preload(ms):
        # Sustained-load conditioning, disclosed in the JSON line.  A GPU that goes from idle to a streaming load slows
        # both kernels by 10-15 % from ~4 to ~14 ms after the onset, then settles (tools/step_series.py: steps 3-12 of a
        # cold process; with 30 ms of back-to-back streaming kernels in front, step 0 already runs at the settled
        # rate).  W = 5 warm-up steps are 4.5 ms: without this the timed steps sit exactly on that transient, which no
        # step of a training loop ever sees.  Not the benchmark step and not extra warm-up steps: plain copies of
        # 512 MiB (a slice of the logits where they are that large) into scratch, enqueued with no synchronisation
        # in front of the W warm-up steps.
        n_el = 1 << 27                                               # 512 MiB read + 512 MiB written per copy
        src = xs.view(-1)[:n_el] if xs.numel() >= n_el else torch.zeros((n_el,), device=dev)
        scratch = torch.empty_like(src)
        per_copy_ms = 2.0 * n_el * 4 / 5.0e12 * 1e3                  # ~0.21 ms at ~5 TB/s
        for _ in range(int(ms / per_copy_ms) + 1):
            torch.mul(src, 1.0, out=scratch)
        del scratch, src

    def timed_run(preload_ms, events):
        """[preload] -> W warm-up steps -> fence -> K timed steps -> fence.  Returns (seconds of this rank, last total)."""
        if preload_ms > 0:
            preload(preload_ms)
        for _ in range(a.warmup):
            step(last=True)          # the closing reduction is warmed up too
        fence()
        t0 = time.perf_counter()
        total = None
        for i in range(a.steps):
            total = step(i if events else None, last=(i == a.steps - 1))
        fence()
        return time.perf_counter() - t0, total

    def over_ranks(dt_local):
        """MAX over ranks (the contract), plus the spread."""
        if dist is None:
            return dt_local, dt_local, dt_local
        t = torch.tensor([dt_local], dtype=torch.float64, device=dev)
        ts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(ts, t)
        vals = [float(x.item()) for x in ts]
        return max(vals), min(vals), max(vals)

    # 1. the same protocol from an idle GPU, no preload (what rounds 1-2 and a first step after a pause measure)
    torch.cuda.synchronize()
    time.sleep(0.05)
    dt_cold, _ = timed_run(0.0, events=False)
    dt_cold = over_ranks(dt_cold)[0]
    # 2. the headline: sustained-load state
    dt, total = timed_run(a.preload_ms, events=True)
    from warp_rnnt_amd import debug as rnnt_debug
    kernel_ran = rnnt_debug.last_lattice_kernel()
    dt, dt_min, dt_max = over_ranks(dt)
    if dist is not None:
        one = torch.ones((1,), device=dev)
        dist.all_reduce(one)                      # ranks that really took part in an RCCL all-reduce
        rccl_ranks = int(one.item())
    ms_step = dt * 1e3 / a.steps
    loss_val = float(total.item())
    # 3. the spread of that very protocol: three more runs of it, same W, K, preload and fences (at K = 20
    #    the timed region is 17 ms; the mean of one such region travels better with its neighbours beside it)
    repeats = [ms_step]
    for _ in range(3):
        d, _ = timed_run(a.preload_ms, events=False)
        repeats.append(over_ranks(d)[0] * 1e3 / a.steps)
    # the only exchange of the multi-GPU path, alone: K asynchronous scalar all-reduces behind each other
    allreduce_us = None
    allreduce_us_ranks = None
    node = None
    if dist is not None:
        x = torch.ones((1,), device=dev)
        for _ in range(3):
            dist.all_reduce(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        hs = [dist.all_reduce(x, async_op=True) for _ in range(a.steps)]
        for h in hs:
            h.wait()
        torch.cuda.synchronize()
        allreduce_us = (time.perf_counter() - t1) / a.steps * 1e6
        # first-contact record: who took part, on which device, over which transport -- every rank's own account
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "device": f"cuda:{local}", "device_name": props.name,
                "pci_bus": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                               getattr(props, "pci_device_id", 0)),
                "allreduce_scalar_us": round(allreduce_us, 1)}
        text = ""
        if rccl_log:
            for cand in (rccl_log, rccl_log + "." + str(os.getpid())):
                try:
                    with open(cand) as f:
                        text += f.read()
                except OSError:
                    pass
        mine["rccl"] = parse_rccl_debug(text)
        mine["rccl"]["debug_lines_read"] = text.count("\n")     # (0: RCCL wrote nothing where it was told to)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        us = [g["allreduce_scalar_us"] for g in gathered]
        allreduce_us_ranks = [min(us), max(us)]
        try:
            v = torch.cuda.nccl.version()
            lib_version = ".".join(str(i) for i in v) if isinstance(v, tuple) else str(v)
        except Exception:      # noqa: BLE001
            lib_version = None
        transports = {g["rccl"]["transport"] for g in gathered}
        node = {"ranks": gathered, "rccl_version": next((g["rccl"]["version"] for g in gathered if g["rccl"]["version"]), None)
                or lib_version, "rccl_version_torch": lib_version,
                "transport": transports.pop() if len(transports) == 1 else "mixed",
                "distinct_devices": len({g["pci_bus"] for g in gathered})}

    # 4. the box's own yardstick, so that readings from different leases can be told apart from changes of the kernels:
    #    a plain streaming copy of the log-softmax's bytes (the logits -> a scratch tensor of the same size; in place for
    #    the config that runs its log-softmax in place), same stream, same conditioning (it runs right behind the timed
    #    runs: the GPU is in its sustained-load state).  torch's vectorised elementwise kernel.
    copy_gbs = None
    try:
        scratch = xs if inplace else torch.empty_like(xs)
        for _ in range(3):
            torch.mul(xs, 1.0, out=scratch)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        creps = max(5, min(a.steps, 20))
        c0.record()
        for _ in range(creps):
            torch.mul(xs, 1.0, out=scratch)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * xs.numel() * 4 / (c0.elapsed_time(c1) / creps * 1e-3) / 1e9
        del scratch
    except RuntimeError:      # (no room for the scratch tensor: the yardstick is left out, nothing else changes)
        copy_gbs = None

    # dominant kernel (dense log-softmax stream): average launch duration from the HIP events
    # recorded inside the timed region, on the stream the kernel runs on
    k_ms = sum(ev_a[i].elapsed_time(ev_b[i]) for i in ev_steps) / len(ev_steps)
    alg_bytes = 8.0 * V * cells      # unfused log-softmax = 8V B/cell (4V read + 4V write)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC counters: NOT measured by this process (counters need rocprofv3 around
    # it, in separate --pmc passes); the figure is the one tools/collect_profiles.sh recorded for this workload
    # and is labelled with where it came from.
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            rec = json.load(f).get(a.config, {})
            traffic = rec.get("traffic_bytes")
            if traffic is not None:
                traffic_src = ("profiles/hbm_traffic.json: " + rec.get("source", "rocprofv3 --pmc passes") +
                               " (a committed measurement of the same command, not taken in this run)")
    except OSError:
        pass
    # the rest of the step after the log-softmax = the loss entry itself: gather=True -> gather prologue +
    # alpha/beta sweeps + gradients (S2 16 B/cell + S3 32 B/cell); gather=False -> the dense
    # core, 4V+24 B/cell
    g_ms = sum(ev_b[i].elapsed_time(ev_c[i]) for i in ev_steps) / len(ev_steps)
    g_bytes = (48.0 if gather else 4.0 * V + 24.0) * cells
    g_achieved = g_bytes / (g_ms * 1e-3) / 1e9

    extras = {}
    if rank == 0 and not inplace and not skip_secondary:
        # secondary timings (outside the timed region): loss only, and the fused-from-logits entry
        lp = ops.log_softmax(xs)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        reps = max(3, min(a.steps, 20))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            warp_rnnt.rnnt_loss(lp, ys, xn, yn, gather=gather, fastemit_lambda=lam)
        e1.record()
        for _ in range(reps):
            ops.loss(xs, ys, xn, yn, ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED_DIAGONAL, 0, lam)
        e2.record()
        torch.cuda.synchronize()
        extras["loss_only_ms"] = round(e0.elapsed_time(e1) / reps, 4)
        extras["fused_from_logits_ms"] = round(e1.elapsed_time(e2) / reps, 4)
        # the UNCHANGED caller's forward step (benchmark.py:65-70 verbatim: F.log_softmax, then the loss)
        for _ in range(2):
            warp_rnnt.rnnt_loss(torch.nn.functional.log_softmax(xs, -1), ys, xn, yn, gather=gather, fastemit_lambda=lam)
        torch.cuda.synchronize()
        e3, e4 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e3.record()
        for _ in range(reps):
            warp_rnnt.rnnt_loss(torch.nn.functional.log_softmax(xs, -1), ys, xn, yn, gather=gather, fastemit_lambda=lam)
        e4.record()
        torch.cuda.synchronize()
        extras["step_torch_log_softmax_ms"] = round(e3.elapsed_time(e4) / reps, 4)
        # the same call shape with the library's lazy log_softmax: rnnt_loss recognises the handle and runs the fused
        # logits -> pairs -> loss path; the log-probs never materialise.  NOT the headline (that stays on the materialised
        # path: ops.log_softmax + rnnt_loss), reported beside it.
        from warp_rnnt_amd.functional import log_softmax as lazy_log_softmax
        for _ in range(2):
            warp_rnnt.rnnt_loss(lazy_log_softmax(xs), ys, xn, yn, gather=gather, fastemit_lambda=lam)
        torch.cuda.synchronize()
        e5, e6 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e5.record()
        for _ in range(reps):
            warp_rnnt.rnnt_loss(lazy_log_softmax(xs), ys, xn, yn, gather=gather, fastemit_lambda=lam)
        e6.record()
        torch.cuda.synchronize()
        extras["ms_per_step_lazy_log_softmax"] = round(e5.elapsed_time(e6) / reps, 4)
        extras.update(parity_of_timed_batch(lp, ys, xn, yn, lam))
        if N * T * U * V <= 2_000_000_000:
            # what a reference maintainer who links binding.cpp against this library gets
            from tools import cabi_probe
            extras.update(cabi_probe.time_entries(lp, ys, xn, yn, reps=max(3, min(reps, 10))))
            # ... and its compact sequence (binding.cpp:139-204) on a ragged batch of this shape, next to the native
            # compact entry on the same tensors
            cx = cabi_probe.ragged_compact_batch(N, T, U, V, dev)
            extras.update(cabi_probe.time_compact_entries(*cx, reps=max(3, min(reps, 10)), backward=False))
            del cx
        if gather:
            extras["roofline_gather"] = gather_roofline(lp, ys, N, T, U, V, reps)
        del lp
        # full training step (forward + backward to d/d logits), reference-style chain vs fused entry
        from warp_rnnt_amd.fused import rnnt_loss_from_logits
        xg = xs.detach().clone().requires_grad_(True)

        def chain():
            xg.grad = None
            warp_rnnt.rnnt_loss(torch.log_softmax(xg, -1), ys, xn, yn, gather=gather, fastemit_lambda=lam,
                                reduction="sum").backward()

        from warp_rnnt_amd.functional import log_softmax as native_log_softmax

        def native_chain():     # every kernel ours, the log-probs materialised (lazy=False)
            xg.grad = None
            warp_rnnt.rnnt_loss(native_log_softmax(xg, lazy=False), ys, xn, yn, gather=gather, fastemit_lambda=lam,
                                reduction="sum").backward()

        def fused():
            xg.grad = None
            rnnt_loss_from_logits(xg, ys, xn, yn, fastemit_lambda=lam, reduction="sum").backward()

        def lazy_chain():       # the reference's call shape, unchanged but for the import of log_softmax
            xg.grad = None
            warp_rnnt.rnnt_loss(native_log_softmax(xg), ys, xn, yn, gather=gather, fastemit_lambda=lam,
                                reduction="sum").backward()

        for name, fn in (("train_step_torch_log_softmax_chain_ms", chain),
                         ("train_step_native_log_softmax_chain_ms", native_chain),
                         ("train_step_lazy_log_softmax_ms", lazy_chain),
                         ("train_step_fused_logits_ms", fused)):
            fn()
            torch.cuda.synchronize()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(5):
                fn()
            a1.record()
            torch.cuda.synchronize()
            extras[name] = round(a0.elapsed_time(a1) / 5, 4)
        del xg

    if rank == 0 and inplace:
        # the in-place configuration (c5: 144 GB of logits per rank) runs none of the secondary timings -- they clone -- but
        # the lazy log_softmax needs no second tensor: the same call shape, logits read once, nothing written back.  (The
        # timed steps have turned xs into log-probabilities in place; log_softmax of those is the same tensor, so the
        # workload is the same.)
        from warp_rnnt_amd.functional import log_softmax as lazy_log_softmax
        for _ in range(2):
            warp_rnnt.rnnt_loss(lazy_log_softmax(xs), ys, xn, yn, gather=gather, fastemit_lambda=lam)
        torch.cuda.synchronize()
        e5, e6 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(3, min(a.steps, 10))
        e5.record()
        for _ in range(reps):
            warp_rnnt.rnnt_loss(lazy_log_softmax(xs), ys, xn, yn, gather=gather, fastemit_lambda=lam)
        e6.record()
        torch.cuda.synchronize()
        extras["ms_per_step_lazy_log_softmax"] = round(e5.elapsed_time(e6) / reps, 4)

    if rank == 0:
        n_global = a.global_batch if a.global_batch else world * N
        value = n_global / (ms_step * 1e-3)
        pub = PUBLISHED_UTT_S[a.config]
        out = {
            "metric": "RNN-T loss+grad throughput (log_softmax + rnnt_loss forward, grads included)",
            "value": round(value, 2),
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(ms_step, 4),
            "ms_per_step_cold": round(dt_cold * 1e3 / a.steps, 4),     # same W/K from an idle GPU, no preload
            "lattice_kernel": kernel_ran,                             # the lattice kernel the headline's steps launched
            "ms_per_step_repeats": [round(r, 4) for r in repeats],    # the headline's protocol, four times (first = headline)
            "ms_per_step_min": round(min(repeats), 4),
            "ms_per_step_rank_min_max": [round(dt_min * 1e3 / a.steps, 4), round(dt_max * 1e3 / a.steps, 4)],
            "allreduce_scalar_us": None if allreduce_us is None else round(allreduce_us, 1),
            "allreduce_scalar_us_rank_min_max": allreduce_us_ranks,
            "node": node,        # n_gpus > 1: every rank's device, PCI bus, RCCL version and the transport RCCL chose
            "higher_is_better": True,
            "scaling": "strong" if a.global_batch else "weak",
            "vs_baseline": round(value / pub, 2) if pub else None,
            "dtype": "f32",
            "data": "synthetic (N(0,1) logits, labels in [1,V), full lengths; benchmark.py:9-28 protocol)",
            "config": {"workload": f"{a.config}: N={N}/GPU (global {n_global}), T={T}, U={U}, V={V}, "
                                   f"gather={gather}, fastemit_lambda={lam}",
                       "baseline_row": "README.md:51 78.88 ms @ RTX 2070 Super" if a.config == "c4" else None,
                       "parallelism": f"batch-sharded x{world}, 1 scalar all-reduce/step" if world > 1 else "single GPU"},
            "loss_checksum": round(loss_val, 3),
            "roofline": {"bound": "hbm", "kernel": "k_lsm_regs / k_lsm_small / k_lsm_large (log-softmax over V)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # the same against what a plain copy of those bytes reaches on THIS box in THIS run (copy_gbs)
                         "frac_of_copy": None if not copy_gbs else round(achieved / copy_gbs, 4),
                         "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes": alg_bytes, "kernel_ms": round(k_ms, 4)},
            # north_star asks for ">= 60 % of HBM peak on the gather path": this is that path, priced the same way
            # (launch-to-launch HIP events around the loss entry; the sweeps inside it are latency-bound)
            "roofline_loss_path": {"bound": "hbm",
                                   "kernels": ("k_to_diagonal + k_lattice_* + k_grads" if gather else
                                               "k_to_diagonal + k_lattice_* + k_grads + k_expand"),
                                   "achieved": round(g_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(g_achieved / HBM_PEAK_GBS, 4), "traffic": None,
                                   "algorithmic_bytes": g_bytes, "kernels_ms": round(g_ms, 4)},
            "copy_gbs": None if not copy_gbs else round(copy_gbs, 1),   # plain streaming copy of the same bytes, this box
            "rccl_ranks": rccl_ranks,
            "rccl_group": dist is not None,     # True: every step ended in costs.sum() + one RCCL all-reduce
            "preload_ms": a.preload_ms,         # streaming copies enqueued in front of the warm-up steps (not steps)
        }
        out.update(extras)
        if not a.no_cpu_baseline and world == 1:   # the CPU leg is reported at N=1 only
            utts = a.cpu_utts or (16 if a.config in ("c2", "c4") else 4)
            if a.config == "c5":
                out["cpu_baseline"] = None
            else:
                out["cpu_baseline"] = cpu_baseline(cfg, utts)
        emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
