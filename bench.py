#!/usr/bin/env python3
"""Headline benchmark: images/sec of the DeiT-tiny <- DeiT-base distillation training step (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic batch resident in HBM: mixup (device tensors) -> student forward
with feature taps -> DistillationLoss (teacher forward on a side HIP stream + LRKD terms) -> accuracy -> zero_grad ->
backward -> gradient all-reduce (N > 1) -> fused AdamW.  It is deltakd_amd.engine.train_one_epoch run over K batches.
Prints ONE JSON line on rank 0 (contract in the task statement), with "roofline" and "cpu_baseline" objects.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[3] (the one the metric is quoted on): exp/lrkd-deit-tiny.sh with a DeiT-base teacher
    "lrkd": dict(student="deit_tiny_patch16_224", teacher="deit_base_distilled_patch16_224", distillation_type="lrkd",
                 alpha=0.1, tau=3.0, lrkd_rank=64, lrkd_alpha=0.2, lrkd_beta=0.2, lrkd_gamma=0.2),
    "soft": dict(student="deit_tiny_distilled_patch16_224", teacher="deit_small_distilled_patch16_224",
                 distillation_type="soft", alpha=0.1, tau=3.0),
    "mgd": dict(student="deit_tiny_patch16_224", teacher="deit_base_distilled_patch16_224", distillation_type="mgd",
                alpha=0.5, tau=3.0, mgd_alpha=7e-5, mgd_mask_ratio=0.5),
    "none": dict(student="deit_tiny_patch16_224", teacher="deit_tiny_patch16_224", distillation_type="none", alpha=0.0, tau=1.0),
    "hard": dict(student="deit_tiny_distilled_patch16_224", teacher="deit_small_distilled_patch16_224",
                 distillation_type="hard", alpha=0.5, tau=3.0),
    "diffkd": dict(student="deit_tiny_patch16_224", teacher="deit_base_distilled_patch16_224", distillation_type="diffkd",
                   alpha=0.1, tau=3.0),
    "wasskd": dict(student="deit_small_patch16_224", teacher="vit_large_patch16_224", distillation_type="wasskd", alpha=0.1, tau=3.0,
                   wasskd_type="l1"),
}

F_FWD = {"deit_tiny_patch16_224": 2.507e9, "deit_tiny_distilled_patch16_224": 2.522e9, "deit_small_patch16_224": 9.198e9,
         "deit_small_distilled_patch16_224": 9.248e9, "deit_base_distilled_patch16_224": 35.314e9,
         "vit_large_patch16_224": 123.109e9}      # SURVEY.md section 8(d), FLOP per image


def make_args(cfg, batch, epochs=1):
    a = dict(dataset="imagenet-1k", batch_size=batch, epochs=epochs, lr=5e-4, weight_decay=1e-4, opt="adamw", opt_eps=1e-8,
             opt_betas=None, mixup=0.8, cutmix=1.0, cutmix_minmax=None, mixup_prob=1.0, mixup_switch_prob=0.5, mixup_mode="batch",
             smoothing=0.1, drop_path_rate=0.1, wasskd_type="l1", mgd_alpha=7e-5, mgd_mask_ratio=0.5, lrkd_rank=32, lrkd_alpha=0.1,
             lrkd_beta=0.1, lrkd_gamma=0.1, amp=False, rank=0, print_freq=0, current_epoch=0)
    a.update({k: v for k, v in cfg.items() if k not in ("student", "teacher")})
    return SimpleNamespace(**a)


# The student's fused kernels (the path the north star's 40 % MFMA target names) and their ALGORITHMIC bytes per launch at the headline
# shape (B = 256, N = 197, D = 192, hidden 768; every tensor the launch must touch, once -- DESIGN.md section 4):
STUDENT_KERNELS = ("mlp192_kernel<0", "mlp192_kernel<1", "attn192_fwd_kernel", "attn192_bwd_kernel", "gemm_nt_lnbwd_kernel", "gemm_tn192g_kernel")


def student_algorithmic_bytes(B, N=197, D=192, hidden=768, blocks_per_wgrad_launch=6):
    M = B * N
    return {
        # x1 f32 in, y2 bf16, pre + h bf16 [M, hidden], x2 f32 out, next block's y1 bf16 (+ weights, negligible)
        "mlp192_kernel<0>": M * (4 * D + 2 * D + 2 * 2 * hidden + 4 * D + 2 * D),
        # g f32 in / out, pre bf16 in, dH bf16 out, x1 f32 in, dF bf16 in, dF2 bf16 out
        "mlp192_kernel<1>": M * (2 * 4 * D + 2 * 2 * hidden + 4 * D + 2 * 2 * D),
        # y1 bf16 in, qkv bf16 out (the backward reads it), o bf16 out, x f32 in / out
        "attn192_fwd_kernel": M * (2 * D + 2 * 3 * D + 2 * D + 2 * 4 * D),
        # dY bf16, qkv bf16, o bf16 in; dqkv bf16 out
        "attn192_bwd_kernel": M * (2 * D + 2 * 3 * D + 2 * D + 2 * 3 * D),
        # dqkv bf16 in (K = 3 D), x f32 in, g f32 in / out, next scale-cast bf16 out
        "gemm_nt_lnbwd_kernel": M * (2 * 3 * D + 4 * D + 2 * 4 * D + 2 * D),
        # four weight gradients per block: both operands of each, once (fc1: dH | y2, fc2: dF | h, qkv: dqkv | y1, proj: dY | o)
        "gemm_tn192g_kernel": blocks_per_wgrad_launch * M * 2 * ((hidden + D) * 2 + (3 * D + D) + (D + D)),
    }


def _host_cpus():
    """(usable cpus, model string): the scheduler affinity, cut down by a cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return n, model


def cpu_baseline(cfg, batch=32, warmup=2, steps=10, budget_s=75.0):
    """The oracle (CPU restatement of the reference path, fp32, torch CPU threads) timed on this host: SURVEY.md section 8(d) --
    batch 32, 2 warm-up + 10 timed steps.  Threads = min(usable cpus, 32), set deliberately (one thread per logical cpu of a
    256-thread host oversubscribed the box's CPU share and understated the CPU by 10x in round 2).  Stops early after
    ``budget_s`` seconds of timed steps so that the default bench run stays within minutes; the sample says what was run."""
    from oracle import loss_ref, vit_ref
    cpus, model = _host_cpus()
    threads = int(os.environ.get("DKD_CPU_BASELINE_THREADS", str(max(1, min(cpus, 32)))))
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(42)
        args = make_args(cfg, batch)
        kind = cfg["distillation_type"]
        teacher = vit_ref.create_model_ref(cfg["teacher"], 1000, 0.1).eval()
        student = vit_ref.create_model_ref(cfg["student"], 1000, 0.1).train()
        loss_ref.attach_aux_ref(student, teacher, kind, args.lrkd_rank)
        for p in teacher.parameters():
            p.requires_grad = False
        crit = loss_ref.DistillationLossRef(loss_ref.SoftTargetCrossEntropyRef(), teacher, kind, args.alpha, args.tau)
        opt = torch.optim.AdamW(student.parameters(), lr=5e-4, weight_decay=1e-4)
        x = torch.randn(batch, 3, 224, 224)
        y = torch.softmax(torch.randn(batch, 1000), 1)
        draws = {"noise": torch.rand(batch, 196)} if kind == "mgd" else {}

        def step():
            if kind in ("soft", "hard"):
                out, feats = student(x), None
            else:
                out, feats = loss_ref.forward_with_features_ref(student, x)
            loss = crit(x, out, student, feats, y, args, draws)
            opt.zero_grad()
            loss.backward()
            opt.step()
        for _ in range(warmup):
            step()
        t0 = time.time()
        done = 0
        while done < steps and (done == 0 or time.time() - t0 < budget_s):
            step()
            done += 1
        dt = time.time() - t0
    finally:
        torch.set_num_threads(prev)
    return {"value": batch * done / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{warmup} warm-up + {done} timed steps of batch {batch} (same models, loss and optimizer as the GPU step; fp32 "
                      f"torch CPU, {threads} threads on {cpus} usable of {os.cpu_count()} logical cpus, {model}"
                      + ("; its LRKD targets are torch.linalg.svd of each batch's own matrices, model/loss.py:318-326 -- the quantity the GPU "
                         "leg's default mode converges to on every batch" if kind == "lrkd" else "") + ")"}


def measure_traffic_live(config, batch, timeout_s=150):
    """HBM-side bytes per launch of the NT-GEMM kernels, MEASURED during this run: two short child runs of this script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes; the program
    itself after `--`, no env / shell hop), single stream, 2 + 2 steps, before this process touches the GPU.  bytes per launch =
    (2 FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE reports half of a 16-B/lane coalesced stream, WRITE_SIZE is exact).
    -> ({kernel family: bytes}, note) or (None, why not)."""
    import collections
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    fams = ("gemm_nt_kernel<64", "gemm_nt_kernel<128", "gemm_nt256_kernel") + STUDENT_KERNELS
    raw = collections.defaultdict(dict)
    tmp = tempfile.mkdtemp(prefix="dkd_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", DKD_BENCH_PMC_CHILD="1")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", out, "-o", "r", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--gpus", "1", "--steps", "2", "--warmup", "2", "--config", config, "--batch", str(batch),
                   "--no-cpu-baseline", "--no-side-stream", "--traffic", "file", "--no-other-configs"]
            # own session: on a timeout the WHOLE group goes (rocprofv3 is a launcher; killing only it could leave the python grandchild
            # holding the GPU under the timed loop -- ADVICE round 4)
            proc = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                so, se = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                time.sleep(2.0)             # (the session's children are gone with the group; let the driver release the device)
                return None, f"TIMEOUT: the rocprofv3 --pmc {ctr} pass exceeded {timeout_s} s; its process group was killed"
            if proc.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} pass failed (rc {proc.returncode}): {(se or so)[-200:]}"
            tot, cnt = collections.Counter(), collections.Counter()
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != ctr:
                        continue
                    fam = next((k for k in fams if k in row["Kernel_Name"]), None)
                    if fam:
                        tot[fam] += float(row["Counter_Value"])
                        cnt[fam] += 1
            for k in tot:
                raw[k][ctr] = tot[k] / cnt[k]
                raw[k]["dispatches"] = cnt[k]
    except OSError as e:
        return None, f"live PMC passes failed: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {(k + ">" if k.endswith(("<64", "<128", "<0", "<1")) else k): (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024
           for k, v in raw.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    if not res:
        return None, "the PMC passes produced no rows for the NT-GEMM kernels"
    return res, {"FETCH_SIZE_KiB": {k: v.get("FETCH_SIZE") for k, v in raw.items()}, "WRITE_SIZE_KiB": {k: v.get("WRITE_SIZE") for k, v in raw.items()},
                 "dispatches": {k: v.get("dispatches") for k, v in raw.items()}}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same args>`
    as a CHILD process (the reference's launch line: exp/lrkd-deit-tiny.sh:14, tools/utils.py:52-63), relay its output (rank 0
    prints the JSON line) and return its exit code.  Nothing in this parent touches the GPU: `torch.cuda.device_count()` does not
    initialise it on this image, and a process that has must not exec another program on this pool."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("DKD_DIST_BACKEND", "nccl") == "nccl":
        missing = ", ".join(f"cuda:{i}" for i in range(have, n))
        print(f"bench.py: --gpus {n} needs {n} GPUs on this node but only {have} "
              f"{'is' if have == 1 else 'are'} visible: missing device(s) {missing}.  One rank per GPU over RCCL cannot start; "
              f"run with --gpus {max(have, 1)} here (DKD_DIST_BACKEND=gloo DKD_FORCE_DEVICE=0 rehearses the N > 1 code path "
              f"on one card).", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (_host_cpus()[0]) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="lrkd", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-stream", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other BASELINE configs (soft, mgd, wasskd)")
    ap.add_argument("--traffic", choices=("live", "file"), default="live",
                    help="roofline.traffic: measured now by two rocprofv3 --pmc child passes (N = 1 only), or read from profiles/pmc_traffic.json")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))              # plain `python bench.py --gpus N`: start the N ranks ourselves (child process)

    live_traffic, live_note = None, None
    # the live passes are skipped inside a pass (DKD_BENCH_PMC_CHILD) and when this process itself runs under a profiler (rocprofv3 /
    # rocprofiler-sdk preloads: a nested profiler would inherit their preload and output environment -- ADVICE round 4)
    profiled = bool(os.environ.get("DKD_BENCH_PMC_CHILD")) or any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ) \
        or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if a.traffic == "live" and a.gpus == 1 and "WORLD_SIZE" not in os.environ and not os.environ.get("DKD_BENCH_NO_PMC") and not profiled:
        live_traffic, live_note = measure_traffic_live(a.config, a.batch)      # (child processes; this one has not touched the GPU yet)
        # (a pass that timed out had its whole process group killed and reaped before we got here: nothing of it can hold the GPU
        # under the timed loop; the line then carries the committed traffic figure and says why -- traffic_source.why_not_live)
    elif profiled and a.traffic == "live":
        live_note = "this process runs under a profiler: live PMC passes skipped"

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("DKD_DIST_BACKEND", "nccl")        # "gloo": rehearsal of the N > 1 path on a one-GPU box
    if "DKD_FORCE_DEVICE" in os.environ:
        local = int(os.environ["DKD_FORCE_DEVICE"])
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, init_method="env://")
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}; start it as `python bench.py --gpus {a.gpus}` "
                 f"(it launches its own ranks) or with torch.distributed.run --nproc-per-node {a.gpus}")
    if backend == "nccl" and "DKD_FORCE_DEVICE" not in os.environ and local >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} needs cuda:{local} but this host shows {torch.cuda.device_count()} GPU(s)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from deltakd_amd import ops
    from deltakd_amd.ddp import DataParallel
    from deltakd_amd.engine import train_one_epoch
    from deltakd_amd.losses import DistillationLoss, call_base_loss
    from deltakd_amd.models import load_teacher_student_model
    from deltakd_amd.optim import create_optimizer
    from deltakd_amd.shims import Mixup, NativeScaler

    n_distinct = 4                       # distinct synthetic batches, rotated: the LRKD subspace solver sees changing teacher taps

    def build(config):
        """Models, optimizer, criterion and the K-batch loop of one BASELINE config -> (run(n), criterion, model, cfg)."""
        cfg = CONFIGS[config]
        args = make_args(cfg, a.batch)
        args.rank = rank
        torch.manual_seed(42)
        np.random.seed(42 + rank)
        teacher, student = load_teacher_student_model(cfg["teacher"], cfg["student"], args.drop_path_rate, args)
        student.to(dev)
        teacher.to(dev)
        optimizer = create_optimizer(args, student)
        # DKD_DP_FORCE=1 (with WORLD_SIZE=1 under torch.distributed.run): wrap the student although the world is one rank, so that a one-GPU box
        # drives the whole N > 1 code path -- broadcast, bucket all-reduces from the block-backward callback, tail sync -- through RCCL
        force = bool(os.environ.get("DKD_DP_FORCE")) and world == 1 and "RANK" in os.environ
        if force and not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(backend, init_method="env://", **({"device_id": dev} if backend == "nccl" else {}))
        model = DataParallel(student, optimizer, force=force) if (world > 1 or force) else student
        side = None if a.no_side_stream else torch.cuda.Stream()
        criterion = DistillationLoss(call_base_loss(args), teacher, cfg["distillation_type"], args.alpha, args.tau, teacher_stream=side)
        mixup_fn = Mixup(mixup_alpha=args.mixup, cutmix_alpha=args.cutmix, prob=args.mixup_prob, switch_prob=args.mixup_switch_prob,
                         label_smoothing=args.smoothing, num_classes=1000)
        scaler = NativeScaler()
        g = torch.Generator(device=dev).manual_seed(42 + rank)
        pool = [(torch.randn(a.batch, 3, 224, 224, device=dev, generator=g), torch.randint(0, 1000, (a.batch,), device=dev, generator=g))
                for _ in range(n_distinct)]
        pos = [0]

        class Loader:                    # K synthetic batches already resident in HBM (4 distinct, rotated).  Mixup writes its mix to a new
                                         # tensor (shims.Mixup, inplace=False), so the resident batches are handed out as they are
            def __init__(self, n):
                self.n = n

            def __len__(self):
                return self.n

            def __iter__(self):
                for _ in range(self.n):
                    x, y = pool[pos[0] % n_distinct]
                    pos[0] += 1
                    yield (x if mixup_fn is not None and not mixup_fn.inplace else x.clone()), y

        def run(n):
            return train_one_epoch(model, teacher, Loader(n), criterion, optimizer, scaler, None, mixup_fn, None, dev, 0, args)
        return run, criterion, model, cfg, force

    run, criterion, model, cfg, force_dp = build(a.config)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.warmup:
        run(a.warmup)
    fence()
    t0 = time.perf_counter()
    if a.steps > 1:
        run(a.steps - 1)
    # Last timed step: HIP events around every NT-GEMM launch (roofline.achieved).  It runs on ONE stream: with the teacher on its
    # side stream two kernels share the CUs and an event pair would time the sharing, not the kernel (rocprofv3's per-kernel
    # durations of a `--no-side-stream` run are the ones to compare with: profiles/).  The step still counts in `value`.
    # The student's weight gradients also stay on the compute stream in this step (DKD_NO_WGRAD_OVERLAP: dkd_block_bwd then issues
    # the grouped wgrad launch itself, inside its probe scope), so that `roofline_student` times every launch whose FLOPs it counts.
    side_saved, criterion.teacher_stream = criterion.teacher_stream, None
    wg_saved = os.environ.get("DKD_NO_WGRAD_OVERLAP")
    os.environ["DKD_NO_WGRAD_OVERLAP"] = "1"
    ev_p0, ev_p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_p0.record()
    ops.probe_begin()
    stats = run(1)
    ev_p1.record()
    criterion.teacher_stream = side_saved
    if wg_saved is None:
        del os.environ["DKD_NO_WGRAD_OVERLAP"]
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    probe_step_ms = ev_p0.elapsed_time(ev_p1)         # the single-stream, event-instrumented last step (part of `value`; steady-state steps are shorter)

    # dominant kernel = the NT-GEMM symbol with the largest summed duration in the last timed step (teacher forward is
    # ~82 % of the step's FLOPs); achieved = its algorithmic FLOPs (2 M N K per launch) / its HIP-event time.
    fam = ops.probe_end_ex()            # {family: (flops, bytes, ms, launches)}
    per = {k: (v[0], v[2], v[3]) for k, v in fam.items() if k in ops.PROBE_SYMBOLS}
    dom = max(per, key=lambda k: per[k][1]) if per else None
    flops, ms, launches = per[dom] if dom else (0.0, 0.0, 0)
    # student backward + distillation-loss path (north_star: ">= 40 % MFMA on the student backward + distill-loss path"): every
    # launch of the 12 dkd_block_bwd calls and of the fused loss kernels in the last (single-stream) step, under HIP events
    sb, sl = fam.get("student_block_bwd", (0.0, 0.0, 0.0, 0)), fam.get("loss_kernels", (0.0, 0.0, 0.0, 0))
    sf = fam.get("student_block_fwd", (0.0, 0.0, 0.0, 0))
    st_ms = sb[2] + sl[2]
    roofline_student = None
    if st_ms > 0:
        roofline_student = {
            "path": "student block backward (dkd_block_bwd x depth, weight gradients included: issued inline on the one stream) + fused "
                    "loss kernels, last timed step, HIP events",
            "flop": sb[0] + sl[0], "algorithmic_bytes": sb[1] + sl[1], "ms": st_ms, "launch_groups": sb[3] + sl[3],
            "mfma": {"achieved": (sb[0] + sl[0]) / (st_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": (sb[0] + sl[0]) / (st_ms * 1e-3) / 1e12 / 2500.0},
            "hbm": {"achieved": (sb[1] + sl[1]) / (st_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": (sb[1] + sl[1]) / (st_ms * 1e-3) / 1e9 / 8000.0},
            "block_bwd_ms": sb[2], "loss_ms": sl[2],
            "student_block_fwd": {"ms": sf[2], "tflops": sf[0] / (sf[2] * 1e-3) / 1e12 if sf[2] else 0.0,
                                  "gbps": sf[1] / (sf[2] * 1e-3) / 1e9 if sf[2] else 0.0},
        }
    if roofline_student is not None:
        alg = student_algorithmic_bytes(a.batch)
        if live_traffic:
            roofline_student["traffic"] = {k: {"measured_bytes_per_launch": live_traffic.get(k), "algorithmic_bytes_per_launch": alg[k],
                                               "ratio": (live_traffic[k] / alg[k]) if live_traffic.get(k) else None}
                                           for k in alg}
            roofline_student["traffic_how"] = ("the same two rocprofv3 --pmc passes as roofline.traffic (FETCH_SIZE / WRITE_SIZE, gfx950 correction); "
                                               "gemm_nt_lnbwd_kernel is the qkv dgrad + norm1 backward (K = 576; the fc1 dgrad lives in mlp192_kernel<1>); "
                                               "gemm_tn192g_kernel is one launch per 6 blocks")
        else:
            roofline_student["traffic"] = None
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    # HBM-side bytes per launch of the dominant kernel: NOT measured in this run (PMC counters need rocprofv3); read from the committed
    # summary of the separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command (tools_dev/collect_profiles.sh)
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if dom and live_traffic and dom in live_traffic:
        traffic = live_traffic[dom]
        traffic_source = {"measured_in_this_run": True,
                          "how": "two child runs of this command (2 + 2 steps, single stream) under rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc "
                                 "WRITE_SIZE, separate passes; bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction)",
                          "raw": live_note}
    elif dom and os.path.exists(tfile):
        tj = json.load(open(tfile))
        traffic = tj.get(a.config, {}).get(dom)
        traffic_source = {"file": "profiles/pmc_traffic.json", "collected_by": tj.get("_source", "tools_dev/collect_profiles.sh"),
                          "commit": tj.get("_commit"), "measured_in_this_run": False,
                          "why_not_live": live_note if isinstance(live_note, str) else ("--traffic file" if a.traffic == "file" else "N > 1 or disabled")}
    f_img = 3 * F_FWD[cfg["student"]] + (F_FWD[cfg["teacher"]] if cfg["distillation_type"] != "none" else 0.0)
    ips = world * a.batch * a.steps / dt
    out = {
        "metric": f"images/sec (whole node) DeiT-tiny<-DeiT-base distill, bs={a.batch}/GPU", "value": ips, "unit": "images/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "probed_last_step_ms": probe_step_ms,
        "steady_ms_per_step": (dt * 1e3 - probe_step_ms) / max(a.steps - 1, 1) if a.steps > 1 else None,
        "config": {"workload": f"exp/{a.config}-deit-tiny.sh: {cfg['student']} <- {cfg['teacher']}, {cfg['distillation_type']}, "
                               f"bs {a.batch}/GPU, 3x224x224 synthetic ({n_distinct} distinct batches rotated), 1000 classes, mixup/cutmix on, drop_path 0.1, AdamW",
                   "global_batch": world * a.batch, "parallelism": f"dp{world}",
                   "model_flops_per_image": f_img, "step_mfma_frac_of_2.5PF": ips * f_img / (world * 2.5e15),
                   "train_loss": stats.get("train_loss"),
                   # which LRKD target computation was timed (deltakd_amd.losses.LowRankTargets; the default converges every batch)
                   "lrkd_mode": criterion.lowrank.mode if cfg["distillation_type"] == "lrkd" else None},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0,
                     "traffic": traffic, "traffic_source": traffic_source, "kernel": dom, "launches": launches, "avg_launch_us": ms / max(launches, 1) * 1e3,
                     "flop_per_launch_avg": flops / max(launches, 1),
                     "other_gemm_kernels": {k: {"tflops": v[0] / (v[1] * 1e-3) / 1e12, "launches": v[2], "sum_ms": v[1]}
                                            for k, v in per.items() if k != dom}},
        "roofline_student": roofline_student,
    }
    if world > 1 or force_dp:
        # what proves the communicator: every rank's device (UUID, index, local rank), gathered THROUGH the process group that reduced
        # the gradients, and the collective library's version
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": dev.index,
                "uuid": str(getattr(props, "uuid", "")), "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        try:
            ccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                       # gloo rehearsal on a build without the binding
            ccl = None
        out["comm"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": ccl,
                       "ranks": gathered, "distinct_devices": len({g["uuid"] or (g["device_index"], g["pci_bus_id"]) for g in gathered}),
                       "overlapped_grad_buckets": getattr(model, "n_buckets", None),
                       "allreduce_calls_per_step": getattr(model, "collectives", 0) / max(1, a.steps + a.warmup),
                       "allreduce_bytes_per_step": getattr(model, "bytes_reduced", 0) / max(1, a.steps + a.warmup)}
    # BASELINE.json configs 2, 3, 5 beside the headline: soft, mgd, wasskd-L1 for 2 + 5 steps each on the same
    # box, same loop, no probe / PMC / CPU leg -- short runs: the headline's models are released first
    if world == 1 and not force_dp and not a.no_other_configs and a.config == "lrkd":
        others = {}
        del run, criterion, model
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        for name in ("soft", "mgd", "wasskd"):
            try:
                r2, c2, m2, cf2, _ = build(name)
                r2(2)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                r2(5)
                torch.cuda.synchronize()
                d2 = time.perf_counter() - t1
                fi = 3 * F_FWD[cf2["student"]] + F_FWD[cf2["teacher"]]
                others[name] = {"value": a.batch * 5 / d2, "unit": "images/s", "ms_per_step": d2 / 5 * 1e3, "steps": 5, "warmup": 2,
                                "step_mfma_frac": a.batch * 5 / d2 * fi / 2.5e15, "models": f"{cf2['student']} <- {cf2['teacher']}",
                                "model_flops_per_image_excl_loss_net": fi}
                del r2, c2, m2
                gc.collect()
                torch.cuda.empty_cache()
            except Exception as e:                  # a failing side config must not lose the headline line
                others[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out["other_configs"] = others
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:              # the CPU leg must not lose the measured GPU line
                out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": None, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"[:300]}
        print(json.dumps(out), flush=True)
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
