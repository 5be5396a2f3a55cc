#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: Hourglass + critic phase-2 TRAIN step
(4 critic fwd, 3 critic bwd, decoder fwd+bwd, mix, losses, Adam), synthetic 64x64x3 frames, batch 512/GPU, fp32.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see the contract in the task description):
  value     = whole-job A-images/s (N images of A per GPU per step; weak scaling, inputs resident in HBM)
  roofline  = algorithmic HBM bytes of one step (4.841 MB per A-image, fp32, chfak 1)
              / average duration of one step's HIP-graph launch, timed with HIP events on the launch stream
  cpu_baseline = the CPU oracle (a port of the reference step, oracle/hourglass_ref.py) timed on this box's
              host cores on a bounded sample of the same workload (rank 0, --gpus 1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

# dmabuf IPC only on this driver: must be in the environment BEFORE the HIP runtime initialises (RCCL / torch read it at init)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

ALGO_BYTES_PER_IMAGE = 4.841e6      # SURVEY.md section 8(d), fp32, chfak 1 (1 210 343 elements x 4 B)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


_COLLECT = None      # the side-block child collects its lines here instead of printing them


def emit(line: dict):
    """One JSON line on stdout -- or, inside the side-block child, one more entry of its result list."""
    if _COLLECT is not None:
        _COLLECT.append(line)
    else:
        print(json.dumps(line), flush=True)


def csrc_sha16():
    """sha256 (first 16 hex digits) over the HIP sources + the C ABI header: ties a committed counter summary to the kernels it measured."""
    import glob
    import hashlib
    pkg = os.path.join(REPO, "critic-guided-segmentation-of-rewarding-objects-in-first-person-views_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(pkg, "*.hip")) + glob.glob(os.path.join(pkg, "*.h")) + [os.path.join(REPO, "include", "cgs_hip.h")]):
        with open(f, "rb") as fp:
            h.update(os.path.basename(f).encode() + b"\0" + fp.read())
    return h.hexdigest()[:16]


def measured_traffic(n):
    """(HBM bytes per step, source) from the newest committed PMC summary for this batch size (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes, FETCH_SIZE doubled for the 16-B/lane streaming kernels per the gfx950 rule; tools/sq_counters.sh + .py +
    traffic_from_counters.py).  Counters cannot be collected from inside the timed run: the summary carries the hash of the kernel sources
    it was measured on, and `source` says whether that is THIS build.  (None, None) when no summary exists."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")), reverse=True):      # the newest summary first
        try:
            with open(path) as fp:
                t = json.load(fp)
            if t.get("n_images") == n:
                return t["traffic_bytes_per_step"], {"file": "profiles/" + os.path.basename(path), "csrc_sha16": t.get("csrc_sha16"),
                                                     "measured_on_this_build": t.get("csrc_sha16") == csrc_sha16()}
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=512, help="A-images (= B-images) per GPU per step")
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--fp16-mask-head", action="store_true",
                    help="--mode infer only: fp16 operands for the masker.0 GEMM (BASELINE config 4); ~1e-3 abs in the masks")
    ap.add_argument("--mode", choices=["train", "infer", "phase1", "cli-train"], default="train",
                    help="train = the headline phase-2 step (default); infer = eval-mode critic+masker (main.py:1130-1151); "
                         "phase1 = critic regression step (main.py:183-200)")
    ap.add_argument("--fp16", action="store_true",
                    help="--mode infer only: BASELINE config 4 -- fp16 activations and weights in every layer, fp32 accumulation (opt-in precision)")
    ap.add_argument("--force-pg", action="store_true",
                    help="--gpus 1 only: create a ONE-rank RCCL group and keep the data-parallel launch form (step graph -> all-reduce -> Adam): "
                         "rehearses the N > 1 code path on a 1-GPU box")
    ap.add_argument("--dp-eager-allreduce", action="store_true",
                    help="data parallel: keep the gradient all-reduce OUTSIDE the HIP graphs (step graph -> eager all-reduce -> Adam graph). "
                         "This is the default at --gpus N > 1 until an N > 1 RCCL run has exercised the captured form; on the 1-rank "
                         "--force-pg rehearsal the default is the captured form")
    ap.add_argument("--dp-graph-allreduce", action="store_true",
                    help="data parallel, N > 1: record the gradient all-reduce in the step's HIP graph (one graph launch per step) when the "
                         "trial capture + replay succeeds on EVERY rank (the decision is an all-reduced flag: parallel.collective_capturable)")
    ap.add_argument("--chfak", type=int, default=1, help="other model sizes (5 = the paper's) on the shape-generic kernels: --mode train or infer, one GPU, secondary measurement")
    ap.add_argument("--config", type=int, default=0,
                    help="5 = BASELINE config 5, a SIDE measurement: the build-defined 128x128 Hourglass (no reference counterpart) on the bf16 "
                         "kernels, batch 256: --mode train = the phase-2 training step, --mode infer = the eval-mode forward")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="headline line only: no CPU baseline and no side block (the quick form tools / tests use)")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the side block (BASELINE configs 4 and 5 and the paper's model size, measured by ONE child process after the "
                         "headline's timed region and reported under the extra key \"side\" of the same JSON line)")
    ap.add_argument("--side-timeout", type=float, default=240.0, help="seconds the parent waits for the side-block child")
    ap.add_argument("--side-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend of --gpus N / --force-pg: nccl (= RCCL over xGMI, the product path).  gloo is a REHEARSAL form for boxes "
                         "with fewer GPUs than ranks: ranks share the visible GPU(s) (local rank modulo device count) and the gradient bucket is "
                         "reduced through the host -- it exercises every world > 1 branch of this script, its numbers mean nothing")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--prime-s", type=float, default=0.4,
                    help="untimed graph replays for this many seconds right after capture (before the --warmup steps): lets the chip's "
                         "clocks settle from idle so that a short --steps/--warmup run measures the steady state; never inside the timed region")
    return ap.parse_args()


def dp_graph_arg(args):
    """--dp-eager-allreduce -> False, --dp-graph-allreduce -> True, neither -> None (parallel.resolve_dp_graph: graph form on a 1-rank
    group, eager at N > 1, CGS_DP_GRAPH overrides)."""
    if args.dp_eager_allreduce:
        return False
    return True if args.dp_graph_allreduce else None


def rccl_report(dist, dev, rank, world):
    """RCCL's own view of the job, printed by every rank on stderr and returned for the JSON line: library version, the ranks and
    devices the communicator spans (gathered through a collective, so it is what the backend saw, not what the launcher intended)."""
    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    mine = torch.tensor([rank, torch.cuda.current_device(), os.getpid()], device=dev, dtype=torch.int64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    members = [{"rank": int(t[0]), "device": int(t[1]), "pid": int(t[2])} for t in allr]
    name = torch.cuda.get_device_name(dev)
    print(f"[bench] rank {rank}: RCCL {ver} (torch backend 'nccl'), world {world}, device {torch.cuda.current_device()} = {name}; "
          f"communicator members: {members}", file=sys.stderr, flush=True)
    return {"version": ver, "members": members, "device_name": name, "nccl_debug": os.environ.get("NCCL_DEBUG")}


def allreduce_latency(eng, pg, dev, world, graph_ok, reps=200):
    """Bare latency of the step's ONE collective -- a sum all-reduce of the optimiser group's flat fp32 gradient bucket (25 661 floats =
    103 KB when the critic is live) -- so that a scaling loss is attributable: median over `reps` calls of the HIP-event time around one
    call on the launch stream, eager (includes the call's host launch gap) and as a replayed one-node HIP graph; MAX over ranks."""
    import torch.distributed as dist
    lo, cnt = eng._opt_range()
    cdev = torch.device("cpu") if dist.get_backend(pg) == "gloo" else dev
    buf = torch.zeros(cnt, device=cdev)
    stream = torch.cuda.current_stream()

    def median_us(fn):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        dist.barrier()
        torch.cuda.synchronize()
        for a, b in evs:
            a.record(stream); fn(); b.record(stream)
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in evs)
        m = torch.tensor([t[len(t) // 2] * 1e3], device=cdev, dtype=torch.float64)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        return float(m.item())

    for _ in range(10):
        dist.all_reduce(buf, group=pg)
    out = {"bytes": cnt * 4, "reps": reps, "eager_median": median_us(lambda: dist.all_reduce(buf, group=pg)), "graph_median": None}
    if graph_ok:
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            dist.all_reduce(buf, group=pg)
        for _ in range(10):
            g.replay()
        out["graph_median"] = median_us(g.replay)
    return out


def synthetic(n, rank, dev):
    """uint8 frames + targets, seeds rank*3 + {0,1,2}."""
    def gen(s):
        return torch.Generator().manual_seed(rank * 3 + s)
    A = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, generator=gen(0)).to(dev)
    B = torch.randint(0, 256, (n, 64, 64, 3), dtype=torch.uint8, generator=gen(1)).to(dev)
    Y = torch.rand(n, generator=gen(2)).to(dev)
    return A, B, Y


def g1_weights():
    import numpy as np
    raw = dict(np.load(os.path.join(REPO, "tests", "golden", "g1_weights_chfak1.npz")))
    pc = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("critic/")}
    pm = {k.split("/", 1)[1]: torch.from_numpy(v) for k, v in raw.items() if k.startswith("masker/")}
    return pc, pm


def host_cores():
    """(threads to use, affinity count, os.cpu_count(), cgroup quota or None): "all available cores" = the scheduler affinity capped by
    the container's CPU quota (cpu.max) -- more threads than the quota only oversubscribes.  CGS_CPU_THREADS overrides."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fp:
            q, per = fp.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    threads = avail if quota is None else max(1, min(avail, int(quota + 0.999)))
    if os.environ.get("CGS_CPU_THREADS"):
        threads = max(1, int(os.environ["CGS_CPU_THREADS"]))
    return threads, avail, os.cpu_count(), quota


def cpu_baseline(n, steps, dropout, warm=3):
    """Times the CPU oracle (test infrastructure, used here ONLY as the reported baseline) on the host cores: BASELINE.md section 3 --
    all available cores, `warm` warm-up + `steps` timed phase-2 steps at N = 64 (the reference's batch) and at N = n, bounded in time."""
    from oracle import hourglass_ref as orc
    threads, avail, ncpu, quota = host_cores()
    torch.set_num_threads(threads)
    print(f"[bench] cpu baseline: {threads} threads (affinity {avail}, cpu_count {ncpu}, cgroup quota {quota})", file=sys.stderr, flush=True)
    pc, pm = g1_weights()

    def leg(nn, budget_s):
        gen = torch.Generator().manual_seed(0)
        A = torch.randint(0, 256, (nn, 64, 64, 3), dtype=torch.uint8, generator=gen)
        B = torch.randint(0, 256, (nn, 64, 64, 3), dtype=torch.uint8, generator=gen)
        Y = torch.rand(nn, generator=gen)
        Pc, Pm = orc.leafify(pc), orc.leafify(pm)
        tensors = list(Pc.values()) + list(Pm.values())
        opt = orc.AdamRef(tensors)

        def one():
            for t in tensors:
                t.grad = None
            total, *_ = orc.phase2_loss(Pc, Pm, orc.u8_to_nchw(A), orc.u8_to_nchw(B), Y, p=dropout, training=True)
            total.backward()
            opt.step([t.grad for t in tensors])

        tw = time.perf_counter()
        nwarm = 0
        for _ in range(warm):
            one()
            nwarm += 1
            if time.perf_counter() - tw > budget_s / 3:
                break
        t0 = time.perf_counter()
        done = 0
        for _ in range(steps):
            one()
            done += 1
            if time.perf_counter() - t0 > budget_s:      # bounded sample
                break
        dt = (time.perf_counter() - t0) / done
        print(f"[bench] cpu baseline N={nn}: {nwarm} warm-up + {done} steps, {dt * 1e3:.1f} ms/step", file=sys.stderr, flush=True)
        return nn / dt, dt * 1e3, nwarm, done

    v64, ms64, w64, d64 = leg(64, 6.0)
    v, ms, w, d = leg(n, 18.0)
    model = ""
    try:
        with open("/proc/cpuinfo") as fp:
            for line in fp:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": v, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{d} phase-2 train steps of the same N={n} workload ({w} warm-up), PyTorch CPU fp32, dropout {dropout}",
            "ms_per_step": ms, "cpu_model": model,
            "cores_visible": {"affinity": avail, "cpu_count": ncpu, "cgroup_quota": quota},
            "n64": {"value": v64, "unit": "images/s", "ms_per_step": ms64,
                    "sample": f"{d64} steps at the reference's batch N=64 ({w64} warm-up)"}}


def cli_train_mode(args):
    """End-to-end throughput of the reference's mask-training loop as `main.py -train` runs it (Handler.segmentation_training,
    main.py:340-463): batch 32 + 32 | 64 (N = 64), per step three numpy index draws, two torch shift draws, one 128-entry index
    upload, on-device gather + roll, the fused step.  Synthetic frames; the critic is the G1 fixture, thresholds at its 40 / 60 %
    quantiles so the >= 500 asserts hold."""
    import tempfile
    import numpy as np
    from cgs_amd import cli, handler
    pc, pm = g1_weights()
    rs = np.random.RandomState(0)
    nfr = 16384
    X = rs.randint(0, 256, (nfr, 64, 64, 3)).astype(np.uint8)
    X[: nfr // 2] = (X[: nfr // 2] * 0.3).astype(np.uint8)
    Y = rs.rand(7, nfr)
    tmp = tempfile.mkdtemp(prefix="cgs_bench_")
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        a = cli.parse_args(["--model", "m", "--dropout", str(args.dropout)])
        H = handler.Handler(a)
        H.critic.load_state_dict(pc); H.masker.load_state_dict(pm)
        H.X, H.Y = X, Y
        eng = H._engine(64)
        preds = H._sweep_preds(eng, X).numpy()
        a.high_rew_thresh, a.low_rew_thresh = float(np.quantile(preds, 0.6)), float(np.quantile(preds, 0.4))
        H.segmentation_training()
    finally:
        os.chdir(cwd)
    emit({"metric": "main.py -train mask-training loop images/sec, 64x64x3, N=64 (reference batch)",
                      "value": H.train_images_per_s, "unit": "images/s", "n_gpus": 1, "higher_is_better": True, "dtype": "f32",
                      "data": "synthetic", "config": {"workload": "Handler.segmentation_training: index draws + upload + device gather/roll + "
                                                       "fused phase-2 step, N_A = N_B = 64, one epoch over the high-value set"}})


def prime_fixed(run, seconds, nominal_ms):
    """Untimed replays before the warm-up so that a short timed region sees settled clocks (as the headline's priming does): a FIXED count
    seconds / nominal_ms -- never a wall-clock decision, since under data parallelism every rank must issue the same number of steps."""
    for _ in range(int(seconds * 1e3 / nominal_ms) if seconds > 0 else 0):
        run()
    torch.cuda.synchronize()


def seeded_state(layout, seed):
    """Stand-in weights for model sizes without a committed checkpoint: U(+-1/sqrt(fan_in)) per tensor, reference key names / shapes."""
    import numpy as np
    rs = np.random.RandomState(seed)
    sd, bound = {}, 1.0
    for key, seg in layout.segs.items():
        if key.endswith(".weight"):
            bound = 1.0 / float(np.prod(seg.ref_shape[1:])) ** 0.5
        sd[key] = torch.from_numpy(rs.uniform(-bound, bound, size=seg.ref_shape).astype(np.float32))
    return sd


def model_flops(cf, neck=32):
    """(critic forward, masker forward, features.0) FLOPs per image at channel factor cf (186.6 MFLOP forward at cf 5)."""
    d, nb = [8 * cf, 8 * cf, 8 * cf, 16 * cf], neck * cf
    l0 = 18 * 3 * d[0] * 4096
    fc = l0 + 18 * (d[0] * d[1] * 1024 + d[1] * d[2] * 256 + d[2] * d[3] * 64) + 2 * (16 * d[3] * nb + nb * nb + nb)
    fm = 2 * nb * nb + 18 * ((d[3] + nb) * d[3] * 16 + (d[2] + d[3]) * d[2] * 64 + (d[1] + d[2]) * d[1] * 256 +
                             (d[0] + d[1]) * d[0] * 1024 + (3 + d[0]) * 16 * 4096 + 16 * 4096)
    return fc, fm, l0


def generic_mode(args, dev, rank):
    """chfak != 1 on the shape-generic kernels (the paper's model is chfak 5): the phase-2 training step (--mode train) or the
    eval-mode critic + masker (--mode infer), priced against the dense fp32 peak -- at 62 FLOP/B this size is compute-bound."""
    from cgs_amd import generic as gen, generic_engine, spec
    cf, n = args.chfak, args.batch
    lc, lm = spec.critic_layout(cf), spec.masker_layout(cf)
    fcf, fmf, l0 = model_flops(cf)
    A, B, Y = synthetic(n, rank, dev)
    if args.mode == "train":
        eng = generic_engine.GenericEngine(n, chfak=cf, device=dev, dropout=args.dropout, use_graph=not args.no_graph)
        eng.load_state(seeded_state(lc, 11), seeded_state(lm, 12))
        eng.phase2_step(A, B, Y)
        run = lambda: eng.phase2_step()
        # critic forward on [B|A|rep|inj], masker forward; critic data gradients (features.0's only for the mixes) and weight
        # gradients on [A|rep|inj]; masker data + weight gradients
        flops = 4 * fcf + fmf + (6 * fcf - l0) + 2 * fmf
        what = f"phase-2 step (main.py:344-463), N_A = N_B = {n}, dropout {args.dropout}, chfak {cf}: shape-generic MFMA kernels"
    elif args.fp16:
        eng = generic_engine.GenericEngine(n, chfak=cf, device=dev, dropout=args.dropout)
        eng.load_state(seeded_state(lc, 11), seeded_state(lm, 12))
        run = lambda: eng.infer(A, fp16=True)
        flops = fcf + fmf
        what = f"eval-mode critic(collect) + masker forward, chfak {cf}: fp16 activations / weights, fp32 accumulate (priced against the fp32 peak)"
    else:
        fc, fm = torch.empty(lc.total, device=dev), torch.empty(lm.total, device=dev)
        lc.flatten({k: v.to(dev) for k, v in seeded_state(lc, 11).items()}, fc)
        lm.flatten({k: v.to(dev) for k, v in seeded_state(lm, 12).items()}, fm)

        def run():
            c = gen.critic_forward(fc, lc, A, cf)
            return gen.masker_forward(fm, lm, A, [c[f"e{i}"] for i in range(5)], cf)["Z"]
        flops = fcf + fmf
        what = f"eval-mode critic(collect) + masker forward, chfak {cf}: shape-generic MFMA kernels"
    prime_fixed(run, args.prime_s, 5.5 if args.mode == "train" else 1.2)
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ach = flops * n / dt / 1e12
    emit({"metric": f"Hourglass+critic {args.mode} images/sec, 64x64x3 batch={n}, chfak={cf} (generic kernels)", "value": n / dt,
                      "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3,
                      "higher_is_better": True, "dtype": "f16 (f32 accumulate)" if (args.fp16 and args.mode == "infer") else "f32",
                      "data": "synthetic", "config": {"workload": what, "batch": n},
                      "roofline": {"bound": "mfma", "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3,
                                   "traffic": None, "flops_per_image": flops}})


def config5_mode(args, dev, pg=None, world=1, rank=0):
    """BASELINE config 5 (side line): 128x128x3 frames, batch 256 per GPU, bf16 storage / fp32 accumulate, the build-defined six-stage variant.
    --mode train: the phase-2 training step (Hourglass128.phase2_step), data parallel under --gpus N (one replica per rank, the flat gradient
    bucket all-reduced over RCCL inside the step's HIP graph; weak scaling); otherwise the eval-mode forward."""
    from cgs_amd import hourglass128
    n = 256 if args.batch == 512 else args.batch
    net = hourglass128.Hourglass128(*hourglass128.Hourglass128.seeded_state(31), device=dev,
                                    process_group=pg, force_allreduce=args.force_pg, dp_graph=dp_graph_arg(args))
    X = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(10 * rank)).to(dev)
    train = args.mode == "train"
    if train:
        B = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(10 * rank + 1)).to(dev)
        Y = torch.rand(n, generator=torch.Generator().manual_seed(10 * rank + 2)).to(dev)
        net.phase2_step(X, B, Y, use_graph=not args.no_graph)
        if pg is not None:
            print(f"[bench] rank {rank}: all-reduce inside the step graph: {net.dp_single_graph} ({net.dp_capture_note})", file=sys.stderr, flush=True)
        run = lambda: net.phase2_step()
    else:
        run = lambda: net.infer(X)

    def barrier():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    prime_fixed(run, args.prime_s, 1.0 if train else 0.2)
    for _ in range(max(args.warmup, 3)):         # (a fixed count on every rank: each step issues a collective)
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = (time.perf_counter() - t0) / args.steps
    if pg is not None:                           # max over ranks
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    el, flops = hourglass128.Hourglass128.model_cost()
    if train:
        # layer-granular model as SURVEY 8(d) builds it for the 64x64 step: 4 critic forwards + the mask forward, 3 critic backward + the mask
        # backward passes at ~2x a forward's elements / FLOPs each (data + weight gradient)
        elc, flc = hourglass128.Hourglass128.critic_cost()
        el_step, fl_step = 4 * elc + (el - elc) + 2 * (3 * elc + (el - elc)), 4 * flc + (flops - flc) + 2 * (3 * flc + (flops - flc))
        ach = 2.0 * el_step * n / dt / 1e9
        emit({"metric": "Hourglass-128 (build-defined) train images/sec, 128x128x3 batch=%d per GPU" % n, "value": world * n / dt, "unit": "images/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
                          "scaling": "weak", "dtype": "bf16 (f32 accumulate, f32 master weights)", "data": "synthetic",
                          "config": {"workload": "BASELINE config 5 as a training step: phase-2 step (4 critic fwd, 3 critic bwd, mask fwd + bwd, mix, losses, "
                                                 "Adam) of the six-stage 128x128 variant (no reference counterpart, parity unpinned), bf16 activations / "
                                                 "gradients, weight gradients on v_mfma_f32_16x16x32_bf16", "batch": n, "global_batch": n * world, "parallelism": f"dp{world}",
                                     "allreduce_in_step_graph": bool(net.dp_single_graph), "collective_backend": ("nccl" if pg is not None else None),
                                     "final_total_loss": float(net._train.losses[5])},
                          "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                       "algorithmic_elements_per_image": el_step, "flops_per_image": fl_step,
                                       "bf16_TFLOPs": fl_step * n / dt / 1e12}})
        return
    ach = 2.0 * el * n / dt / 1e9          # bf16: 2 bytes per element of the layer-granular traffic model
    emit({"metric": "Hourglass-128 (build-defined) inference images/sec, 128x128x3 batch=%d" % n, "value": n / dt, "unit": "images/s",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
                      "dtype": "bf16 (f32 accumulate)", "data": "synthetic",
                      "config": {"workload": "BASELINE config 5: eval-mode critic + mask forward of the six-stage 128x128 variant (no reference "
                                             "counterpart, parity unpinned), bf16 activations and weights, MFMA GEMM for the 1x1 pointwise layer", "batch": n},
                      "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                   "algorithmic_elements_per_image": el, "flops_per_image": flops}})


def side_mode(args, dev, world, rank):
    """Secondary measurements of the same path (not the headline metric): inference and the phase-1 step."""
    from cgs_amd import engine
    if args.mode == "cli-train":
        return cli_train_mode(args)
    n = args.batch
    eng = engine.HourglassEngine(n, device=dev, dropout=args.dropout, use_graph=not args.no_graph)
    eng.load_state(*g1_weights())
    A, B, Y = synthetic(n, rank, dev)
    if args.mode == "infer":
        run = lambda: eng.infer(A, fp16_mask_head=args.fp16_mask_head, fp16=args.fp16)
        bytes_per_img, what = 1.126e6, "eval-mode critic(collect)+masker forward, fp32 (SURVEY 8d: 0.563 MB/img at fp16 -> 1.126 MB fp32)"
        if args.fp16:
            bytes_per_img, what = 0.563e6, ("eval-mode critic(collect)+masker forward, fp16 activations and weights in every layer, fp32 "
                                           "accumulate (BASELINE config 4; SURVEY 8d: 0.563 MB/img at fp16)")
        if args.fp16_mask_head:
            # priced on what this path stores: the fp32 layer model WITHOUT the 16-channel mask-head intermediate (written + read = 2 x 16 x
            # 4096 x 4 B = 0.524 MB per image in SURVEY 8d's model) -- the one-kernel mask head keeps it in LDS
            bytes_per_img -= 2 * 16 * 4096 * 4.0
            what += ("; masker.0 GEMM with fp16 operands / fp32 accumulate, everything else fp32; byte model = SURVEY 8d fp32 layers minus "
                     "the mask-head intermediate the fused path never stores (0.602 MB/img)")
    else:
        eng.phase1_step(A, Y)
        run = lambda: eng.phase1_step()
        bytes_per_img, what = (67457 + 78529) * 4.0, "phase-1 critic regression step (critic fwd + bwd + Adam)"
    prime_fixed(run, args.prime_s, 0.25 * max(1.0, n / 2048))
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ach = bytes_per_img * n / dt / 1e9
    emit({"metric": f"Hourglass {args.mode} images/sec, 64x64x3 batch={n}", "value": n / dt, "unit": "images/s",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
                      "dtype": "f16 (f32 accumulate)" if (args.mode == "infer" and args.fp16) else
                               ("f32+f16 mask-head operands" if (args.mode == "infer" and args.fp16_mask_head) else "f32"),
                      "data": "synthetic", "config": {"workload": what, "batch": n},
                      "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                   "traffic": None}})


SIDE_RUNS = (      # (key, bench.py arguments): each is the side measurement a reader would run by hand with these flags
    ("config4_fp16_infer_batch2048", ["--mode", "infer", "--fp16", "--batch", "2048", "--steps", "50", "--warmup", "5"]),
    ("config5_train_batch256", ["--config", "5", "--mode", "train", "--steps", "50", "--warmup", "5"]),
    ("config5_infer_batch256", ["--config", "5", "--mode", "infer", "--steps", "50", "--warmup", "5"]),
    ("chfak5_train_batch512", ["--chfak", "5", "--mode", "train", "--steps", "20", "--warmup", "3"]),
)


def side_traffic(key):
    """Counter traffic (bytes per launch) of a side workload from the newest committed PMC summary that names it (collected like the
    headline's: tools/sq_counters.sh with the side run's flags, FETCH_SIZE / WRITE_SIZE in separate passes), or (None, None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_side_traffic.json")), reverse=True):
        try:
            with open(path) as fp:
                t = json.load(fp)
            if key in t:
                return t[key]["traffic_bytes_per_step"], {"file": "profiles/" + os.path.basename(path), "csrc_sha16": t[key].get("csrc_sha16"),
                                                          "measured_on_this_build": t[key].get("csrc_sha16") == csrc_sha16()}
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def side_child_main():
    """The side block's child process: started by the parent BEFORE anything there touched the GPU (so no process that holds a HIP
    context forks), it imports torch, then blocks on one line of stdin until the parent's timed region is over (EOF = the parent died:
    exit).  It then runs SIDE_RUNS one after the other in this one process and prints ONE JSON object {"side": {...}}."""
    global _COLLECT
    go = sys.stdin.readline()
    if not go.strip():
        return 3
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    side = {}
    for key, argv in SIDE_RUNS:
        old = sys.argv
        sys.argv = [old[0]] + argv + ["--no-cpu-baseline"]
        t0 = time.perf_counter()
        try:
            a = parse()
            _COLLECT = []
            if a.config == 5:
                config5_mode(a, dev)
            elif a.chfak != 1:
                generic_mode(a, dev, 0)
            else:
                side_mode(a, dev, 1, 0)
            line = _COLLECT[-1]
            if line.get("roofline") is not None and line["roofline"].get("traffic") is None:
                tr, src = side_traffic(key)
                line["roofline"]["traffic"], line["roofline"]["traffic_source"] = tr, src
            line["argv"] = " ".join(argv)
            line["wall_s"] = round(time.perf_counter() - t0, 2)
            side[key] = line
        except Exception as e:       # noqa: BLE001 -- a failing side run must not take the others (or the headline line) with it
            side[key] = {"error": f"{type(e).__name__}: {e}", "argv": " ".join(argv)}
        finally:
            _COLLECT = None
            sys.argv = old
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    print(json.dumps({"side": side}), flush=True)
    return 0


def start_side_child():
    """Popen of `python bench.py --side-child` (stdin = the go pipe, stdout = its one JSON line, stderr passed through)."""
    # (no torchrun variables, and no profiler preload: a headline run under rocprofv3 must not hand the child the tool's environment)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LD_PRELOAD") and not k.startswith(("ROCP", "ROCPROF"))}
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--side-child"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                            text=True, env=env)


def finish_side_child(child, timeout_s):
    """Sends the go line, waits (bounded) and returns the child's {"side": ...} object -- or {"error": ...}; never raises."""
    try:
        out, _ = child.communicate("go\n", timeout=timeout_s)
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        if child.returncode != 0 or not lines:
            return {"error": f"side child exited with code {child.returncode} and {len(lines)} JSON line(s)"}
        return json.loads(lines[-1])["side"]
    except subprocess.TimeoutExpired:
        child.kill()               # (the exact process this parent started)
        try:
            child.communicate(timeout=10)
        except Exception:          # noqa: BLE001
            pass
        return {"error": f"side child exceeded {timeout_s:.0f} s"}
    except Exception as e:          # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start N ranks (one per GPU) as a CHILD torchrun job and
    exit with its code.  Nothing in this parent has touched the GPU (torch.cuda.device_count() does not initialise HIP on
    this image), and the parent does not exec: it waits for the child."""
    have = torch.cuda.device_count()
    if have < args.gpus and not (args.backend == "gloo" and have > 0):      # (the gloo rehearsal shares the visible GPUs between ranks)
        print(f"[bench] --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr, flush=True)
        return 2
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.side_child:
        raise SystemExit(side_child_main())
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # the side block's child is started HERE, before this process creates its HIP context; it idles on a pipe until the headline is done
    headline = args.gpus == 1 and world == 1 and args.mode == "train" and args.chfak == 1 and args.config == 0 and not args.force_pg
    # (--no-cpu-baseline = the quick form tools and tests use: headline line only, no side block either)
    # (no torch.cuda call before the fork: /dev/kfd says whether a GPU node exists; the child reports a missing GPU itself)
    side_proc = start_side_child() if (headline and not args.no_side and not args.no_cpu_baseline and os.path.exists("/dev/kfd")) else None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.backend == "gloo":         # rehearsal: ranks share the visible GPU(s)
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    ranks_seen, backend, rccl_view = 1, None, None
    if world > 1 or args.force_pg:     # (forced: a 1-rank RCCL group, to rehearse the N>1 code path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)   # nccl == RCCL on ROCm
        pg = dist.group.WORLD
        ranks_seen, backend = dist.get_world_size(), dist.get_backend()
        if ranks_seen != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the collective backend sees {ranks_seen} ranks")
        rccl_view = rccl_report(dist, dev, rank, world) if backend == "nccl" else {"version": None, "members": None, "rehearsal_backend": backend}

    from cgs_amd import engine
    n = args.batch
    if args.config == 5:
        if world > 1 and args.mode != "train":
            raise SystemExit("--config 5 --mode infer is a one-GPU side measurement (--mode train takes --gpus N: data parallel)")
        r = config5_mode(args, dev, pg, world, rank)
        if pg is not None:
            torch.distributed.destroy_process_group()
        return r
    if args.chfak != 1:
        if args.mode not in ("train", "infer") or world > 1:
            raise SystemExit("--chfak != 1: --mode train / infer on one GPU (a secondary measurement; the headline is chfak 1)")
        return generic_mode(args, dev, rank)
    if args.mode != "train":
        return side_mode(args, dev, world, rank)
    eng = engine.HourglassEngine(n, device=dev, dropout=args.dropout, use_graph=not args.no_graph, process_group=pg,
                                 force_allreduce=args.force_pg, dp_graph=dp_graph_arg(args))
    eng.load_state(*g1_weights())
    A, B, Y = synthetic(n, rank, dev)
    eng.phase2_step(A, B, Y)            # inputs become resident; first call = eager step + graph capture
    if pg is not None:
        print(f"[bench] rank {rank}: all-reduce inside the step graph: {eng.dp_single_graph} ({eng.dp_capture_note})", file=sys.stderr, flush=True)

    def barrier():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # setup, untimed: replay the captured step until the clocks have settled from idle (a fresh box starts the 15 ms of a
    # --steps 20 run at its idle clocks otherwise).  The priming COUNT is a fixed function of --prime-s (a nominal 0.65 ms per
    # step), identical on every rank: under data parallelism each step issues a gradient all-reduce, so ranks must not decide
    # from their own wall clocks how many steps (= collectives) they run.  The parameters / optimiser state are put back to the
    # state right after the capture step afterwards, so the warm-up and timed steps train from the G1 weights, not from a
    # state a thousand steps into fitting one fixed batch.
    snap = eng.snapshot_state()
    nprime = int(-(-args.prime_s / 0.65e-3 // 50)) * 50 if args.prime_s > 0 else 0
    stream = torch.cuda.current_stream()

    def timed_region():
        """prime (untimed) -> restore -> W warm-up steps -> barrier -> EXACTLY K timed steps -> barrier; returns this rank's numbers."""
        barrier()
        primed = 0
        while primed < nprime:
            for _ in range(50):
                eng.phase2_step()
            primed += 50
            torch.cuda.synchronize()
        eng.restore_state(snap)
        if pg is not None:
            # untimed collectives on the gradient bucket itself (RCCL sets channels up lazily on the first calls of a size)
            for _ in range(5):
                eng._allreduce()
            eng.grad.zero_()
        for _ in range(args.warmup):
            eng.phase2_step()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            eng.phase2_step()
        ev1.record(stream)
        barrier()
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)       # HIP events on the launch stream (graph launches go to this stream)
        losses = eng.losses.cpu().tolist()
        print(f"[bench] rank {rank}: {args.steps} steps in {wall:.4f}s = {wall * 1e3 / args.steps:.4f} ms/step (wall), "
              f"{dev_ms / args.steps:.4f} ms/step (HIP events)" + (f", all-reduce in the step graph: {eng.dp_single_graph}" if pg is not None else ""),
              file=sys.stderr, flush=True)
        per_rank_ms = [wall * 1e3 / args.steps]
        if world > 1:
            t = torch.zeros(world, device=torch.device("cpu") if backend == "gloo" else dev, dtype=torch.float64)
            t[rank] = wall
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
            per_rank_ms = [float(x) * 1e3 / args.steps for x in t.tolist()]
            wall = float(t.max().item())                      # MAX over ranks
        return {"wall": wall, "dev_ms": dev_ms, "losses": losses, "per_rank_ms": per_rank_ms, "primed": primed}

    dp_report = None
    if pg is None:
        res = timed_region()
    else:
        # Data parallel: the first N > 1 run decides everything itself.  Unless a launch form was forced on the
        # command line, BOTH forms of the step are timed, each with the same priming / warm-up / K steps from the same snapshot -- eager
        # (step graph -> eager all-reduce -> Adam graph) first, then the form with the all-reduce recorded in the step's HIP graph when the
        # collective trial (parallel.collective_capturable) succeeded on EVERY rank -- and after each timed region the replicas' parameters,
        # Adam moments and step counter are compared bit for bit through an all-gathered 64-bit checksum.  `value` uses the faster form
        # among those whose replicas stayed bit-identical; every decision is made from all-reduced / all-gathered quantities, so all ranks
        # take the same branch.
        from cgs_amd import parallel
        forced = args.dp_eager_allreduce or args.dp_graph_allreduce
        if forced or args.no_graph:
            forms = ["graph" if eng.dp_single_graph else "eager"]
            capt_ok, capt_note = bool(eng.dp_single_graph), eng.dp_capture_note
        else:
            capt_ok, capt_note = parallel.collective_capturable(pg, dev)
            forms = ["eager"] + (["graph"] if capt_ok else [])
        results = {}
        for form in forms:
            if not args.no_graph and not forced:
                eng.set_dp_launch_form(form == "graph")
            eng.restore_state(snap)
            r = timed_region()
            same, sums = parallel.replica_checksums([eng.flat, eng.m, eng.v, eng.step_t], pg, dev)
            r["replicas_bit_identical"], r["replica_checksums"] = same, sums
            results[form] = r
        good = [f for f in forms if results[f]["replicas_bit_identical"]] or forms[:1]
        chosen = min(good, key=lambda f: results[f]["wall"])
        res = results[chosen]
        lat = allreduce_latency(eng, pg, dev, world, capt_ok and not args.no_graph)
        dp_report = {"forms_timed": {f: {"ms_per_step": results[f]["wall"] * 1e3 / args.steps, "per_rank_ms_per_step": results[f]["per_rank_ms"],
                                         "replicas_bit_identical": results[f]["replicas_bit_identical"],
                                         "final_total_loss_rank0": results[f]["losses"][5]} for f in forms},
                     "value_uses_form": chosen,
                     "choice": ("forced on the command line" if forced else
                                "the faster of the forms whose replicas stayed bit-identical" if len(forms) > 1 else
                                f"only the eager form was timed: {capt_note}"),
                     "collective_capturable": capt_ok, "collective_capturable_note": capt_note,
                     "replicas_bit_identical": res["replicas_bit_identical"],
                     "replica_checksums_params_m_v_step": res["replica_checksums"],
                     "allreduce_latency_us": lat}
        if not args.no_graph and not forced:
            eng.set_dp_launch_form(chosen == "graph")
    wall, dev_ms, losses, per_rank_ms, primed = res["wall"], res["dev_ms"], res["losses"], res["per_rank_ms"], res["primed"]
    if rank == 0:
        ms_step = wall * 1e3 / args.steps
        launch_ms = dev_ms / args.steps
        achieved = ALGO_BYTES_PER_IMAGE * n / (launch_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(n)
        out = {
            "metric": f"Hourglass+critic train images/sec, 64x64x3 batch={n}",
            "value": n * world * args.steps / wall,
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "phase-2 train step (4 critic fwd, 3 critic bwd, decoder fwd+bwd, mix, losses, Adam), "
                                   "synthetic 64x64x3 uint8 frames resident in HBM, chfak=1, neck=32",
                       "contrastive_batchsize": n // 2, "N_A": n, "N_B": n, "global_batch": n * world,
                       "dropout": args.dropout, "lfak": 5, "L1": 0.5, "inject": True, "live": True,
                       "parallelism": f"dp{world}", "hip_graph": not args.no_graph, "priming_steps_untimed": primed,
                       "ranks_seen_by_collective_backend": ranks_seen, "collective_backend": backend,
                       "per_rank_ms_per_step": per_rank_ms, "allreduce_in_step_graph": bool(getattr(eng, "dp_single_graph", False)),
                       "allreduce_launch_form_note": getattr(eng, "dp_capture_note", None) if pg is not None else None,
                       "rccl": rccl_view,
                       "replicas_bit_identical": (dp_report["replicas_bit_identical"] if dp_report is not None else None),
                       "dp": dp_report,
                       "gradient_allreduce": (("one flat fp32 bucket (25 661 floats) per step, " +
                                               ("recorded in the step's HIP graph (one graph launch per step)" if eng.dp_single_graph
                                                else f"eager between the two step graphs ({eng.dp_capture_note})")) if pg is not None else None)},
            # achieved / frac: ALGORITHMIC bytes of SURVEY 8(d)'s layer-granular model per second (the contract's definition), not
            # bytes that crossed the HBM pins: the fused step moves fewer (traffic), see measured_hbm_GBs and fp32_TFLOPs
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "one phase-2 step = one HIP-graph launch",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_IMAGE * n,
                         "launch_ms": launch_ms,
                         "measured_hbm_GBs": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None,
                         "fp32_TFLOPs": 86.75e6 * n / (launch_ms * 1e-3) / 1e12, "fp32_frac_of_157TF": 86.75e6 * n / (launch_ms * 1e-3) / 157.3e12},
            "final_losses": dict(zip(("critic", "replace", "inject", "l1", "l2", "total"), losses[:6])),
        }
        if side_proc is not None:
            # side block: the other BASELINE configs, each measured by the child with the flags shown in its "argv" -- after the
            # headline's timed region, before the CPU baseline takes the host cores; this process only waits
            torch.cuda.synchronize()
            out["side"] = finish_side_child(side_proc, args.side_timeout)
            side_proc = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, args.cpu_steps, args.dropout)
        print(json.dumps(out), flush=True)
    if side_proc is not None:           # (not reached on rank 0 of a headline run; any other path that started one closes it)
        side_proc.stdin.close()
        side_proc.wait()
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
