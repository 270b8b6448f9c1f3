#!/usr/bin/env python3
"""Benchmark of the hot path: one training step (forward + CE + backward + gradient all-reduce + clip + AdamW) of
BERT-base + ResNet-50 + the reference's attention fusion head on synthetic 224x224 RGB + 128-token batches, bs=64/GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1: either the driver starts the ranks (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
WORLD_SIZE is set, this process is one rank) or — WORLD_SIZE unset — this process starts them itself: it spawns
`torch.distributed.run` as a CHILD process before anything here touches the GPU, relays the children's output (rank 0
prints the JSON line) and exits with their status. It never re-executes itself.

Prints ONE JSON line on rank 0 (BASELINE.json metric: (image,text) pairs/sec/node). `roofline` is measured live: every
bf16 MFMA GEMM launch is timed by the kernels' own clock right after the timed region (mmsa_prof_*; HIP events on the
launch stream beside it); `forward` is the forward-only pass of the same model at the same batch (the north-star's 50 %
MFMA target is stated on the forward); `cpu_baseline` is the CPU oracle's train step (oracle/model.py) timed on the host
cores on a bounded sample (rank 0, N = 1 only). `--mode fwd` times the forward alone as the headline number.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_PAIR = 30.52  # BERT-base S=128 22.348 + ResNet-50 (no fc) 8.174, 2 FLOP/MAC
C3_FWD_GFLOP_PER_PAIR = 176.66  # BERT-large S=256 161.06 + ResNet-101 15.60
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3    # fp32 matrix / vector peak (precision="fp32": exact-fp32 kernels)
# Algorithmic HBM bytes per MFMA GEMM launch are counted by the library itself for the launches of the roofline pass
# (mmsa_prof_last_bytes: every distinct operand of a problem read once — a 3x3 implicit-GEMM gather counts each source pixel once —
# the output written once, epilogue side operands and side outputs included) and divided by the launch count of the same pass.
TRAFFIC_FILES = ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 20 warm-up, >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of exactly --steps steps each; `value` is the "
                    "FIRST region (the driver's contract), the median of all of them is reported beside it (§8(d))")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 64; 32 for --model large)")
    ap.add_argument("--seq", type=int, default=None, help="tokens per text (default 128; 256 for --model large)")
    ap.add_argument("--model", default="base", choices=["base", "large"],
                    help="base = BERT-base + ResNet-50 (BASELINE configs[1]); large = BERT-large + ResNet-101 (configs[3])")
    ap.add_argument("--mode", default="train", choices=["train", "fwd"],
                    help="train = the BASELINE metric; fwd = forward only (training-mode BatchNorm, no backward)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="bf16 (BASELINE configs[1]); fp32 = the exact mode; fp8 = configs[4]: e4m3 operands in the text encoder's "
                         "forward Linears over bf16 storage (quote it with --batch 128)")
    ap.add_argument("--cpu-baseline-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-steps", type=int, default=10, help="timed steps of the exact (fp32) mode reported beside the bf16 "
                    "headline as `exact_mode` (default workload, 1 GPU only; 0 = skip)")
    ap.add_argument("--host-inputs", action="store_true", help="PCIe-inclusive variant: every step takes its batch from "
                    "pinned host memory through the double-buffered DevicePrefetcher (the default keeps inputs resident)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank code path on a single GPU)")
    ap.add_argument("--print-launch", action="store_true", help="print the child command of the self-launch and exit")
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = 32 if args.model == "large" else 64
    if args.seq is None:
        args.seq = 256 if args.model == "large" else 128
    return args


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_command(argv, gpus, port):
    """The command the self-launch runs: torch.distributed.run with one rank per GPU on this node, this script and the
    caller's own arguments (so the children parse exactly what the parent parsed)."""
    args = [a for a in argv if a != "--print-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args


def self_launch(args, argv):
    """WORLD_SIZE is unset and --gpus N > 1: start the N ranks as children. Nothing in this process has touched the GPU
    (no torch.cuda call, not even an import of torch), and nothing is exec'ed: the children are ordinary subprocesses."""
    cmd = child_command(argv, args.gpus, free_port())
    if args.print_launch:
        print(json.dumps(cmd))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def synth_batch(B, S, vocab, device, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, 224, 224, generator=g)
    ids = torch.randint(0, vocab, (B, S), generator=g)
    ids[:, 0] = 101
    mask = torch.ones(B, S)
    labels = torch.randint(0, 3, (B,), generator=g)
    return image.to(device), ids.to(device), mask.to(device), labels.to(device)


def cpu_baseline(model, B, steps):
    """Reference-style train step of the CPU oracle (fp32) on the host cores: the reported baseline, not the product."""
    import torch
    from oracle import model as OM
    from oracle.bert import BERT_BASE as OB
    from oracle.resnet import RESNET50 as OR
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # one GPU's host share on the pool is 16 cores; more threads than that make the fp32 oracle slower, not faster
    # (measured on the 2x64-core host: 32 threads 1.8 s/step, 128 threads 7.9 s/step at B=8)
    cores = min(cores, 16)
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle train step on {cores} threads ...", file=sys.stderr, flush=True)
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    names = [n for n, _ in model.named_parameters() if n not in ("contrastive_weight", "temperature")]
    cfg = dict(bert=OB, resnet=OR)
    image, ids, mask, labels = synth_batch(B, 128, 30522, "cpu", 1234)
    opt = {}
    OM.train_step(sd, names, image, ids, mask, labels, cfg, opt)  # warm-up (allocations, oneDNN primitives)
    t0 = time.perf_counter()
    for _ in range(steps):
        OM.train_step(sd, names, image, ids, mask, labels, cfg, opt)
    dt = time.perf_counter() - t0
    return {"value": round(B * steps / dt, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{steps} fp32 train steps (fwd+CE+bwd+clip+AdamW) of the CPU oracle at B={B}, S=128, 224x224 after 1 warm-up step"}


def pmc_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 --pmc passes (tools/run_pmc.sh + tools/pmc_traffic.py on the
    GPU box; bench.py cannot run the profiler around itself): an OFFLINE number, labelled as such. (None, None) if absent."""
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            return d["all_gemm"]["hbm_bytes_per_launch"], f"profiles/{name} (offline rocprofv3 --pmc passes, commit {d.get('commit', 'n/a')})"
        except Exception:
            continue
    return None, None


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    if args.print_launch:
        print(json.dumps([sys.executable, os.path.join(ROOT, "bench.py")] + [a for a in argv if a != "--print-launch"]))
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    ranks_seen = 1
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        ones = torch.ones(1, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(ones)  # every rank contributes 1: the collective really spans `world` processes
        ranks_seen = int(ones.item())
        if ranks_seen != world or dist.get_world_size() != world:
            raise SystemExit(f"rank {rank}: all-reduce of ones gave {ranks_seen}, world {world}")

    import multimodal_sentiment_aanalysis_amd as mm
    from multimodal_sentiment_aanalysis_amd import _lib
    from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep

    torch.manual_seed(0)
    if args.model == "large":
        model = mm.MultimodalTransformerModel(bert_config=mm.BERT_LARGE, resnet_config=mm.RESNET101)
        fwd_gflop = C3_FWD_GFLOP_PER_PAIR
    else:
        model = mm.MultimodalTransformerModel()  # BERT-base + ResNet-50 + fusion head, random init (no checkpoints offline)
        fwd_gflop = FWD_GFLOP_PER_PAIR * (args.seq / 128.0 if args.seq != 128 else 1.0)
    peak = PEAK_FP32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS  # (non-MX fp8 MFMA has the bf16 rate)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "base":
        cpu = cpu_baseline(model, 16, args.cpu_baseline_steps)
    trainer = FusedTrainStep(model, device, precision=args.precision)  # (image encoder on its own stream unless MMSA_TWO_STREAMS=0)
    batch = synth_batch(args.batch, args.seq, 30522, device, 1234 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def fwd_step(image, ids, mask, labels):
        # forward of the training graph (batch-statistics BatchNorm, dropout active) without autograd: same kernels as the
        # forward half of a training step, nothing saved for a backward
        with torch.no_grad():
            return model(image, ids, mask, labels)

    step_fn = trainer.step if args.mode == "train" else fwd_step
    if args.mode == "fwd":
        model.train()

    feed = None
    if args.host_inputs:  # 4 distinct pinned host batches, cycled; copies ride a side stream one step ahead
        from multimodal_sentiment_aanalysis_amd.dataLoader import DevicePrefetcher
        host = [tuple(t.cpu().pin_memory() for t in synth_batch(args.batch, args.seq, 30522, "cpu", 1234 + rank + 100 * i))
                for i in range(4)]

        class _Cycle:
            dataset = None

            def __init__(self, n):
                self.n = n

            def __len__(self):
                return self.n

            def __iter__(self):
                return (host[i % len(host)] for i in range(self.n))

        def feed(n):
            return iter(DevicePrefetcher(_Cycle(n), device))

    for b in (feed(args.warmup) if feed else [batch] * args.warmup):
        step_fn(*b)
    L = _lib.load()
    # (1) the timed regions: exactly `steps` steps each, un-instrumented, barrier + synchronize on both sides
    region_s = []
    loss = None
    for _ in range(max(1, args.repeats)):
        sync()
        t0 = time.perf_counter()
        for b in (feed(args.steps) if feed else [batch] * args.steps):
            out = step_fn(*b)
        sync()
        region_s.append(time.perf_counter() - t0)
        loss = out[0] if args.mode == "train" else None
    el = torch.tensor(region_s, dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)  # max over ranks, region by region
    region_s = el.tolist()
    dt = region_s[0]
    median_s = sorted(region_s)[len(region_s) // 2]
    # (2) the roofline passes, right after the timed steps (skipped under rocprofv3, which times the kernels itself):
    #     (a) HIP events on the launch stream around every MFMA GEMM launch — each launch bracketed in exactly one of
    #         `psteps` steps (index % psteps), because an event pair drains the queue around its kernel;
    #     (b) the kernels' own clock: first-workgroup-start / last-workgroup-end stamps (s_memrealtime) written by the
    #         GEMM kernels themselves, nothing added to the queue. (b) is what rocprofv3 --kernel-trace reports and what
    #         `roofline.achieved` uses; (a) still carries ~12 us of queue drain per launch and is reported beside it.
    psteps = min(args.steps, 5)
    STAMP_STEPS = 3  # steps averaged by pass (b); MMSA_PROF_DUMP then holds 3 rows per launch of a step
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    ms_ev, fl_ev, n_ev = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    fwd = None
    alg_bytes_per_launch = None
    isolated = None
    if os.environ.get("MMSA_BENCH_NOPROF", "0") == "0":
        L.mmsa_prof_mode(0)
        L.mmsa_prof_begin(psteps * 2400)
        sync()
        for ps in range(psteps):
            L.mmsa_prof_sample(psteps, ps)
            step_fn(*batch)
        sync()
        L.mmsa_prof_end(ctypes.byref(ms_ev), ctypes.byref(fl_ev), ctypes.byref(n_ev))
        L.mmsa_prof_mode(1)
        L.mmsa_prof_begin(STAMP_STEPS * 2400)
        sync()
        for _ in range(STAMP_STEPS):
            step_fn(*batch)
        sync()
        dump = os.environ.get("MMSA_PROF_DUMP")  # per-launch shape table: <path> for this pass, <path>.fwd.csv for the forward pass
        L.mmsa_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
        alg_bytes_per_launch = L.mmsa_prof_last_bytes() / n.value if n.value else None
        # (c) the same GEMM launches with the two encoders on ONE stream: in the timed steps the image encoder runs on its own
        #     stream, so a GEMM of one encoder shares the chip with whatever the other encoder has in flight and its own duration
        #     stretches (the step gets shorter, each kernel longer). `roofline.achieved` is the as-run figure (what rocprofv3 sees
        #     for this command); `roofline.isolated` is the kernel alone on the chip — the figure that describes the kernel.
        img_net = getattr(trainer, "_image_net", None)
        if getattr(trainer, "two_streams", False) and img_net is not None and getattr(img_net, "_side", None) is not None:
            img_net.join()
            sync()
            img_net.use_side_stream(False)
            had_wgrad_stream = getattr(img_net, "_wgrad_stream", None) is not None
            if had_wgrad_stream:
                img_net.use_wgrad_stream(False)
            if dump:
                os.environ["MMSA_PROF_DUMP"] = dump + ".isolated.csv"
            ms_i, fl_i, n_i = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            step_fn(*batch)
            L.mmsa_prof_begin(STAMP_STEPS * 2400)
            sync()
            for _ in range(STAMP_STEPS):
                step_fn(*batch)
            sync()
            L.mmsa_prof_end(ctypes.byref(ms_i), ctypes.byref(fl_i), ctypes.byref(n_i))
            img_net.use_side_stream(True)
            if had_wgrad_stream:
                img_net.use_wgrad_stream(True)
            if ms_i.value > 0:
                tf_i = fl_i.value / (ms_i.value * 1e-3) / 1e12
                isolated = {"achieved": round(tf_i, 2), "frac": round(tf_i / peak, 4),
                            "kernel_ms_per_step": round(ms_i.value / STAMP_STEPS, 3), "launches": n_i.value // STAMP_STEPS,
                            "what": "the same launches with everything on one stream (MMSA_TWO_STREAMS=0 MMSA_WGRAD_STREAM=0): each kernel alone on the chip"}
        if dump:
            os.environ["MMSA_PROF_DUMP"] = dump + ".fwd.csv"
        L.mmsa_prof_mode(0)
        ms.value /= STAMP_STEPS
        fl.value /= STAMP_STEPS
        n.value //= STAMP_STEPS
        if args.mode == "train":
            # (3) the forward alone (north_star: ">= 50 % MFMA roofline on the forward at bs=64"): 30 passes of the
            #     training-mode forward, then its GEMM launches by the kernels' own clock
            model.train()
            for _ in range(5):
                fwd_step(*batch)
            sync()
            t0 = time.perf_counter()
            for _ in range(30):
                fwd_step(*batch)
            sync()
            f_dt = (time.perf_counter() - t0) / 30
            fms, ffl, fn_ = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            L.mmsa_prof_mode(1)
            L.mmsa_prof_begin(STAMP_STEPS * 2400)
            for _ in range(STAMP_STEPS):
                fwd_step(*batch)
            sync()
            L.mmsa_prof_end(ctypes.byref(fms), ctypes.byref(ffl), ctypes.byref(fn_))
            L.mmsa_prof_mode(0)
            f_tf = args.batch * fwd_gflop / f_dt / 1e3
            g_tf = ffl.value / (fms.value * 1e-3) / 1e12 if fms.value > 0 else 0.0
            fwd = {"ms": round(f_dt * 1e3, 3), "pairs_per_s_per_gpu": round(args.batch / f_dt, 1),
                   "algorithmic_tflops": round(f_tf, 2), "frac_of_peak": round(f_tf / peak, 4),
                   "gemm_tflops": round(g_tf, 2), "gemm_frac_of_peak": round(g_tf / peak, 4),
                   "gemm_ms": round(fms.value / STAMP_STEPS, 3), "gemm_launches": fn_.value // STAMP_STEPS,
                   "what": "training-mode forward of the same model and batch, no autograd; 30 passes after the timed steps"}
    # (2b) N > 1: how long a step WAITS for the gradient collectives the backward did not hide — timing events on the compute stream
    #      around the join with the collective stream (GradReducer.finish), five extra steps after the timed regions
    collective = None
    if world > 1 and args.mode == "train" and getattr(trainer, "reducer", None) is not None:
        trainer.exposed_events = []
        for _ in range(5):
            trainer.step(*batch)
        sync()
        ex = sorted(e0.elapsed_time(e1) for e0, e1 in trainer.exposed_events)
        trainer.exposed_events = None
        ex_t = torch.tensor([ex[len(ex) // 2]], dtype=torch.float64, device=device)
        dist.all_reduce(ex_t, op=dist.ReduceOp.MAX)
        collective = {"exposed_ms_per_step": round(ex_t.item(), 3),
                      "form": "reduce-scatter + sharded AdamW + all-gather" if getattr(trainer, "shard_optimizer", False) else "all-reduce",
                      "payload": getattr(trainer.reducer, "payload", "fp32"),
                      "collectives_per_step": len(trainer.reducer.issued),
                      "what": "median over 5 steps (max over ranks) of the time the compute stream waits at the join with the "
                              "collective stream before the clip; the all-gather of the reduce-scatter form is not included"}
    # (3b) the tolerance-meeting mode beside the headline (north_star: logits within 1e-3, loss within 1e-4 of the CPU reference
    #      — met by precision="fp32", tests/test_golden_gpu.py::test_c0_full_size_fp32; bf16 storage cannot):
    #      a second model from the same seed (= the same initial weights), --exact-steps timed steps of the exact mode on the same
    #      batch, and the distance of its first-step logits from the bf16 step's at those weights.
    exact = None
    if (rank == 0 and world == 1 and args.mode == "train" and args.model == "base" and args.precision == "bf16"
            and args.exact_steps > 0 and not args.host_inputs and os.environ.get("MMSA_BENCH_NOPROF", "0") == "0"):
        try:
            torch.manual_seed(0)
            model_x = mm.MultimodalTransformerModel()
            trainer_x = FusedTrainStep(model_x, device, precision="fp32")
            for _ in range(3):
                trainer_x.step(*batch)
            # what the storage precision alone does to the logits: ONE dropout-free twin of the model (same seed = the headline's
            # initial weights; the head's dropout masks are drawn per call, so models with dropout on cannot be compared), its
            # training-mode forward in bf16 and in fp32 storage on the same batch
            from multimodal_sentiment_aanalysis_amd.engine import materialize
            torch.manual_seed(0)
            model_d = mm.MultimodalTransformerModel(dropout=0.0)
            model_d.train()
            lgs = []
            for prec in ("bf16", "fp32"):
                materialize(model_d, device, prec)
                with torch.no_grad():
                    lgs.append(model_d(*batch)[0].detach().float().clone())
            dlog = (lgs[0] - lgs[1]).abs().max().item()
            del model_d
            sync()
            t0 = time.perf_counter()
            for _ in range(args.exact_steps):
                trainer_x.step(*batch)
            sync()
            x_dt = (time.perf_counter() - t0) / args.exact_steps
            x_tf = args.batch * 3 * fwd_gflop / x_dt / 1e3
            exact = {"precision": "fp32", "ms_per_step": round(x_dt * 1e3, 3), "pairs_per_s": round(args.batch / x_dt, 2),
                     "steps": args.exact_steps, "warmup": 3, "step_algorithmic_tflops": round(x_tf, 2),
                     "frac_of_fp32_peak": round(x_tf / PEAK_FP32_TFLOPS, 4),
                     "dlogits_bf16_vs_fp32_forward": round(dlog, 6),
                     "what": "the same workload with precision=\"fp32\" (fp32 storage, fp32-MFMA GEMMs and attention): the mode that meets "
                             "north_star's 1e-3 / 1e-4 against the CPU reference (test_c0_full_size_fp32); dlogits_bf16_vs_fp32_forward = max |logits| "
                             "difference between the bf16-storage and the fp32-storage training-mode forward of a dropout-free twin "
                             "(same seed, same batch): what bf16 STORAGE costs on this random-init BatchNorm graph (DESIGN.md section 4)"}
            del trainer_x, model_x
            torch.cuda.empty_cache()
        except Exception as e:  # a report, never a reason to lose the line
            exact = {"error": str(e)[:200]}
    # (4) the peak restated on THIS box: CU count x sustained matrix-core clock x MFMA FLOP/CU/clk. The clock is
    #     measured under a dense bf16 MFMA load on pseudo-random operands (mmsa_mfma_clock_probe: ~0.5 s of back-to-back launches,
    #     s_memtime / s_memrealtime stamps of the last one); bf16 16x16x32 = 16384 FLOP per 16 cycles per SIMD = 4096 FLOP/CU/clk.
    clock = None
    if rank == 0 and os.environ.get("MMSA_BENCH_NOPROF", "0") == "0":
        try:
            cus = torch.cuda.get_device_properties(device).multi_processor_count
            ws = torch.zeros(cus * 1040, dtype=torch.uint8, device=device)
            L.mmsa_mfma_clock_probe(ctypes.c_void_p(ws.data_ptr()), cus, 4000, 60, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            st = ws[:cus * 16].view(torch.int64).view(cus, 2).double()
            ghz = (st[:, 0] / st[:, 1] * 0.1).median().item()
            mfma_tf = cus * 4 * 4000 * 16 * 16384 / (st[:, 1].median().item() * 1e-8) / 1e12  # what that loop itself delivered
            clock = {"cus": cus, "sustained_mfma_clock_ghz": round(ghz, 3), "max_clock_ghz": round(getattr(torch.cuda.get_device_properties(device), "clock_rate", 2400000) / 1e6, 3),
                     "peak_at_sustained_clock_tflops": round(cus * 4096 * ghz / 1e3, 1),
                     "mfma_only_loop_tflops": round(mfma_tf, 1),
                     "what": "dense bf16 v_mfma_f32_16x16x32 loop on pseudo-random operands, one wave per SIMD on every CU, 60 back-to-back launches; "
                             "clock = s_memtime / s_memrealtime of the last launch (median over workgroups); 4096 FLOP/CU/clk"}
        except Exception as e:  # the probe is a report, never a reason to lose the line
            clock = {"error": str(e)}
    if rank == 0:
        pairs = args.batch * world * args.steps
        flop_mult = 3 if args.mode == "train" else 1
        gemm_tflops = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        gemm_tflops_ev = fl_ev.value / (ms_ev.value * 1e-3) / 1e12 if ms_ev.value > 0 else 0.0
        step_tflops = pairs * flop_mult * fwd_gflop / dt / 1e3 / world
        traffic, traffic_src = pmc_traffic() if (args.model == "base" and args.mode == "train" and args.precision == "bf16"
                                                 and args.batch == 64 and args.seq == 128) else (None, None)
        names = {"base": "BERT-base S=%d + ResNet-50 224x224", "large": "BERT-large S=%d + ResNet-101 224x224"}
        what = ("train step (fwd+CE+bwd+grad all-reduce+clip+AdamW)" if args.mode == "train" else
                "forward only (training-mode BatchNorm, no autograd)")
        metric = "(image,text) pairs/sec/node, BERT-base+ResNet50 bs=64/GPU"
        if args.mode != "train" or args.model != "base":
            metric = "(image,text) pairs/sec/node, %s, %s, bs=%d/GPU" % (names[args.model] % args.seq, args.mode, args.batch)
        out = {
            "metric": metric,
            "value": round(pairs / dt, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic" + (" (host-resident, PCIe-inclusive)" if args.host_inputs else ""),
            "config": {"workload": "%s: %s + MHA fusion head, 3-class CE, random init" % (what, names[args.model] % args.seq),
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "seq_len": args.seq,
                       "image": "224x224x3", "parallelism": f"dp{world}", "ranks_seen_by_all_reduce": ranks_seen,
                       "backend": args.backend if world > 1 else None},
            "protocol": {"regions": len(region_s), "steps_per_region": args.steps,
                         "ms_per_step_by_region": [round(s / args.steps * 1e3, 3) for s in region_s],
                         "median_ms_per_step": round(median_s / args.steps * 1e3, 3),
                         "median_pairs_per_s": round(pairs / median_s, 2),
                         "note": "value / ms_per_step = the FIRST region after the warm-up (max over ranks)"},
            "roofline": {"bound": "mfma", "achieved": round(gemm_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(alg_bytes_per_launch) if alg_bytes_per_launch else None,
                         "streams": (1 + (1 if getattr(trainer, "two_streams", False) else 0)
                                     + (1 if getattr(getattr(trainer, "_image_net", None), "_wgrad_stream", None) is not None else 0)),
                         "isolated": isolated,
                         "kernel": "every matrix-core GEMM launch (gemm2_kernel + split-K reducer; precision fp32: the "
                                   "fp32-MFMA kernel) of 3 steps right after the timed steps, averaged per step (NT/NN/TN, "
                                   "implicit-GEMM convolutions, grouped weight gradients); duration = in-kernel clock, "
                                   "first workgroup start to last workgroup end",
                         "launches": n.value, "kernel_ms_per_step": round(ms.value, 3),
                         "achieved_hip_events": round(gemm_tflops_ev, 2),
                         "kernel_ms_per_step_hip_events": round(ms_ev.value, 3),
                         "step_algorithmic_tflops_per_gpu": round(step_tflops, 2),
                         "step_frac_of_peak": round(step_tflops / peak, 4)},
        }
        if clock is not None:
            out["roofline"]["peak_on_this_box"] = clock
            if clock.get("peak_at_sustained_clock_tflops") and args.precision != "fp32":
                out["roofline"]["frac_of_peak_at_sustained_clock"] = round(gemm_tflops / clock["peak_at_sustained_clock_tflops"], 4)
        if fwd is not None:
            out["forward"] = fwd
        if exact is not None:
            out["exact_mode"] = exact
        if collective is not None:
            out["collective"] = collective
        if loss is not None:
            out["loss"] = round(float(loss), 5)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
