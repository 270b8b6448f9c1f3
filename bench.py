#!/usr/bin/env python3
"""Benchmark of the hot path: one training step (forward + CE + backward + gradient all-reduce + clip + AdamW) of
BERT-base + ResNet-50 + the reference's attention fusion head on synthetic 224x224 RGB + 128-token batches, bs=64/GPU.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (BASELINE.json metric: (image,text) pairs/sec/node). `roofline` is measured live: every
bf16 MFMA GEMM launch of the timed region is bracketed by HIP events on its launch stream (mmsa_prof_*); `cpu_baseline`
is the CPU oracle's train step (oracle/model.py) timed on the host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_GFLOP_PER_PAIR = 30.52  # BERT-base S=128 22.348 + ResNet-50 (no fc) 8.174, 2 FLOP/MAC
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
# Algorithmic HBM bytes of an average MFMA GEMM launch of this step (284 launches per step — the four weight and two bias
# gradients of a BERT layer are one launch, a stride-2 3x3 data gradient is four — ~20.6 GFLOP each on average): every
# distinct operand of a launch read once (a 3x3 implicit-GEMM gather counts each source pixel once), C written once (fp32
# for weight gradients): 16.73 GB per step over the per-shape table (profiles/r01_gemm_shapes.csv) + 67 MB for the
# parity classes' re-reads of dY = 16.80 GB / 284.
ALG_BYTES_PER_GEMM_LAUNCH = 59.15e6


def synth_batch(B, S, vocab, device, seed):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, 224, 224, generator=g)
    ids = torch.randint(0, vocab, (B, S), generator=g)
    ids[:, 0] = 101
    mask = torch.ones(B, S)
    labels = torch.randint(0, 3, (B,), generator=g)
    return image.to(device), ids.to(device), mask.to(device), labels.to(device)


def cpu_baseline(model, B, steps):
    """Reference-style train step of the CPU oracle (fp32) on the host cores: the reported baseline, not the product."""
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
    """HBM bytes per GEMM launch from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json, produced by
    tools/pmc_traffic.py on the GPU box; bench.py cannot run the profiler around itself). None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)["all_gemm"]["hbm_bytes_per_launch"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--cpu-baseline-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-inputs", action="store_true", help="PCIe-inclusive variant: every step takes its batch from "
                    "pinned host memory through the double-buffered DevicePrefetcher (the default keeps inputs resident)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank code path on a single GPU)")
    args = ap.parse_args()

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
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import multimodal_sentiment_aanalysis_amd as mm
    from multimodal_sentiment_aanalysis_amd import _lib
    from multimodal_sentiment_aanalysis_amd.fused import FusedTrainStep

    torch.manual_seed(0)
    model = mm.MultimodalTransformerModel()  # BERT-base + ResNet-50 + fusion head, random init (no checkpoints offline)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, 16, args.cpu_baseline_steps)
    trainer = FusedTrainStep(model, device, precision=args.precision,
                             two_streams=os.environ.get("MMSA_TWO_STREAMS", "0") == "1")  # A/B on one box: 19.23 ms single stream, 19.50 with two
    batch = synth_batch(args.batch, args.seq, 30522, device, 1234 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
        trainer.step(*b)
    L = _lib.load()
    # (1) the timed region: exactly `steps` steps, un-instrumented (HIP events around every GEMM launch cost ~2 ms
    #     per step, so they are kept out of the throughput number)
    sync()
    t0 = time.perf_counter()
    for b in (feed(args.steps) if feed else [batch] * args.steps):
        loss, _ = trainer.step(*b)
    sync()
    dt = time.perf_counter() - t0
    # (2) the roofline pass: the same steps again with HIP events on the launch stream around every MFMA GEMM launch
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
    if os.environ.get("MMSA_BENCH_NOPROF", "0") == "0":
        L.mmsa_prof_mode(0)
        L.mmsa_prof_begin(psteps * 1200)
        sync()
        for ps in range(psteps):
            L.mmsa_prof_sample(psteps, ps)
            trainer.step(*batch)
        sync()
        L.mmsa_prof_end(ctypes.byref(ms_ev), ctypes.byref(fl_ev), ctypes.byref(n_ev))
        L.mmsa_prof_mode(1)
        L.mmsa_prof_begin(STAMP_STEPS * 1200)
        sync()
        for _ in range(STAMP_STEPS):
            trainer.step(*batch)
        sync()
        L.mmsa_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
        L.mmsa_prof_mode(0)
        ms.value /= STAMP_STEPS
        fl.value /= STAMP_STEPS
        n.value //= STAMP_STEPS
    el = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    dt = el.item()
    if rank == 0:
        pairs = args.batch * world * args.steps
        gemm_tflops = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        gemm_tflops_ev = fl_ev.value / (ms_ev.value * 1e-3) / 1e12 if ms_ev.value > 0 else 0.0
        step_tflops = pairs * 3 * FWD_GFLOP_PER_PAIR / dt / 1e3 / world
        out = {
            "metric": "(image,text) pairs/sec/node, BERT-base+ResNet50 bs=64/GPU",
            "value": round(pairs / dt, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic" + (" (host-resident, PCIe-inclusive)" if args.host_inputs else ""),
            "config": {"workload": "train step (fwd+CE+bwd+grad all-reduce+clip+AdamW): BERT-base S=%d + ResNet-50 224x224 + "
                                   "MHA fusion head, 3-class CE, random init" % args.seq,
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "seq_len": args.seq,
                       "image": "224x224x3", "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": round(gemm_tflops, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(),
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_GEMM_LAUNCH,
                         "kernel": "gemm2_kernel (+ split-K reducer): every MFMA GEMM launch of 3 steps right after the "
                                   "timed steps, averaged per step (NT/NN/TN, implicit-GEMM convolutions, grouped weight gradients); duration = "
                                   "in-kernel clock, first workgroup start to last workgroup end",
                         "launches": n.value, "kernel_ms_per_step": round(ms.value, 3),
                         "achieved_hip_events": round(gemm_tflops_ev, 2),
                         "kernel_ms_per_step_hip_events": round(ms_ev.value, 3),
                         "step_algorithmic_tflops_per_gpu": round(step_tflops, 2),
                         "step_frac_of_peak": round(step_tflops / PEAK_BF16_TFLOPS, 4)},
            "loss": round(float(loss), 5),
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
