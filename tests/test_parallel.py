"""N > 1 path on CPU: 2-rank gloo tests of the bucketed gradient reducer, the initial-state
broadcast and the rank-strided resumable sampler."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(0)
    shared = nn.Linear(8, 8)
    net = nn.ModuleDict({"a": shared, "a_alias": shared, "b": nn.Linear(8, 4),
                         "unused": nn.Linear(3, 3)})
    return net


def _worker(rank, world, port, q, exchange="all_reduce", overlap=True):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mono_vifi_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _net()
        if rank != 0:     # perturb: broadcast must restore rank 0's state
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
        parallel.broadcast_module_states([net], src=0)
        params = parallel.unique_parameters(net.values())
        assert len(params) == 6            # alias de-duplicated
        red = parallel.BucketedGradReducer(params, world, bucket_mb=0.0001,   # many tiny buckets
                                           exchange=exchange, overlap=overlap)
        assert red.num_buckets > 1
        used = {id(p) for k in ("a", "b") for p in net[k].parameters()}
        nb_used = sum(all(id(p) in used for p in b.params) for b in red.buckets)      # buckets that fill
        nb_unused = red.num_buckets - nb_used
        parallel.reset_comm_counts()
        torch.manual_seed(100)
        x_all = torch.randn(4 * world, 8)
        x = x_all[rank * 4:(rank + 1) * 4]
        for it in range(2):                # twice: zero_grad starts every step from grad = None
            red.zero_grad()
            y = net["b"](torch.relu(net["a"](x)) + net["a_alias"](x))   # shared module used twice
            y.pow(2).mean().backward()
            # where the exchanges were issued: every bucket that filled went out from a hook DURING
            # backward (the overlapped case); finish() only issues what never filled (the unused module)
            in_backward = red.issued_from_hook
            import warnings
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                red.finish()
            said = sum("exchanged AFTER backward" in str(w_.message) for w_ in caught)
            # the bucket of the unused module cannot fill: its exposed exchange is announced ONCE (first step only)
            assert said == (1 if (overlap and it == 0 and nb_unused > 0) else 0), (said, it)
            if overlap:
                assert in_backward >= nb_used and red.issued_from_finish == red.num_buckets - in_backward
                assert red.issued_from_finish <= nb_unused
            else:
                assert in_backward == 0 and red.issued_from_finish == red.num_buckets
        # after the exchange every gradient IS its bucket slice (packed by one multi-tensor copy per bucket); a
        # parameter no rank used leaves with grad None, as under DDP(find_unused_parameters=True) (train.py:208)
        assert all(p.grad.data_ptr() == red.bucket_view(p).data_ptr() for p in params if id(p) in used)
        assert all(p.grad is None for p in params if id(p) not in used)
        grads = [p.grad.clone() if p.grad is not None else None for p in params]
        # exactly one exchange per bucket per step, whichever form it takes
        cnt, nb = parallel.comm_counts(), red.num_buckets
        if exchange == "all_reduce":
            assert cnt == {"grad_all_reduce": 2 * nb}, cnt
        else:
            assert cnt == {"grad_reduce_scatter": 2 * nb, "grad_all_gather": 2 * nb}, cnt
            assert all(b.buf.numel() % world == 0 for b in red.buckets)
        q.put((rank, [g.tolist() if g is not None else None for g in grads], [p.detach().tolist() for p in params]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,overlap", [("all_reduce", True), ("all_reduce", False),
                                              ("reduce_scatter", True), ("reduce_scatter", False)])
def test_bucketed_reducer_matches_full_batch_gradient(exchange, overlap):
    """Both forms of the bucket exchange (one all-reduce; reduce-scatter + all-gather on the flat
    buffer), issued from the hooks during backward or after it, give the
    full-batch gradient on every rank with one exchange per bucket."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # reference: one process, the whole global batch
    net = _net()
    torch.manual_seed(100)
    x_all = torch.randn(4 * world, 8)
    y = net["b"](torch.relu(net["a"](x_all)) + net["a_alias"](x_all))
    y.pow(2).mean().backward()
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mono_vifi_amd import parallel
    ref_params = parallel.unique_parameters(net.values())
    for rank, grads, params in res:
        for g, p, rp in zip(grads, params, ref_params):
            assert torch.allclose(torch.tensor(p), rp.detach(), atol=0), "broadcast failed"
            if rp.grad is None:
                assert g is None, f"rank {rank}: an unused parameter must keep grad None"
                continue
            assert torch.allclose(torch.tensor(g), rp.grad, atol=1e-6), f"rank {rank} grad mismatch"
    assert res[0][1] == res[1][1]          # identical on both ranks


def test_distributed_sampler_partitions_and_resumes():
    from mono_vifi_amd import datasets

    class D:
        def __len__(self):
            return 103
    world = 4
    seen = []
    for r in range(world):
        s = datasets.CustomDistributedSampler(D(), seed=5, num_replicas=world, rank=r)
        s.set_epoch(3)
        idx = list(s)
        assert len(idx) == 103 // world == len(s)
        seen += idx
        s.set_start_iter(10)               # mid-epoch resume skips consumed samples
        assert list(s) == idx[10:]
    assert len(set(seen)) == len(seen) == 100          # disjoint, truncated to a multiple of world
    s0 = datasets.CustomSampler(D(), seed=5)
    s0.set_epoch(3)
    g = torch.Generator()
    g.manual_seed(8)
    assert list(s0) == torch.randperm(103, generator=g).tolist()
    s1 = datasets.CustomDistributedSampler(D(), seed=5, num_replicas=world, rank=1)
    s1.set_epoch(3)
    assert list(s1) == torch.randperm(103, generator=g.manual_seed(8)).tolist()[:100][1::world]


def test_tail_bucket_holds_the_last_arriving_gradients():
    """The bucket that fills last (the first-registered parameters: backward reaches them last) cannot hide its
    exchange behind anything, so it is kept small: `tail_mb` splits it off the last full-size bucket."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mono_vifi_amd import parallel
    mb = 1 << 18                                   # floats per MiB
    sizes = [mb // 2, mb // 2, mb, 3 * mb, 2 * mb, 4 * mb, 6 * mb]      # registration order
    params = [nn.Parameter(torch.zeros(n)) for n in sizes]
    red = parallel.BucketedGradReducer(params, world_size=1, bucket_mb=8.0, tail_mb=1.0)
    got = [[p.numel() for p in b.params] for b in red.buckets]
    # backward order = reversed registration order; 8 MiB buckets: [6], [4, 2], [3, 1, .5, .5] -> the last one
    # gives its last-arriving <= 1 MiB to a bucket of their own
    assert got == [[6 * mb], [4 * mb, 2 * mb], [3 * mb, mb], [mb // 2, mb // 2]]
    assert red.buckets[-1].params[-1] is params[0]
    for p in params:                               # every parameter owns a slice of its bucket; no gradient yet
        v = red.bucket_view(p)
        assert v.shape == p.shape and p.grad is None
        assert any(v.data_ptr() >= b.buf.data_ptr() and
                   v.data_ptr() < b.buf.data_ptr() + 4 * b.buf.numel() for b in red.buckets)
    # no split when the last bucket is small anyway, or when switched off
    red2 = parallel.BucketedGradReducer([nn.Parameter(torch.zeros(n)) for n in sizes], world_size=1,
                                        bucket_mb=8.0, tail_mb=0.0)
    assert [[p.numel() for p in b.params] for b in red2.buckets][-1] == [3 * mb, mb, mb // 2, mb // 2]


def test_single_process_keeps_backward_gradients_and_none_for_unused():
    """Without an exchange (one process, no forced collectives) nothing is packed: a gradient stays the tensor backward
    produced (no add into a zeroed bucket, no memset), a parameter the step does not reach keeps grad None (the
    optimiser skips it, as in the reference's single-process run), and zero_grad() resets both every step."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mono_vifi_amd import parallel
    torch.manual_seed(0)
    net = _net()
    params = parallel.unique_parameters(net.values())
    red = parallel.BucketedGradReducer(params, world_size=1, bucket_mb=0.0001)
    assert not red.exchanging and red.num_buckets > 1
    x = torch.randn(4, 8)
    ref = None
    for it in range(2):
        red.zero_grad()
        assert all(p.grad is None for p in params)
        net["b"](torch.relu(net["a"](x)) + net["a_alias"](x)).pow(2).mean().backward()
        red.finish()
        assert red.issued_from_hook + red.issued_from_finish == red.num_buckets
        used = [p for k in ("a", "b") for p in net[k].parameters()]
        assert all(p.grad is not None and p.grad.data_ptr() != red.bucket_view(p).data_ptr() for p in used)
        assert all(p.grad is None for p in net["unused"].parameters())
        got = [p.grad.clone() for p in used]
        if ref is not None:
            assert all(torch.equal(a, b) for a, b in zip(got, ref))     # no accumulation across steps
        ref = got
