"""Data-parallel layer on CPU: 2 processes over gloo (127.0.0.1).  Checks that the
bucketed, hook-driven all-reduce yields exactly the mean of the per-rank
gradients, that parameters are broadcast from rank 0, that bucket order follows
backward order (last layer first), and that set_to_none zero_grad is survived."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(12, 32), nn.ReLU(), nn.Linear(32, 16), nn.ReLU(), nn.Linear(16, 5))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from hawkeye_amd import ddp
    r, w, _ = ddp.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    model = _net(100 + rank)                       # different init per rank: broadcast must fix it
    red = ddp.GradientAllReducer(model, bucket_mb=0.002)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    torch.manual_seed(7)
    x = torch.randn(8, 12)
    y = torch.randint(0, 5, (8,))
    xs, ys = x[rank::world], y[rank::world]
    for step in range(2):
        if step == 0:
            red.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)        # hostile caller: hook must re-attach the views
            for b in red.buckets:
                b.flat.zero_()
        loss = nn.functional.cross_entropy(model(xs), ys)
        loss.backward()
        red.finish()
        if step == 0:
            grads = [p.grad.clone() for p in model.parameters()]
        opt.step()
    if rank == 0:
        torch.save({'grads': grads, 'params': [p.detach().clone() for p in model.parameters()],
                    'buckets': red.describe()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_gloo_allreduce(tmp_path, world):
    """2 ranks, and 8 (the node size of BASELINE.json's scaling points: one process per GPU) - every rank trains its own
    shard, the hook-driven bucketed all-reduce must reproduce the single-process mean gradient on all of them."""
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    # single-process reference: same init as rank 0, mean of the shard losses
    model = _net(100)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    torch.manual_seed(7)
    x = torch.randn(8, 12)
    y = torch.randint(0, 5, (8,))
    for step in range(2):
        opt.zero_grad()
        loss = sum(nn.functional.cross_entropy(model(x[r::world]), y[r::world]) for r in range(world)) / world
        loss.backward()
        if step == 0:
            for g, p in zip(got['grads'], model.parameters()):
                torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-7)
        opt.step()
    for a, p in zip(got['params'], model.parameters()):
        torch.testing.assert_close(a, p.detach(), rtol=1e-5, atol=1e-6)
    assert len(got['buckets']) >= 2                # tiny bucket limit -> several buckets, last layer first
    assert got['buckets'][0][0] >= 1


def test_single_process_reducer_is_identity():
    from hawkeye_amd import ddp
    model = _net(3)
    red = ddp.GradientAllReducer(model)
    red.zero_grad()
    x = torch.randn(4, 12)
    model(x).sum().backward()
    red.finish()
    ref = _net(3)
    ref(x).sum().backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(p.grad, q.grad)
    assert sum(n for n, _ in red.describe()) == 6


def test_bucket_views_follow_parameter_memory_format():
    """channels_last conv weights get channels_last gradient views inside the flat bucket (same bytes, matching strides);
    a step through them equals plain SGD."""
    import torch
    from hawkeye_amd import ddp
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1))
    ref.load_state_dict(net.state_dict())
    net = net.to(memory_format=torch.channels_last)
    red = ddp.GradientAllReducer(net, broadcast=False)
    for p in net.parameters():
        assert p.grad.stride() == p.stride() and p.grad.shape == p.shape
    flat_ptrs = {b.flat.data_ptr() for b in red.buckets}
    assert all(any(fp <= p.grad.data_ptr() < fp + b.nbytes for fp, b in zip(flat_ptrs, red.buckets)) for p in net.parameters())
    x = torch.randn(2, 3, 6, 6)
    o1, o2 = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9), torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9)
    for _ in range(2):
        red.zero_grad()
        net(x.contiguous(memory_format=torch.channels_last)).square().mean().backward()
        red.finish()
        o1.step()
        o2.zero_grad()
        ref(x).square().mean().backward()
        o2.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, atol=1e-6)


def test_tiny_parameter_joins_the_next_bucket():
    """A bias in front of a weight far above the bucket limit must not become a collective of its own."""
    from hawkeye_amd import ddp
    net = nn.Sequential(nn.Linear(64, 8), nn.ReLU(), nn.Linear(8, 600000))     # last layer: 2.4 MB bias, 19 MB weight
    red = ddp.GradientAllReducer(net, bucket_mb=4.0, broadcast=False)
    sizes = red.describe()
    assert sizes[0][0] == 1 and sizes[0][1] > 2.0            # the 2.4 MB bias alone is big enough to stand alone
    net2 = nn.Sequential(nn.Linear(64, 8), nn.ReLU(), nn.Linear(8, 1000))       # 4 KB bias: joins its 32 KB weight
    red2 = ddp.GradientAllReducer(net2, bucket_mb=0.03, broadcast=False)
    first = red2.describe()[0]
    assert first[0] == 2


def test_zero_grad_clears_nothing_and_first_gradient_is_written():
    """zero_grad() drops `.grad` instead of clearing the flat buffers: the step's first gradient is WRITTEN into its view
    (no memset + read-modify-write), a second backward before the step accumulates in place, and a parameter that gets no
    gradient in a step never hands the previous step's values to the optimiser."""
    from hawkeye_amd import ddp
    torch.manual_seed(1)
    net = nn.Sequential(nn.Linear(6, 5), nn.ReLU(), nn.Linear(5, 3))
    extra = nn.Linear(4, 2)                                   # a branch that only some steps use
    holder = nn.ModuleList([net, extra])
    red = ddp.GradientAllReducer(holder, bucket_mb=0.0001, broadcast=False)
    x, u = torch.randn(7, 6), torch.randn(7, 4)
    red.zero_grad()
    assert all(p.grad is None for p in holder.parameters())
    (net(x).sum() + extra(u).sum()).backward()
    red.finish()
    views = {p: v for b in red.buckets for p, v in zip(b.params, b.views)}
    for p in holder.parameters():
        assert p.grad.data_ptr() == views[p].data_ptr()       # landed in the flat buffer
    first = {p: p.grad.clone() for p in holder.parameters()}
    # second step: `extra` unused -> the flat buffers still hold its old gradient, the optimiser must not see it
    red.zero_grad()
    net(x).sum().backward()
    red.finish()
    for p in extra.parameters():
        assert p.grad is None                                 # (no process group: nothing reduced, the optimiser skips it)
    for p in net.parameters():
        torch.testing.assert_close(p.grad, first[p])
    # two backwards before one step accumulate in place
    red.zero_grad()
    net(x).sum().backward()
    net(x).sum().backward()
    red.finish()
    for p in net.parameters():
        torch.testing.assert_close(p.grad, 2 * first[p])
        assert p.grad.data_ptr() == views[p].data_ptr()
