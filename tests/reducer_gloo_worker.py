"""world_size-2 gloo worker: rsuper_amd.reducer.GradReducer on a host-side module == mean of the ranks' local gradients,
over two consecutive steps (bucket re-arming), with small buckets so several collectives are in flight."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.train_ddp import init_distributed  # noqa: E402
from rsuper_amd.reducer import GradReducer  # noqa: E402


def make():
    return torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv3d(4, 6, 3, padding=1), torch.nn.ReLU(),
                               torch.nn.Conv3d(6, 2, 1))


def main():
    rank, local, world = init_distributed(backend='gloo')
    assert world == 2
    torch.manual_seed(rank)                      # different init per rank: the reducer must broadcast rank 0's parameters
    net = make()
    red = GradReducer(net, bucket_mb=0.0005)     # ~130 floats per bucket -> several buckets
    assert len(red.buckets) >= 3
    ref = make()
    torch.manual_seed(0)
    ref0 = make()
    for a, b in zip(net.parameters(), ref0.parameters()):
        assert torch.equal(a, b), 'parameters were not broadcast from rank 0'
    for step in range(2):
        for p in net.parameters():
            p.grad = None
        x = torch.randn(2, 1, 6, 6, 6, generator=torch.Generator().manual_seed(100 + 10 * step + rank))
        net(x).square().mean().backward()
        red.finish()
        ref.load_state_dict(net.state_dict())
        acc = [torch.zeros_like(p) for p in ref.parameters()]
        for r in range(world):
            ref.zero_grad()
            xr = torch.randn(2, 1, 6, 6, 6, generator=torch.Generator().manual_seed(100 + 10 * step + r))
            ref(xr).square().mean().backward()
            for a, p in zip(acc, ref.parameters()):
                a += p.grad / world
        for p, a in zip(net.parameters(), acc):
            assert torch.allclose(p.grad, a, atol=1e-6), (p.grad - a).abs().max()
            b, off = red._slot[p]
            assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * off, 'gradient does not live in its bucket'
    # bf16 on the wire (RSUPER_DDP_BF16): same flow, the mean of bf16-rounded gradients -- within 2^-8 of the largest entry of the f32 mean
    red.remove()
    redw = GradReducer(net, bucket_mb=0.0005, wire_dtype=torch.bfloat16)
    for p in net.parameters():
        p.grad = None
    x = torch.randn(2, 1, 6, 6, 6, generator=torch.Generator().manual_seed(500 + rank))
    net(x).square().mean().backward()
    redw.finish()
    acc = [torch.zeros_like(p) for p in ref.parameters()]
    for r in range(world):
        ref.zero_grad()
        xr = torch.randn(2, 1, 6, 6, 6, generator=torch.Generator().manual_seed(500 + r))
        ref(xr).square().mean().backward()
        for a, p in zip(acc, ref.parameters()):
            a += p.grad / world
    for p, a in zip(net.parameters(), acc):
        assert p.grad.dtype == torch.float32 and (p.grad - a).abs().max() <= 2.0 ** -8 * a.abs().max() + 1e-12, (p.grad - a).abs().max()
        assert not torch.equal(p.grad, a) or a.abs().max() == 0, 'the bf16 wire format was not used'
    redw.remove()
    red = GradReducer(net, bucket_mb=0.0005)
    # a second backward pass before finish() (gradient accumulation) is refused instead of corrupting the buckets
    for p in net.parameters():
        p.grad = None
    x = torch.randn(2, 1, 6, 6, 6)
    net(x).square().mean().backward()
    try:
        net(x).square().mean().backward()
        raise SystemExit('second backward before finish() was not detected')
    except RuntimeError:
        pass
    red.reset()
    # the bucket that completes last (first-registered parameters) is split off and kept small
    big = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4096), torch.nn.Linear(4096, 64))
    red.remove()
    red2 = GradReducer(big, bucket_mb=64, tail_mb=0.001)
    assert len(red2.buckets) == 2 and red2.buckets[-1].flat.numel() <= 262 and red2.buckets[-1].params[-1] is big[0].weight
    import copy
    big._rsuper_reducer = red2
    assert copy.deepcopy(big)._rsuper_reducer is None          # make_ema() after wrap_ddp must not clone buckets / hooks
    red2.remove()
    red = GradReducer(net, bucket_mb=0.0005)
    # a parameter without gradient must be reported (find_unused_parameters=False semantics)
    for p in net.parameters():
        p.grad = None
    net[0](torch.randn(1, 1, 4, 4, 4)).sum().backward()
    try:
        red.finish()
        raise SystemExit('missing gradient was not detected')
    except RuntimeError:
        pass
    dist.barrier()
    if rank == 0:
        print('REDUCER_OK')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
