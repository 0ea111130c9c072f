"""world_size-2 gloo worker for tests/test_host_cpu.py: exercises init_distributed / wrap_ddp / shard_indices and the
bench's max-over-ranks timing reduction on CPU (the compute kernels are MI355X-only)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.train_ddp import init_distributed, wrap_ddp, shard_indices  # noqa: E402


def main():
    rank, local, world = init_distributed(backend='gloo')
    assert world == 2
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv3d(4, 2, 1))
    ddp = wrap_ddp(net, local)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(2, 1, 8, 8, 8, generator=g)
    ddp(x).square().mean().backward()
    # reference: average of the two ranks' local gradients
    grads = [p.grad.clone() for p in net.parameters()]
    net2 = torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv3d(4, 2, 1))
    net2.load_state_dict(net.state_dict())
    acc = [torch.zeros_like(p) for p in net2.parameters()]
    for r in range(world):
        net2.zero_grad()
        xr = torch.randn(2, 1, 8, 8, 8, generator=torch.Generator().manual_seed(100 + r))
        net2(xr).square().mean().backward()
        for a, p in zip(acc, net2.parameters()):
            a += p.grad / world
    for gq, a in zip(grads, acc):
        assert torch.allclose(gq, a, atol=1e-6), (gq - a).abs().max()
    assert shard_indices(list(range(8)), rank, world) == list(range(8))[rank::2]
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == 2.0
    dist.barrier()
    if rank == 0:
        print('DDP_OK')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
