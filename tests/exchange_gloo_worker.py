"""world_size-2 gloo worker: rsuper_amd.graph.exchange_gradients (the gradient exchange GraphedNetwork runs after its backward replay) ==
mean of the ranks' tensors, in place, over several buckets, for tensors of mixed shapes; a one-element list and an empty list too."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsuper_amd.train_ddp import init_distributed  # noqa: E402
from rsuper_amd.graph import exchange_gradients  # noqa: E402


def tensors(rank):
    g = torch.Generator().manual_seed(50 + rank)
    return [torch.randn(s, generator=g) for s in ((4, 1, 3, 3, 3), (4,), (6, 4, 3, 3, 3), (2, 6, 1, 1, 1), (2,), (7, 5))]


def main():
    rank, local, world = init_distributed(backend='gloo')
    assert world == 2
    mine = tensors(rank)
    ptrs = [t.data_ptr() for t in mine]
    want = [(a + b) / 2 for a, b in zip(tensors(0), tensors(1))]
    exchange_gradients(mine, bucket_bytes=600)           # ~150 floats per bucket -> several collectives in flight
    for t, w, p in zip(mine, want, ptrs):
        assert t.data_ptr() == p, 'the exchange must work in place (the tensors are the static gradient buffers of the backward graph)'
        assert torch.allclose(t, w, atol=1e-7), (t - w).abs().max()
    one = [torch.full((3,), float(rank + 1))]
    exchange_gradients(one)
    assert torch.equal(one[0], torch.full((3,), 1.5))
    exchange_gradients([])
    dist.barrier()
    if rank == 0:
        print('EXCHANGE_OK')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
