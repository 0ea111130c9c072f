#!/usr/bin/env python3
"""Stand-alone probe of the defect behind DESIGN.md 3.4 (ii): an ATen multi-block reduction captured in a hipGraph.  Captures `y = x.sum(0)` for a
(27648, 26) tensor, replays it with different kinds of eager work between the replays and reports the first replay whose result is wrong.
Result on this ROCm 7.0 / PyTorch 2.10 build: all variants stay correct -- the defect needs the context of the training step (it shows with
RSUPER_DEBUG_ATEN_BIAS_SUM=1 in `tools/medformer_step.py 60 bf16 netgraph-report`, where GraphedNetwork's self-verification reports it at
replay 12, and never without eager GPU work between the replays).  Kept as the starting point for a minimal reproducer.
Usage: python tools/repro_graph_reduce.py"""
import torch

dev = 'cuda'
torch.manual_seed(0)
x = torch.randn(27648, 26, device=dev)
ref = x.double().sum(0).float()


def run(kind, n=40):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            y = x.sum(0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        y = x.sum(0)
    other = torch.randn(27648, 26, device=dev)
    big = torch.randn(4096, 4096, device=dev)
    bad = None
    for i in range(n):
        g.replay()
        if kind == 'sync':
            torch.cuda.synchronize()
        elif kind == 'eager_sum':
            z = other.sum(0); float(z[0])
        elif kind == 'eager_alloc':
            t = [torch.empty(1 << 20, device=dev) for _ in range(8)]; float(big.sum()); del t
        elif kind == 'eager_gemm':
            z = big @ big; float(z[0, 0])
        torch.cuda.synchronize()
        err = float((y - ref).abs().max() / ref.abs().max())
        if err > 1e-3 and bad is None:
            bad = (i + 1, err)
    print(f'{kind:12s}: ' + ('all %d replays correct' % n if bad is None else 'first wrong result at replay %d (relative error %.2e)' % bad))


for kind in ('none', 'sync', 'eager_sum', 'eager_alloc', 'eager_gemm'):
    run(kind)
