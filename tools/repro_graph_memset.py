#!/usr/bin/env python3
"""Minimal reproducer of the hipGraph defect behind the "garbage after a few replays" failures: a hipMemsetAsync(ptr, 0, n)
captured into a graph writes the byte 0xC0 instead of 0 in replays that follow eager launches on the same stream (ROCm 7.2 / MI355X).
A captured graph = [memset(acc, 0, 8 bytes); acc += 1 (kernel)].  Every replay must leave acc == 1.0.  Between replays, every `every`-th
iteration runs some eager work (a conv net forward / backward by default, like the self-verification of rsuper_amd.graph).
Usage: python tools/repro_graph_memset.py [replays] [every] [plain|heavy]"""
import ctypes, struct, sys
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
every = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kind = sys.argv[3] if len(sys.argv) > 3 else 'heavy'
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
acc = torch.zeros(1, device='cuda', dtype=torch.float64)
one = torch.ones(1, device='cuda', dtype=torch.float64)
other = torch.zeros(4096, device='cuda', dtype=torch.uint8)
net = torch.nn.Sequential(torch.nn.Conv3d(1, 16, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv3d(16, 16, 3, padding=1)).cuda()
x = torch.randn(2, 1, 48, 48, 48, device='cuda')
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    rc = hip.hipMemsetAsync(acc.data_ptr(), 0, 8, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    acc += one
bad = []
for i in range(n):
    g.replay()
    v = float(acc)                       # synchronises
    if v != 1.0:
        bad.append((i, v, struct.pack('<d', v - 1.0).hex()))
    if every and i % every == every - 1:
        if kind == 'heavy':
            net(x).square().mean().backward()
        elif kind == 'memset':                      # an eager memset of another buffer with another value / size
            hip.hipMemsetAsync(other.data_ptr(), 0xC0, 4096, torch.cuda.current_stream().cuda_stream)
        elif kind == 'memset0':
            hip.hipMemsetAsync(other.data_ptr(), 0, 26, torch.cuda.current_stream().cuda_stream)
        elif kind == 'clone':
            y = [x.clone() for _ in range(20)]; torch._foreach_copy_(y, [x] * 20); del y
        elif kind == 'mix':
            net(x).square().mean().backward(); y = [x.clone() for _ in range(20)]; del y
            hip.hipMemsetAsync(other.data_ptr(), 0, 26, torch.cuda.current_stream().cuda_stream)
            z = torch.full((1 << 20,), float('nan'), device='cuda'); del z
        else:
            (x * 2).sum()
print(f'{len(bad)} of {n} replays left acc != 1.0', bad[:5])
