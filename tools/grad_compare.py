#!/usr/bin/env python3
"""Step-0 parameter gradients / logits of several runs (tools/drift.py --grads FILE) against a reference run (normally the f32 mode): a deterministic,
chaos-free measure of what each arithmetic variant does to ONE forward + backward.  Usage: grad_compare.py REF.pt NAME=FILE.pt [NAME=FILE.pt ...]"""
import sys
import torch
ref = torch.load(sys.argv[1])
runs = [(a.split('=')[0], torch.load(a.split('=')[1])) for a in sys.argv[2:]]
l2 = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm().clamp_min(1e-300))
cos = lambda u, v: float((u.double() * v.double()).sum() / (u.double().norm() * v.double().norm()).clamp_min(1e-300))
cat = lambda g: torch.cat([v.flatten() for v in g.values()])
print(f"{'run':10s} {'loss':>10s} {'logits relL2':>12s} {'grads relL2':>11s} {'cosine':>8s} {'|g|/|g_ref|':>11s}  per-tensor cosine min / median; worst tensors")
print(f"{'ref':10s} {ref['loss']:10.6f}")
rg = cat(ref['grads'])
for name, r in runs:
    g = cat(r['grads'])
    pc = {k: cos(r['grads'][k], ref['grads'][k]) for k in ref['grads']}
    ks = sorted(pc, key=pc.get)
    print(f"{name:10s} {r['loss']:10.6f} {l2(r['logits'], ref['logits']):12.4f} {l2(g, rg):11.4f} {cos(g, rg):8.4f} {float(g.double().norm() / rg.double().norm()):11.4f}  "
          f"{pc[ks[0]]:.3f} / {sorted(pc.values())[len(pc) // 2]:.3f}; " + ', '.join(f'{k} {pc[k]:.3f}' for k in ks[:4]))
if len(runs) >= 2:
    a, b = runs[0], runs[1]
    print(f"# {a[0]} vs {b[0]} directly: logits relL2 {l2(a[1]['logits'], b[1]['logits']):.4f}, grads relL2 {l2(cat(a[1]['grads']), cat(b[1]['grads'])):.4f}, cosine {cos(cat(a[1]['grads']), cat(b[1]['grads'])):.4f}")
    # level-wise: group tensors by module prefix
    import collections
    grp = collections.OrderedDict()
    for k in ref['grads']:
        grp.setdefault(k.split('.')[0], []).append(k)
    print('# per module (all its tensors together): cosine vs ref   ' + '   '.join(n for n, _ in runs))
    for m, ks in grp.items():
        row = []
        for n, r in runs:
            u = torch.cat([r['grads'][k].flatten() for k in ks]); v = torch.cat([ref['grads'][k].flatten() for k in ks])
            row.append(f'{cos(u, v):.4f}')
        print(f'  {m:10s} ' + '  '.join(row))
