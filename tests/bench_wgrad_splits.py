import os, sys, math, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from rsuper_amd.hip import ops, lib
dev='cuda'; B=2
for (s, ci, co) in [(96,32,32),(48,64,64),(96,96,64)]:
    x = torch.randn((B,s,s,s,ci), device=dev).bfloat16(); dy = torch.randn((B,s,s,s,co), device=dev).bfloat16()
    mr = torch.stack([torch.zeros(B,ci,device=dev), torch.ones(B,ci,device=dev)],-1).contiguous()
    dw = torch.empty((co,ci,3,3,3), device=dev)
    nch = -(-ci//32); gy = 1 if co<=32 else 3*(-(-co//64))
    for blocks in (128, 256, 512, 768, 1024, 2048):
        splits = max(1, blocks // (nch*gy))
        ws = torch.empty((splits*27*co*ci,), device=dev)
        def fn():
            lib.check(ops._L().rsuper_conv3_wgrad(lib.BF16, 1, x.data_ptr(), ci, ci, mr.data_ptr(), None, 0, 0, None, dy.data_ptr(), co, co, None, 0, 0,
                                                  dw.data_ptr(), None, ws.data_ptr(), B, s, s, s, splits, ops._stream()), 'wgrad')
        fn(); torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10): fn()
        en.record(); torch.cuda.synchronize()
        t = st.elapsed_time(en)/10
        fl = 2.0*B*s**3*co*ci*27
        print(f'S{s} {ci}->{co} blocks~{splits*nch*gy:5d} splits {splits:4d}: {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF', flush=True)
