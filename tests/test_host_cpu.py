"""CPU tests of the host logic and of the C-ABI library surface (no compute calls without a GPU)."""
import argparse
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from rsuper_amd.hip import lib
    so = lib.SO_PATH
    if not os.path.exists(so):
        import __graft_entry__ as ge
        ge.build()
    L = ctypes.CDLL(so)
    hdr = open(os.path.join(ROOT, 'include', 'rsuper_hip.h')).read()
    declared = set(re.findall(r'\b(rsuper_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f'{name} declared in include/rsuper_hip.h but not exported'
    assert set(lib.exported_symbols()) == declared, set(lib.exported_symbols()) ^ declared
    assert b'gfx950' in lib.lib().rsuper_version()


def test_no_cpu_fallback():
    from rsuper_amd.hip import lib
    from rsuper_amd.model.dim3.unet import UNet
    net = UNet(1, 8, num_classes=3)
    with pytest.raises(lib.RSuperHipError):
        net(torch.zeros(1, 1, 16, 16, 16))
    if not torch.cuda.is_available():
        with pytest.raises(lib.RSuperHipError):
            lib.require_device()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'r-super_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert 'oracle' not in src.replace('the oracle', ''), f'{f} references oracle/'


def test_state_dict_matches_reference_layout():
    from rsuper_amd.model.dim3.unet import UNet
    from oracle.unet_oracle import unet_param_shapes
    for b, k in ((8, 5), (32, 26)):
        sd = UNet(1, b, num_classes=k).state_dict()
        sh = unet_param_shapes(1, b, k)
        assert set(sd) == set(sh)
        assert all(tuple(sd[n].shape) == sh[n] for n in sh)
    assert sum(v.numel() for v in UNet(1, 32, num_classes=26).state_dict().values()) == 40561338


def test_get_model_and_yaml_keys():
    import yaml
    from rsuper_amd.model.utils import get_model
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'r-super_amd', 'config', 'abdomenatlas_ufo', 'unet_3d.yaml')))
    args = argparse.Namespace(model='unet', dimension='3d', **cfg)
    net = get_model(args)
    assert net.outc.weight.shape == (26, 32, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        get_model(argparse.Namespace(model='attention_unet', dimension='3d', **cfg))
    # MedFormer from its YAML (reference key names, config/abdomenatlas_ufo/medformer_3d.yaml); chan_num is not forwarded, as in the reference
    mcfg = yaml.safe_load(open(os.path.join(ROOT, 'r-super_amd', 'config', 'abdomenatlas_ufo', 'medformer_3d.yaml')))
    margs = argparse.Namespace(model='medformer', dimension='3d', classification_branch=False, **mcfg)
    mf = get_model(margs, classes=['c%d' % i for i in range(42)])
    assert mf.outc.weight.shape == (42, 32, 1, 1, 1) and mf.aux_out.weight.shape == (42, 128, 1, 1, 1)
    assert len(mf.down4.trans_blocks.blocks) == 6 and len(mf.up4.conv_blocks) == 2 and mf.map_fusion.in_proj[2].weight.shape == (320, 320, 1, 1, 1)
    assert 35e6 < sum(p.numel() for p in mf.parameters()) < 40e6          # SURVEY: 37.9 M parameters
    with pytest.raises(NotImplementedError):
        get_model(argparse.Namespace(**{**vars(margs), 'classification_branch': True}))


def test_lr_schedule_and_ema_alpha(golden):
    from rsuper_amd.training.utils import exp_lr_scheduler_with_warmup, ema_alpha_for_step

    class O:
        param_groups = [{'lr': 6e-4}]
    lrs = []
    for e in [0, 1, 3, 5, 6, 50, 99]:
        o = O()
        o.param_groups = [{'lr': 6e-4}]
        lrs.append(exp_lr_scheduler_with_warmup(o, e, 5, 100))
    np.testing.assert_allclose(lrs, golden['train_step']['lr_sched'], rtol=1e-12)
    assert ema_alpha_for_step(0.99, 0) == 0 and ema_alpha_for_step(0.99, 1) == 0.5 and ema_alpha_for_step(0.99, 1000) == 0.99


def test_lesion_groups_and_ball_geometry(golden):
    from rsuper_amd.training import losses_foundation as lf
    import synth
    assert lf.lesion_groups(synth.TINY_CLASSES) == {'kidney_lesion': 1, 'pancreas_lesion': 4}
    assert lf.lesion_groups(synth.PANTS_CLASSES) == {'pancreas_lesion': 19}
    p = golden['primitives']
    for d in [1, 3, 5, 7, 8, 10, 15, 31, 40]:
        d_odd, ks = lf.ball_kernel_geometry(d)
        assert [ks, lf.ball_nnz(d_odd)] == list(p[f'ball_{d}_edge_nnz'])
    with pytest.raises(NotImplementedError):                 # the single-channel view refuses; calculate_loss merges first
        lf.lesion_groups(['kidney_lesion_1', 'kidney_lesion_2'])
    assert lf.lesion_channel_lists(synth.MULTI_CH_CLASSES) == {'pancreas_lesion': [3, 4]}
    t = torch.arange(2 * 5 * 2, dtype=torch.float32).reshape(2, 5, 2)
    m, names = lf.merge_lesion_channels(t, synth.MULTI_CH_CLASSES)
    assert names == ['pancreas_lesion'] and torch.equal(m[:, 0], torch.maximum(t[:, 3], t[:, 4]))


def test_pick_bn():
    from rsuper_amd.hip.ops import pick_bn
    from rsuper_amd.hip import lib
    L = lib.lib()
    assert L.rsuper_conv3_variant(-1) == 3          # default: round-1 choice + weight-stationary kernel on single-chunk 32-column launches
    assert pick_bn(32, torch.bfloat16) == 32 and pick_bn(64, torch.bfloat16) == 64 and pick_bn(128, torch.bfloat16) == 128
    assert pick_bn(96, torch.bfloat16) == 32 and pick_bn(96, torch.float32) == 32 and pick_bn(320, torch.bfloat16) == 64 and pick_bn(128, torch.float32) == 64


def test_fused_optimizer_resumes_from_torch_adamw_state():
    """resume_load_optimizer_state (rsuper_train/utils.py:58-62) hands FusedAdamWEMA a torch.optim.AdamW state_dict, whose
    `step` entries are per-parameter tensors: they are normalised to the shared int step count the fused kernel takes."""
    from rsuper_amd.training.utils import FusedAdamWEMA
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    ref = torch.optim.AdamW(ps, lr=6e-4, eps=1e-5, weight_decay=0.05)
    for _ in range(3):
        for p in ps:
            p.grad = torch.randn_like(p)
        ref.step()
    sd = ref.state_dict()
    assert torch.is_tensor(sd['state'][0]['step'])
    opt = FusedAdamWEMA(ps, lr=6e-4, betas=(0.9, 0.999), eps=1e-5, weight_decay=0.05)
    opt.load_state_dict(sd)
    sts = [opt._state_for(p) for p in ps]
    assert all(isinstance(st['step'], int) and st['step'] == 3 for st in sts)
    assert all(torch.equal(st['exp_avg'], ref.state[p]['exp_avg']) for st, p in zip(sts, ps))
    assert len({st['step'] for st in sts}) == 1


def test_parser_yaml_merge_fills_only_unset_attributes():
    """get_parser (train_ddp.py:392-548): a YAML key only fills attributes argparse does not define (:491-502); the named
    overrides follow (:513-529); `--batch_size` is the global batch (:632)."""
    from rsuper_amd.train_ddp import get_parser, merge_config, AverageMeter
    a = get_parser(['--lr', '0.001', '--classes_number', '42', '--batch_size', '4', '--epochs', '3', '--crop_size', '64'])
    assert a.base_lr == 0.001 and a.classes == 42 and a.batch_size == 4 and a.batch_size_global == 4 and a.epochs == 3
    assert a.training_size == [64, 64, 64]
    assert a.block == 'BasicBlock' and a.iter_per_epoch == 1000 and a.optimizer == 'adamw' and a.ema is True     # YAML-only keys
    assert a.loss == 'ball_dice_last' and a.report_volume_loss_basic == 1 and a.volume_loss_tolerance == 0.2       # CLI defaults
    b = get_parser([])
    assert b.epochs is None            # argparse defines --epochs (default None): the YAML's `epochs: 100` never fills it
    assert b.base_lr == 0.0006         # --lr always has a value, so it always overrides base_lr
    ns = argparse.Namespace(x=1)
    merge_config(ns, {'x': 2, 'y': 3})
    assert ns.x == 1 and ns.y == 3
    m = AverageMeter('loss', ':6.4f')
    m.update(2.0, 2); m.update(4.0, 2)
    assert m.avg == 3.0 and m.val == 4.0 and 'loss' in str(m)


def test_shard_indices_round_robin():
    from rsuper_amd.train_ddp import shard_indices
    chunk = list(range(10))
    assert shard_indices(chunk, 0, 4) == [0, 4, 8] and shard_indices(chunk, 3, 4) == [3, 7]


def test_ddp_world2_gloo_cpu():
    """N>1 path on CPU: two gloo ranks average gradients of a host-side module through wrap_ddp and agree."""
    script = os.path.join(ROOT, 'tests', 'ddp_gloo_worker.py')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29533', script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'DDP_OK' in r.stdout


def test_pack_bits_matches_reference_writer_and_needs_device():
    """pack_bits == np.packbits(bool, axis=0) per sample (dataset_abdomenatlas_UFO.py:955); the inflate step has no CPU path."""
    import numpy as np
    import torch
    from rsuper_amd.training.dataset import pack_bits, unpack_bits_device
    from rsuper_amd.hip.lib import RSuperHipError
    x = (np.random.default_rng(0).random((2, 26, 4, 5, 6)) < 0.5)
    p = pack_bits(x)
    assert p.shape == (2, 4, 4, 5, 6) and p.dtype == np.uint8
    assert np.array_equal(p[1], np.packbits(x[1], axis=0))
    assert np.array_equal(np.unpackbits(p[0], axis=0)[:26].astype(bool), x[0])
    with pytest.raises(RSuperHipError):
        unpack_bits_device(torch.from_numpy(p), 26)


def test_chunked_sampler_matches_reference_sequences():
    """ChunkedSampler == the imported reference's index sequences for every (case, rank, epoch) in tests/golden/sampler.npz."""
    import random
    import numpy as np
    from rsuper_amd.training.dataset import ChunkedSampler
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'sampler.npz'))
    for ci, (n, spe, shuffle, seed, world, epochs) in enumerate(g['cases'].tolist()):
        seen = {}
        for rank in range(world):
            s = ChunkedSampler(n, spe, shuffle=bool(shuffle), seed=seed, rank=rank, world_size=world)
            assert len(s) == int(g[f'c{ci}_r{rank}_len'][0])
            for e in range(epochs):
                s.set_epoch(e)
                random.seed(1000 + e)
                got = list(iter(s))
                assert got == g[f'c{ci}_r{rank}_e{e}'].tolist(), (ci, rank, e)
                seen.setdefault(e, []).append(got)
        for e, parts in seen.items():              # ranks partition the epoch chunk round-robin
            assert sum(len(p) for p in parts) == spe


def test_grad_reducer_world2_gloo_cpu():
    """rsuper_amd.reducer.GradReducer (the GPU default behind wrap_ddp) on two gloo ranks: rank-0 broadcast, bucketed mean
    of gradients living in the flat buckets, re-arming across steps, missing-gradient detection."""
    script = os.path.join(ROOT, 'tests', 'reducer_gloo_worker.py')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29537')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29537', script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'REDUCER_OK' in r.stdout


def test_ddp_world2_matches_two_reference_ranks_averaged():
    """SURVEY.md section 8(e) parity definition: "N independent reference ranks averaged".  Two gloo ranks run the tiny training step on their OWN
    batches (oracle as the model on CPU), the product's GradReducer exchanges the gradients, and every parameter's exchanged gradient equals the
    mean of the UNMODIFIED reference's two ranks (tests/golden/ddp2.npz from gen_golden_ddp.py; train_ddp.py:623-668, sampler.py:132)."""
    script = os.path.join(ROOT, 'tests', 'ddp_fixture_gloo_worker.py')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29545', OMP_NUM_THREADS='4')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29545', script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'DDP_FIXTURE_OK' in r.stdout


def test_graph_gradient_exchange_world2_gloo_cpu():
    """rsuper_amd.graph.exchange_gradients (what GraphedNetwork runs after its backward replay in a distributed run) on two gloo ranks:
    in-place mean over several flat buckets."""
    script = os.path.join(ROOT, 'tests', 'exchange_gloo_worker.py')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29541', script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'EXCHANGE_OK' in r.stdout


def test_ops_are_registered_with_the_dispatcher():
    """north_star: "the conv / norm / loss ops are registered as custom HIP ops": every operator of hip/ops.py is a torch.library op
    rsuper::<name> with a schema and kernels at the CUDA / AutogradCUDA keys only -- a CPU tensor finds no kernel (no CPU fallback)."""
    import torch
    from rsuper_amd.hip import ops, library   # noqa: F401  (import registers)
    names = ['basic_block', 'maxpool2', 'upsample_trilinear', 'stem_conv', 'head_conv', 'conv3', 'cl_planar', 'squeeze_excite',
             'bidir_attention', 'channel_norm', 'depthwise_conv3']
    for n in names:
        op = getattr(torch.ops.rsuper, n)
        schema = op.default._schema
        assert schema.name == f'rsuper::{n}' and len(schema.arguments) >= 1 and len(schema.returns) >= 1
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f'rsuper::{n}', 'CUDA')
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f'rsuper::{n}', 'AutogradCUDA')
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f'rsuper::{n}', 'CPU')
    assert str(torch.ops.rsuper.basic_block.default._schema).startswith('rsuper::basic_block(Tensor xa, Tensor mra, Tensor? xb')
    with pytest.raises(NotImplementedError):
        torch.ops.rsuper.maxpool2(torch.zeros(1, 2, 2, 2, 8))
    assert ops.MaxPoolFn.apply.__self__.op is torch.ops.rsuper.maxpool2          # the modules' call sites go through the dispatcher
    # ... and the loss operators (training/losses_foundation.py): fused plane sums + segmentation loss with derivatives, dilation + ball search without
    from rsuper_amd.training import losses_foundation as lf
    for n, autograd in [('plane_partials', True), ('seg_from_sums', True), ('dilate_volume', False), ('ball_search', False)]:
        schema = getattr(torch.ops.rsuper, n).default._schema
        assert schema.name == f'rsuper::{n}'
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f'rsuper::{n}', 'CUDA')
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f'rsuper::{n}', 'AutogradCUDA') == autograd
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f'rsuper::{n}', 'CPU')
    assert lf._PartialsFn.op is torch.ops.rsuper.plane_partials and lf._SegFromSums.op is torch.ops.rsuper.seg_from_sums
    assert ops.dilate_volume is torch.ops.rsuper.dilate_volume and ops.ball_search is torch.ops.rsuper.ball_search
    with pytest.raises(NotImplementedError):
        ops.dilate_volume(torch.zeros(4, 4, 4, dtype=torch.uint8), 3)
