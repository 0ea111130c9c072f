"""TEST INFRASTRUCTURE -- CPU restatement of the reference MedFormer forward pass, driven by a state_dict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.  Plain torch fp32/fp64 on the CPU; every function
cites the reference lines it restates (paths relative to rsuper_train/model/dim3/).  Pinned by tests/golden/medformer.npz, which
tests/golden/gen_golden_medformer.py produces from the imported, unmodified reference class.

Configuration (`cfg`): the MedFormer constructor arguments that are not recoverable from tensor shapes -- map_size, conv_num,
trans_num, num_heads, fusion_depth, fusion_heads, aux_loss (medformer.py:83-110).  Activation is ReLU and the norm InstanceNorm3d, as
in config/abdomenatlas_ufo/medformer_3d.yaml.
"""
import torch
import torch.nn.functional as F

IN_EPS_CNA = 1e-4     # ConvNormAct builds norm(ch, eps=1e-4) (conv_layers.py:40-43)
IN_EPS = 1e-5         # bare norm(dim) elsewhere (medformer_utils.py:117-118,160): nn.InstanceNorm3d default


def _has(sd, prefix):
    return any(k.startswith(prefix) for k in sd)


def conv_norm_act(sd, pre, x, act=True, norm=True, groups=1):
    """ConvNormAct(preact=True): conv(act(norm(x))) (conv_layers.py:46-51); padding = kernel // 2, no bias."""
    w = sd[pre + '.conv.weight']
    h = F.instance_norm(x, eps=IN_EPS_CNA) if norm else x
    h = F.relu(h) if act else h
    return F.conv3d(h, w, None, 1, w.shape[-1] // 2, 1, groups)


def basic_block(sd, pre, x):
    """BasicBlock (conv_layers.py:71-94)."""
    out = conv_norm_act(sd, pre + '.conv1', x)
    out = conv_norm_act(sd, pre + '.conv2', out)
    return out + (conv_norm_act(sd, pre + '.shortcut', x) if _has(sd, pre + '.shortcut.') else x)


def ds_conv(sd, pre, x):
    """DepthwiseSeparableConv (conv_layers.py:126-157): per-channel 3x3x3 then 1x1x1, no bias."""
    wd = sd[pre + '.depthwise.weight']
    return F.conv3d(F.conv3d(x, wd, None, 1, wd.shape[-1] // 2, 1, wd.shape[0]), sd[pre + '.pointwise.weight'])


def patch_merging(sd, pre, x):
    """PatchMerging (medformer_utils.py:142-178): the 8 parity sub-lattices concatenated on channels (i, j, k nested, k fastest),
    InstanceNorm, depthwise-separable reduction."""
    parts = [x[:, :, i::2, j::2, k::2] for i in range(2) for j in range(2) for k in range(2)]
    return ds_conv(sd, pre + '.reduction', F.instance_norm(torch.cat(parts, 1), eps=IN_EPS))


def se_block(sd, pre, x):
    """SEBlock (conv_layers.py:159-174): global mean -> 1x1 (+bias) -> ReLU -> 1x1 (+bias) -> sigmoid -> scale."""
    s = x.mean((2, 3, 4), keepdim=True)
    s = F.relu(F.conv3d(s, sd[pre + '.excitation.0.weight'], sd[pre + '.excitation.0.bias']))
    return x * torch.sigmoid(F.conv3d(s, sd[pre + '.excitation.2.weight'], sd[pre + '.excitation.2.bias']))


def mb_conv(sd, pre, x):
    """MBConv with in_ch == out_ch, stride 1 (conv_layers.py:198-240): expand 1x1, depthwise 3x3x3, SE, project 1x1 (no act),
    identity shortcut."""
    h = conv_norm_act(sd, pre + '.expand_proj', x) if _has(sd, pre + '.expand_proj.') else x
    h = conv_norm_act(sd, pre + '.depthwise', h, groups=sd[pre + '.depthwise.conv.weight'].shape[0])
    h = se_block(sd, pre + '.se', h)
    h = conv_norm_act(sd, pre + '.pointwise', h, act=False)
    assert not _has(sd, pre + '.shortcut.'), 'MBConv with a projection shortcut is not on the MedFormer path'
    return h + x


def _heads_split(t, heads):
    """'b (dim_head heads) d h w -> b heads (d h w) dim_head' (medformer_utils.py:46-55)."""
    b, c = t.shape[:2]
    return t.reshape(b, c // heads, heads, -1).permute(0, 2, 3, 1)


def _heads_merge(t, dhw):
    """'b heads (d h w) dim_head -> b (dim_head heads) d h w' (medformer_utils.py:56-63)."""
    b, heads, L, dh = t.shape
    return t.permute(0, 3, 1, 2).reshape(b, heads * dh, *dhw)


def bidirection_attention(sd, pre, feat, smap, heads):
    """BidirectionAttention.forward (medformer_utils.py:67-99): one score matrix, softmax over map tokens for the feature
    update and over voxels for the map update."""
    fq, fv = ds_conv(sd, pre + '.feat_qv', feat).chunk(2, 1)
    mq, mv = F.conv3d(smap, sd[pre + '.map_qv.weight']).chunk(2, 1)
    dim_head = fq.shape[1] // heads
    fq, fv, mq, mv = (_heads_split(t, heads) for t in (fq, fv, mq, mv))
    attn = torch.einsum('bhid,bhjd->bhij', fq, mq) * dim_head ** -0.5
    f_out = _heads_merge(torch.einsum('bhij,bhjd->bhid', F.softmax(attn, -1), mv), feat.shape[2:])
    m_out = _heads_merge(torch.einsum('bhji,bhjd->bhid', F.softmax(attn, -2), fv), smap.shape[2:])
    f_out = ds_conv(sd, pre + '.feat_out', f_out)
    if (pre + '.map_out.weight') in sd:
        m_out = F.conv3d(m_out, sd[pre + '.map_out.weight'])
    return f_out, m_out


def bidirection_block(sd, pre, x, smap, heads):
    """BidirectionAttentionBlock.forward (medformer_utils.py:131-144)."""
    out, m = bidirection_attention(sd, pre + '.attn', F.instance_norm(x, eps=IN_EPS), F.instance_norm(smap, eps=IN_EPS), heads)
    out = out + (conv_norm_act(sd, pre + '.shortcut', x) if _has(sd, pre + '.shortcut.') else x)
    return mb_conv(sd, pre + '.feedforward', out), m + smap


def trans_blocks(sd, pre, x, smap, n, heads):
    for i in range(n):
        x, smap = bidirection_block(sd, f'{pre}.blocks.{i}', x, smap, heads)
    return x, smap


def map_generation(sd, pre, x, map_size):
    """SemanticMapGeneration.forward (medformer_utils.py:222-236): softmax over voxels of the code logits, weighted sums."""
    B = x.shape[0]
    feat = F.conv3d(x, sd[pre + '.base_proj.weight'], padding=1).flatten(2)                       # B, map_dim, L
    wmap = F.softmax(F.conv3d(x, sd[pre + '.semantic_proj.weight'], padding=1).flatten(2), dim=2)   # B, codes, L
    return torch.einsum('bij,bkj->bik', feat, wmap).reshape(B, feat.shape[1], *map_size)


def transformer(sd, pre, x, depth, heads):
    """TransformerBlock (trans_layers.py:107-125): pre-LayerNorm attention and GELU MLP, residual each."""
    for i in range(depth):
        a, m = f'{pre}.layers.{i}.0', f'{pre}.layers.{i}.1'
        h = F.layer_norm(x, x.shape[-1:], sd[a + '.norm.weight'], sd[a + '.norm.bias'])
        q, k, v = F.linear(h, sd[a + '.fn.to_qkv.weight']).chunk(3, -1)
        B, L, n = q.shape
        q, k, v = (t.reshape(B, L, heads, n // heads).permute(0, 2, 1, 3) for t in (q, k, v))
        att = F.softmax(torch.einsum('bhid,bhjd->bhij', q, k) * (n // heads) ** -0.5, -1)
        o = torch.einsum('bhij,bhjd->bhid', att, v).permute(0, 2, 1, 3).reshape(B, L, n)
        x = F.linear(o, sd[a + '.fn.to_out.weight'], sd[a + '.fn.to_out.bias']) + x
        h = F.layer_norm(x, x.shape[-1:], sd[m + '.norm.weight'], sd[m + '.norm.bias'])
        h = F.linear(F.gelu(F.linear(h, sd[m + '.fn.fc1.weight'], sd[m + '.fn.fc1.bias'])), sd[m + '.fn.fc2.weight'], sd[m + '.fn.fc2.bias'])
        x = h + x
    return x


def map_fusion(sd, pre, maps, depth, heads):
    """SemanticMapFusion.forward (medformer_utils.py:259-273): project to a common width, one token sequence, transformer, project back."""
    B, _, D, H, W = maps[0].shape
    toks = [F.conv3d(m, sd[f'{pre}.in_proj.{i}.weight']).flatten(2).permute(0, 2, 1) for i, m in enumerate(maps)]
    dim = toks[0].shape[-1]
    out = transformer(sd, pre + '.fusion', torch.cat(toks, 1), depth, heads).chunk(len(maps), 1)
    return [F.conv3d(o.permute(0, 2, 1).reshape(B, dim, D, H, W), sd[f'{pre}.out_proj.{i}.weight']) for i, o in enumerate(out)]


def down_block(sd, pre, x, conv_num, trans_num, heads, map_size, map_generate):
    """down_block.forward (medformer_utils.py:320-335)."""
    out = patch_merging(sd, pre + '.patch_merging', x)
    for i in range(conv_num):
        out = basic_block(sd, f'{pre}.conv_blocks.{i}', out)
    smap = map_generation(sd, pre + '.map_gen', out, map_size) if map_generate else None
    return trans_blocks(sd, pre + '.trans_blocks', out, smap, trans_num, heads)


def up_block(sd, pre, x1, x2, map1, map2, conv_num, trans_num, heads):
    """up_block.forward (medformer_utils.py:372-393)."""
    feat = torch.cat([F.interpolate(x1, size=x2.shape[-3:], mode='trilinear', align_corners=True), x2], 1)
    smap = map1
    if (pre + '.map_reduction.weight') in sd and map2 is not None:
        smap = F.conv3d(torch.cat([map1, map2], 1), sd[pre + '.map_reduction.weight'])
    out, smap = trans_blocks(sd, pre + '.trans_blocks', feat, smap, trans_num, heads)
    for i in range(conv_num):
        out = basic_block(sd, f'{pre}.conv_blocks.{i}', out)
    return out, smap


def medformer_forward(sd, x, cfg):
    """MedFormer.forward (medformer.py:176-203) -> [final logits, aux logits] when cfg['aux_loss'] else the final logits."""
    cn, tn, nh, ms = cfg['conv_num'], cfg['trans_num'], cfg['num_heads'], cfg['map_size']
    x0 = basic_block(sd, 'inc.conv2', F.conv3d(x, sd['inc.conv1.weight'], padding=1))           # inconv (medformer_utils.py:277-291)
    x1, _ = down_block(sd, 'down1', x0, cn[0], tn[0], nh[0], ms, False)
    x2, m2 = down_block(sd, 'down2', x1, cn[1], tn[1], nh[1], ms, True)
    x3, m3 = down_block(sd, 'down3', x2, cn[2], tn[2], nh[2], ms, True)
    x4, m4 = down_block(sd, 'down4', x3, cn[3], tn[3], nh[3], ms, True)
    maps = map_fusion(sd, 'map_fusion', [m2, m3, m4], cfg['fusion_depth'], cfg['fusion_heads'])
    out, smap = up_block(sd, 'up1', x4, x3, maps[2], maps[1], cn[4], tn[4], nh[4])
    out, smap = up_block(sd, 'up2', out, x2, smap, maps[0], cn[5], tn[5], nh[5])
    aux = None
    if cfg.get('aux_loss'):
        aux = F.interpolate(F.conv3d(out, sd['aux_out.weight'], sd['aux_out.bias']), size=x.shape[-3:], mode='trilinear', align_corners=True)
    out, smap = up_block(sd, 'up3', out, x1, smap, None, cn[6], tn[6], nh[6])
    out, smap = up_block(sd, 'up4', out, x0, smap, None, cn[7], tn[7], nh[7])
    out = F.conv3d(out, sd['outc.weight'], sd['outc.bias'])
    return [out, aux] if cfg.get('aux_loss') else out
