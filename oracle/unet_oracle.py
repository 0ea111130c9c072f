"""ORACLE -- test infrastructure only.  Functional fp32 (torch CPU) restatement of the
reference 3D UNet/ResUNet (`block='BasicBlock'`, `norm='in'`, MaxPool down, trilinear up).

Paths below are relative to /root/reference/rsuper_train/model/dim3/.
Parameters are addressed by the reference's state_dict names (unet.py:31-47):
  inc.conv1.weight, inc.conv2.{conv1,conv2}.conv.weight,
  down{i}.conv.{1,2}.{conv1,conv2,shortcut}.conv.weight   (conv.0 is the MaxPool),
  up{i}.conv.{0,1}.{conv1,conv2,shortcut}.conv.weight, outc.{weight,bias}.
"""
import torch
import torch.nn.functional as F

EPS = 1e-4  # conv_layers.py:40-42: every norm inside ConvNormAct uses eps=1e-4

# bf16 emulation (ORACLE for the bf16 storage mode of the HIP path; the reference itself is fp32-only,
# train_ddp.py:315-316).  When `EMULATE_BF16` is set, tensors are rounded to bfloat16 (RNE) exactly where the kernels
# round: every stored activation, the normalised+activated conv input x_hat, and the conv weights; accumulation stays
# fp32.  `unet_forward(..., emulate_bf16=True)` toggles it.
EMULATE_BF16 = False


def _r(t):
    return t.bfloat16().float() if EMULATE_BF16 else t


def instance_norm(x, eps=EPS):
    """nn.InstanceNorm3d(affine=False, track_running_stats=False): per-(n,c) mean and
    BIASED variance over D*H*W (conv_layers.py:40-42)."""
    m = x.mean(dim=(2, 3, 4), keepdim=True)
    v = ((x - m) ** 2).mean(dim=(2, 3, 4), keepdim=True)
    return (x - m) / torch.sqrt(v + eps)


def conv_norm_act(x, w, stride=1):
    """ConvNormAct(preact=True): conv(relu(norm(x))), conv has no bias (conv_layers.py:46-51)."""
    return F.conv3d(_r(F.relu(instance_norm(x))), _r(w), None, stride=stride, padding=1)


def basic_block(x, p, prefix, stride=1):
    """BasicBlock.forward (conv_layers.py:86-94): conv2(conv1(x)) + shortcut(x); the shortcut
    is a full 3x3x3 ConvNormAct when in_ch != out_ch or stride != 1, identity otherwise."""
    out = _r(conv_norm_act(x, p[prefix + '.conv1.conv.weight'], stride))
    out = conv_norm_act(out, p[prefix + '.conv2.conv.weight'])
    ks = prefix + '.shortcut.conv.weight'
    sc = _r(conv_norm_act(x, p[ks], stride)) if ks in p else x
    return _r(out + sc)


def upsample_trilinear_ac(x, size):
    """F.interpolate(mode='trilinear', align_corners=True) (unet_utils.py:69), written as three
    separable linear interpolations with src = dst*(I-1)/(O-1)."""
    for axis, O in zip((2, 3, 4), size):
        I = x.shape[axis]
        if I == O:
            continue
        scale = (I - 1) / (O - 1) if O > 1 else 0.0
        src = torch.arange(O, dtype=torch.float32) * scale
        i0 = src.floor().long().clamp(max=I - 1)
        i1 = (i0 + 1).clamp(max=I - 1)
        w1 = (src - i0.float())
        shape = [1] * 5
        shape[axis] = O
        w1 = w1.view(shape)
        x = x.index_select(axis, i0) * (1 - w1) + x.index_select(axis, i1) * w1
    return x


def unet_forward(p, x, pool=True, emulate_bf16=False):
    """UNet.forward (unet.py:50-64).  p: dict name -> tensor.  Returns logits (B,C,D,H,W)."""
    global EMULATE_BF16
    old, EMULATE_BF16 = EMULATE_BF16, emulate_bf16
    try:
        return _unet_forward(p, x, pool)
    finally:
        EMULATE_BF16 = old


def _unet_forward(p, x, pool=True):
    def down(x, name):
        if pool:
            x = F.max_pool3d(x, 2)                         # unet_utils.py:35-37
            x = basic_block(x, p, name + '.conv.1')
            return basic_block(x, p, name + '.conv.2')
        x = basic_block(x, p, name + '.conv.0', stride=2)  # unet_utils.py:38-39
        return basic_block(x, p, name + '.conv.1')

    def up(x1, x2, name):
        x1 = _r(upsample_trilinear_ac(x1, x2.shape[2:]))   # unet_utils.py:69
        x = torch.cat([x2, x1], dim=1)                     # unet_utils.py:71
        x = basic_block(x, p, name + '.conv.0')
        return basic_block(x, p, name + '.conv.1')

    x1 = _r(F.conv3d(x, p['inc.conv1.weight'], None, padding=1))   # unet_utils.py:14 (no norm/act; f32 weights in the stem)
    x1 = basic_block(x1, p, 'inc.conv2')
    x2 = down(x1, 'down1')
    x3 = down(x2, 'down2')
    x4 = down(x3, 'down3')
    x5 = down(x4, 'down4')
    o = up(x5, x4, 'up1')
    o = up(o, x3, 'up2')
    o = up(o, x2, 'up3')
    o = up(o, x1, 'up4')
    return F.conv3d(o, p['outc.weight'], p['outc.bias'])       # unet.py:47 (f32 weights, f32 logits)


def unet_param_shapes(in_ch, base_ch, num_classes, pool=True):
    """state_dict shapes of UNet(in_ch, base_ch, num_classes, block='BasicBlock', pool=pool) (unet.py:31-47;
    bottleneck is 10*base_ch).  pool=False: no pooling layer in down_block.conv, the strided first block sits at index 0
    and always has a convolutional shortcut (unet_utils.py:38-39, conv_layers.py:82-84)."""
    b = base_ch
    s = {'inc.conv1.weight': (b, in_ch, 3, 3, 3)}

    def blk(prefix, ci, co):
        s[prefix + '.conv1.conv.weight'] = (co, ci, 3, 3, 3)
        s[prefix + '.conv2.conv.weight'] = (co, co, 3, 3, 3)
        if ci != co:
            s[prefix + '.shortcut.conv.weight'] = (co, ci, 3, 3, 3)

    blk('inc.conv2', b, b)
    chans = [b, 2 * b, 4 * b, 8 * b, 10 * b]
    for i in range(4):
        o = 1 if pool else 0
        blk(f'down{i + 1}.conv.{o}', chans[i], chans[i + 1])
        blk(f'down{i + 1}.conv.{o + 1}', chans[i + 1], chans[i + 1])
    for i in range(4):
        ci, co = chans[4 - i], chans[3 - i]
        blk(f'up{i + 1}.conv.0', ci + co, co)
        blk(f'up{i + 1}.conv.1', co, co)
    s['outc.weight'] = (num_classes, b, 1, 1, 1)
    s['outc.bias'] = (num_classes,)
    return s
