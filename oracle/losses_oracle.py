"""ORACLE -- test infrastructure only.  CPU restatement of the R-Super losses
(/root/reference/rsuper_train/training/losses_foundation.py; line numbers below refer to it).

Float arithmetic: torch CPU fp32 with autograd (so gradients w.r.t. logits exist);
integer/byte arithmetic (ball dilation, top-k, ranks): oracle/morph.c.
Tie rule for selection ops: value descending, then linear index ascending (the
reference's torch.topk / torch.sort tie order is implementation-defined).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import morph


# --------------------------------------------------------------------------- helpers
def _u8(t):
    return np.ascontiguousarray(t.detach().cpu().numpy() > 0, dtype=np.uint8)


def dilate(t, k):
    """dilate_volume (:22-46) on a float 0/1 tensor; returns float tensor."""
    return torch.from_numpy(morph.dilate_volume(_u8(t), k)).to(torch.float32)


def lesion_groups(classes):
    """get_lesion_channels (:204-221): channels whose name contains lesion/cyst/pdac/pnet,
    grouped by organ prefix ('pancreatic' -> 'pancreas'); insertion order preserved."""
    groups = {}
    for i, c in enumerate(classes):
        for suffix in ['lesion', 'cyst', 'pdac', 'pnet']:
            if suffix in c:
                name = c[:c.index('_' + suffix) + len('_' + suffix)].replace('pancreatic', 'pancreas')
                groups.setdefault(name, []).append(i)
    return groups


def lesion_channels(t, classes):
    """(:223-228) max-merge sub-channels per organ -> (B, L, ...)."""
    g = lesion_groups(classes)
    return torch.stack([torch.stack([t[:, i] for i in idx], 0).max(0).values for idx in g.values()], dim=1)


def known_voxels(unk, dilation=5):
    """get_known_voxels (:150-165): 1 - dilate(unk, 5)."""
    return 1.0 - dilate(unk.float(), dilation)


# --------------------------------------------------------------------------- Dice / volume
def dice_loss_multiclass(preds, targets, known, class_weights=None):
    """DiceLossMultiClass (:541-607), sigmoid=True.  Adaptive Tversky; alpha is NOT detached."""
    while preds.dim() < 5:   # 3D inputs get two leading dims, 4D one (:543-553)
        preds, targets, known = preds.unsqueeze(0), targets.unsqueeze(0), known.unsqueeze(0)
    N, C = preds.shape[:2]
    P = torch.sigmoid(preds) * known
    T = targets * known
    TP, FP, FN = P * T, P * (1 - T), (1 - P) * T
    fp_c = FP.transpose(0, 1).reshape(C, -1).sum(1)
    fn_c = FN.transpose(0, 1).reshape(C, -1).sum(1)
    alpha = (fp_c / (fp_c + fn_c + 1e-5)).unsqueeze(0).repeat(N, 1).clamp(0.2, 0.8)
    num = TP.sum((-1, -2, -3))
    den = num + alpha * FP.sum((-1, -2, -3)) + (1 - alpha) * FN.sum((-1, -2, -3))
    loss = 1 - num / (den + 1e-5)
    if class_weights is not None:
        cw = class_weights.mean(dim=(-1, -2, -3))
        while cw.dim() < loss.dim():
            cw = cw.unsqueeze(0)
        loss = loss * cw
    return loss.mean()


def dice_based_volume_loss(x, y, tolerance=0.1, E=500.0):
    """(:352-395) |x-y|/(x+y+E) minus its value at the tolerance edge, clamped to [0,1]."""
    loss = torch.abs(x - y) / (x + y + E)
    v = torch.max((1 - tolerance) * y, y.clamp(max=100))
    loss = loss - torch.abs(v - y) / (v + y + E)
    return loss.clamp(0, 1)


def volume_loss_basic(out, mask, volumes, labels, unk, classes, tolerance, class_weights=None):
    """(:250-349).  class_weights: (B,C,1,1,1) or None."""
    x = torch.sigmoid(lesion_channels(out, classes))
    m = lesion_channels(mask, classes)
    t = lesion_channels(labels, classes)
    M = dilate(m, 31)                                                  # :308
    per_voxel_pos = (t.sum((-1, -2, -3), keepdim=True) > 0).float()    # :313
    x = x * (1 - per_voxel_pos)
    vhat = (x * M).sum((-1, -2, -3))                                   # :329, :367
    rv = volumes.sum(-1).unsqueeze(-1).repeat(1, M.shape[1])           # :333-334
    rv = rv * (M.sum((-1, -2, -3)) > 0).float()                        # :335-337
    loss = dice_based_volume_loss(vhat, rv, tolerance=tolerance, E=500.0)
    if class_weights is not None:
        cw = lesion_channels(class_weights.expand(-1, -1, 1, 1, 1), classes).mean(dim=(-1, -2, -3))
        loss = loss * cw
    return loss.mean()


# --------------------------------------------------------------------------- ball machinery
def _odd_ceil(v):
    c = math.ceil(v)
    return c + 1 if c % 2 == 0 else c


def create_ball_kernel(diameter, gaussian=False, gaussian_std=1.5):
    """(:1161-1230)."""
    d_odd = _odd_ceil(diameter)
    ks = _odd_ceil(1.2 * d_odd)
    radius = d_odd / 2.0
    c = torch.arange(ks, dtype=torch.float32) - (ks - 1) / 2.0
    d2 = c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2
    mask = (d2 <= radius ** 2).float()
    if not gaussian:
        return mask
    std = gaussian_std * radius
    k = torch.exp(-d2 / (2.0 * std ** 2)) * mask
    return k / k.sum()


def insert_ball(shape, center, diameter, margin):
    """(:1336-1385) binary ball of diameter*(1+margin) pasted at `center`, clipped at the borders."""
    k = create_ball_kernel(diameter * (1 + margin))
    out = torch.zeros(shape)
    half = k.shape[-1] // 2
    sl_v, sl_k = [], []
    for c, n in zip(center, shape):
        lo, hi = max(0, c - half), min(n, c + half + 1)
        klo = 0 if c - half >= 0 else -(c - half)
        sl_v.append(slice(lo, hi))
        sl_k.append(slice(klo, klo + (hi - lo)))
    out[tuple(sl_v)] = k[tuple(sl_k)]
    return out


def isolate_tumor(x, diameter, tumor_volume, diameter_margin=0.2, volume_margin=0.2, gaussian_std=1.5):
    """(:1387-1532).  x: (D,H,W) non-negative.  Returns (mask, mask_small, mask_big, center)."""
    diameter = int(np.round(diameter).astype(int))
    vol = int(np.round(tumor_volume).astype(int))
    if diameter % 2 == 0:
        diameter += 1
    K = create_ball_kernel(diameter, True, gaussian_std)
    nnz = int((K > 0).sum())
    f32 = False
    if nnz > vol:                      # :1431-1433; vol becomes a 0-dim int64 tensor in the reference,
        vol, f32 = nnz - 1, True       # so later int(t*(1-m)) products are evaluated in float32
    conv = F.conv3d(x[None, None], K[None, None], padding=K.shape[-1] // 2)[0, 0]
    center = np.unravel_index(int(torch.argmax(conv)), conv.shape)      # first maximum
    ball = insert_ball(x.shape, center, diameter, diameter_margin)
    new_dim = diameter
    while float(ball.sum()) < vol:                                       # :1450-1461
        old = new_dim
        new_dim = int(np.round(new_dim * 1.1))
        if old == new_dim:
            new_dim += 1
        if new_dim % 2 == 0:
            new_dim += 1
        if new_dim >= max(x.shape):
            break
        ball = insert_ball(x.shape, center, new_dim, diameter_margin)
    flat = (x * ball).reshape(-1)
    L = flat.numel()
    fl = (lambda v: float(np.float32(v))) if f32 else float
    t = min(L - 1, vol)
    ms = min(0.5, volume_margin)
    t_small = int(np.float32(t) * np.float32(1 - ms)) if f32 else int(t * (1 - ms))
    t_small = max(t_small, min(100, vol))
    t_big = min(L - 1, int(np.float32(vol) * np.float32(1 + volume_margin)) if f32 else int(vol * (1 + volume_margin)))
    ballf = ball.reshape(-1)
    flat_np = flat.detach().numpy()
    masks = [torch.from_numpy(morph.topk_mask(flat_np, k)).float() * ballf for k in (t, t_small, t_big)]
    masks = [m.view(x.shape) for m in masks]
    iters = 0
    while vol < 50 ** 3 and float(masks[0].sum()) < vol * 0.7:           # :1513-1522
        if iters > 5:
            break
        masks = [dilate(m, 7) * ball for m in masks]
        iters += 1
    return masks[0], masks[1], masks[2], center


def gwrp_weights(x, N, c=0.5):
    """GlobalWeightedRankPooling(return_weights=True, hard_cutoff=True) (:442-535) on one (D,H,W) map."""
    L = x.numel()
    Nf = torch.clamp(torch.tensor(float(N), dtype=torch.float32), min=1)
    d = (1 - c) ** (1.0 / Nf)
    idx = torch.arange(L, dtype=torch.float32)
    raw = d ** idx
    w = raw / raw.sum()
    w = w * (idx < Nf).float()
    w = w / w.sum()
    rank = torch.from_numpy(morph.rank_desc(x.detach().reshape(-1).numpy()))
    return w[rank].view(x.shape)


def ball_loss(out, labels, unk, mask, volumes, diameters, classes, apply_dice_loss, margin=0.2,
              class_weights=None, standard_ce=False):
    """(:1537-1864).  class_weights: (B,C,1,1,1) or None.  Returns (bce, dice, debug)."""
    B = out.shape[0]
    x_l = lesion_channels(out, classes)
    m_l = lesion_channels(mask, classes)
    u_l = lesion_channels(unk, classes)
    t_l = lesion_channels(labels, classes)
    cw_l = None
    if class_weights is not None:
        cw_l = lesion_channels(class_weights.expand(-1, -1, *out.shape[2:]), classes)   # (B,L,D,H,W); :1583 quirk is harmless
    Mseg = dilate(m_l, 31)
    u_d = dilate(u_l, 1)                                                  # identity (:1596)
    pen = (((1 - u_d) * (1 - t_l)) + Mseg > 0).float()                    # :1597-1605
    losses, dices, debug = [], [], []
    for b in range(B):
        x, seg = x_l[b], Mseg[b]
        if float(seg.sum()) == 0 or float(volumes[b].sum()) == 0:       # :1625-1661
            zero = torch.zeros_like(x)
            l = F.binary_cross_entropy_with_logits(x, zero, reduction='none') * pen[b]
            if cw_l is not None:
                l = l * cw_l[b]
            losses.append(l.mean())
            if apply_dice_loss:
                dices.append(dice_loss_multiclass(x, zero, pen[b], None if cw_l is None else cw_l[b]))
            debug.append(None)
            continue
        c = [i for i in range(x.shape[0]) if float(seg[i].sum()) > 0][0]
        xc, penal = x[c], pen[b][c]
        cwt = None if cw_l is None else cw_l[b][c]
        tumor_seg = seg.sum(0)
        vols = volumes[b]
        order = [int(i) for i in np.argsort(-vols.numpy(), kind='stable') if float(vols[int(i)]) > 0]
        x_it = torch.sigmoid(xc).detach() * tumor_seg
        pms, pmb = [], []
        for ti in order:                                                  # :1695-1719
            vol = float(vols[ti])
            dmax = float(diameters[b, ti].max())
            if dmax <= 1:
                dmax = 3
            if vol <= 1:
                vol = 9
            pm, pm_s, pm_b, ctr = isolate_tumor(x_it, dmax, vol, margin, margin)
            pms.append(pm_s)
            pmb.append(pm_b)
            x_it = x_it * (1 - pm)
        PM = (torch.stack(pms).sum(0) > 0).float()
        BIG = dilate((torch.stack(pmb).sum(0) > 0).float(), 7)            # :1727-1731
        border = ((BIG - PM) > 0).float()
        penal = penal * (1 - border)
        bce = F.binary_cross_entropy_with_logits(xc, PM, reduction='none') * penal
        if apply_dice_loss:
            dices.append(dice_loss_multiclass(xc, PM, penal, cwt))
        if not standard_ce:
            fw = gwrp_weights(torch.sigmoid(xc).detach() * PM + PM, float(PM.sum()), 0.5) * PM.sum() * PM
            lfg, lbg = bce * fw, bce * (1 - BIG)
            if cwt is not None:
                lfg, lbg = lfg * cwt, lbg * cwt
            losses.append(lfg.mean() + lbg.mean())
        else:
            if cwt is not None:
                bce = bce * cwt
            losses.append(bce.mean())
        debug.append(dict(PM=PM, BIG=BIG, penal=penal))
    l_bce = torch.stack(losses).mean()
    l_dice = torch.stack(dices).mean() if apply_dice_loss else torch.zeros_like(l_bce)
    return l_bce, l_dice, debug


# --------------------------------------------------------------------------- calculate_loss
def calculate_loss(model_output, label, unk_voxels, args, chosen_segment_mask, tumor_volumes_report,
                   tumor_diameters, classes, class_weights=None):
    """calculate_loss (:685-1076), segmentation + report terms (no classification/clip/genesis branches)."""
    result = model_output['segmentation']
    label = label.float()
    unk = unk_voxels.float()
    mask = chosen_segment_mask.float()
    if class_weights is not None and torch.equal(class_weights, torch.ones_like(class_weights)):
        class_weights = None
    cw5 = None if class_weights is None else class_weights[:, :, None, None, None].float()
    known = known_voxels(unk, 5)                                          # :899 / :990
    deep = isinstance(result, (tuple, list))
    heads = list(result) if deep else [result]
    seg_total, rep = 0, {}
    rep_scalar = 0
    for j, r in enumerate(heads):
        aw = args.aux_weight[j] if deep else 1.0
        loss_r = None
        if args.report_volume_loss_basic > 0:
            use_ball = ('ball' in args.loss or 'dynamic' in args.loss or 'dll' in args.loss)
            if deep:
                use_ball = use_ball and not (j != 0 and 'last' in args.loss)      # :924
            if use_ball:
                bce, dice, _ = ball_loss(r, label, unk, mask, tumor_volumes_report, tumor_diameters, classes,
                                         apply_dice_loss=('dice' in args.loss), margin=args.ball_volume_margin,
                                         class_weights=cw5, standard_ce=args.stardard_ce_ball)
                loss_r = {'ball_loss_bce': bce, 'ball_loss_dice': dice}
                if 'both' in args.loss:
                    loss_r['dice_volume_loss'] = volume_loss_basic(r, mask, tumor_volumes_report, label, unk, classes,
                                                                   args.volume_loss_tolerance, cw5)
            else:
                loss_r = {'dice_volume_loss': volume_loss_basic(r, mask, tumor_volumes_report, label, unk, classes,
                                                                args.volume_loss_tolerance, cw5)}
        bce = F.binary_cross_entropy_with_logits(r, label, reduction='none', weight=cw5)      # :945
        seg = (bce * known).mean() + dice_loss_multiclass(r, label, known, cw5)              # :955-956
        seg_total = seg_total + aw * args.seg_loss * seg
        if loss_r is None:
            rep_scalar = rep_scalar + aw * args.report_volume_loss_basic * torch.tensor(0.0)
        else:
            for k, v in loss_r.items():
                w = {'ball_loss_bce': args.ball_bce_weight, 'ball_loss_dice': args.ball_dice_weight}.get(k, 1)
                term = aw * args.report_volume_loss_basic * w * v
                rep[k] = rep[k] + term if k in rep else term
    loss = {'segmentation': seg_total}
    if rep:
        loss.update(rep)
    else:
        loss['report'] = rep_scalar
    overall = 0
    for k in list(loss.keys()):
        overall = overall + loss[k]
    loss['overall'] = overall
    if torch.isnan(overall).any():
        raise ValueError('loss is nan, propagating this can destroy the network weights, STOP!')
    return loss
