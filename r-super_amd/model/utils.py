"""Model factory with the signature of rsuper_train/model/utils.py:11 (`get_model(args, pretrain, classes, classes_cls)`)."""


def get_model(args, pretrain=False, classes=None, classes_cls=None):
    if args.dimension != '3d':
        raise NotImplementedError('gfx950 hot path: 3d models only (R-Super trains 3D volumes)')
    if args.model in ('unet', 'resunet'):          # model/utils.py:81-89
        if pretrain:
            raise ValueError('No pretrain model available')
        from .dim3.unet import UNet
        return UNet(args.in_chan, args.base_chan, num_classes=args.classes, scale=args.down_scale, norm=args.norm,
                    kernel_size=args.kernel_size, block=args.block,
                    compute_dtype=getattr(args, 'compute_dtype', None))
    if args.model == 'medformer':                  # model/utils.py:97-126
        if pretrain:
            raise ValueError('No pretrain model available')
        from .dim3.medformer import MedFormer
        n_cls = args.classes if classes is None else len(classes)
        return MedFormer(args.in_chan, n_cls, args.base_chan, map_size=args.map_size, conv_block=args.conv_block, conv_num=args.conv_num,
                         trans_num=args.trans_num, num_heads=args.num_heads, fusion_depth=args.fusion_depth, fusion_dim=args.fusion_dim,
                         fusion_heads=args.fusion_heads, expansion=args.expansion, attn_drop=args.attn_drop, proj_drop=args.proj_drop,
                         proj_type=args.proj_type, norm=args.norm, act=args.act, kernel_size=args.kernel_size, scale=args.down_scale,
                         aux_loss=args.aux_loss, classification_branch=getattr(args, 'classification_branch', False),
                         compute_dtype=getattr(args, 'compute_dtype', None))     # like the reference, the YAML's chan_num is NOT passed
    raise NotImplementedError(f'model {args.model!r} is outside the accelerated hot path (unet / resunet / medformer)')
