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
    raise NotImplementedError(f'model {args.model!r} is outside the accelerated hot path (SURVEY.md section 8f lists MedFormer as next)')
