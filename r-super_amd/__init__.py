"""rsuper_amd -- MI355X-native (gfx950) implementation of R-Super's training hot path
(rsuper_train/train_ddp.py step) behind the reference's model/ + training/ interfaces.

Sub-packages mirror the reference layout (rsuper_train/):
    rsuper_amd.model.utils.get_model            <- model/utils.py:11
    rsuper_amd.model.dim3.unet.UNet             <- model/dim3/unet.py:12
    rsuper_amd.training.losses_foundation       <- training/losses_foundation.py
    rsuper_amd.training.utils                   <- training/utils.py
    rsuper_amd.train_ddp                        <- train_ddp.py (train_epoch step, DDP worker)
    rsuper_amd.hip                              <- ctypes binding of csrc/librsuper_hip.so (C ABI: include/rsuper_hip.h)
"""
__version__ = '0.1.0'
