"""MI355X-native hot path of Frustum ConvNet: drop-in modules over a C-ABI of hand-written gfx950 kernels.

    from frustum_convnet_amd.det_base import PointNetDet            # same ctor / forward / state_dict as the reference
    from frustum_convnet_amd.query_depth_point import QueryDepthPoint
    from frustum_convnet_amd.config import cfg, merge_cfg_from_file
    from frustum_convnet_amd.train_state import FlatTrainState       # flat parameters / gradients / Adam moments
    from frustum_convnet_amd.inputs import InputBuilder              # batch construction on the device

Nothing here imports torch or loads libfcn_hip.so eagerly: `_native.lib()` does on first use and raises ImportError
when the library has not been built (`python -m frustum_convnet_amd.build`); there is no CPU fallback.
"""
__version__ = "0.1.0"
TARGET_ARCH = "gfx950"

__all__ = ["__version__", "TARGET_ARCH"]
