"""Drop-in for the reference's models/det_base_sunrgbd.py: the five-scale SUN-RGBD variant of PointNetDet
(cfgs/det_sample_sunrgbd.yaml: 2048 points, strides 0.1 .. 1.6 over 8 m -> 80 / 40 / 20 / 10 / 5 windows, 10 classes).

Differences from det_base.py, all of them tables for the same HIP kernels (no second code path):
  * PointNetFeat: pointnet1..5 with nsample 128 / 128 / 256 / 256 / 256 and a fifth (256, 256, 512) MLP
    (models/det_base_sunrgbd.py:113-128);
  * ConvFeatNet: block1_conv1 is 64 wide, block5_conv1 / _conv2 / _merge (512) and block5_deconv (512 -> 256, kernel = stride
    = 8) are added (models/det_base_sunrgbd.py:174-251) -- csrc/fcn_net.hip runs it as its 5-level plan (18 layers);
  * heads over 1024 channels, 2 + 67 output columns (models/det_base_sunrgbd.py:271-279) -- logits rows are 128 wide and
    the loss tail runs its NS = 10 instance.
Same class names, constructor signatures and state_dict keys and order (196 entries) as the reference module, so its checkpoints load
unchanged; cfg.MODEL.FILE = 'models/det_base_sunrgbd.py' maps here.
"""
from . import det_base as _base
from .det_base import PointNetModule  # noqa: F401  (same single-scale module)


class PointNetFeat(_base.PointNetFeat):
    """Five scales (reference: models/det_base_sunrgbd.py:107-170)."""

    SCALES = (([64, 64, 128], 128), ([64, 64, 128], 128), ([128, 128, 256], 256), ([256, 256, 512], 256),
              ([256, 256, 512], 256))


class ConvFeatNet(_base.ConvFeatNet):
    """Five-level Conv1d FCN (reference: models/det_base_sunrgbd.py:174-251)."""

    LEVELS = 5
    WIDTHS = (64, 128, 256, 512, 512)
    DECONV_FIRST = 5


class PointNetDet(_base.PointNetDet):
    """Whole pipeline (reference: models/det_base_sunrgbd.py:256-554)."""

    FEAT_NET = PointNetFeat
    CONV_NET = ConvFeatNet
