"""Drop-in for the reference module models/seg_hrnet.py: 3-channel crops -> 32 heatmaps
(models/seg_hrnet.py:265,324).  `from esa_pose_estimation_amd import seg_hrnet;
net = seg_hrnet.get_seg_model(config)` replaces `from models import seg_hrnet; ...`."""
from .hrnet import HighResolutionNet as _Base


class HighResolutionNet(_Base):
    CIN, NUM_KEYPOINTS = 3, 32


def get_seg_model(cfg, **kwargs):
    model = HighResolutionNet(cfg, **kwargs)
    model.init_weights(cfg.MODEL.PRETRAINED)
    return model
