"""Drop-in for the reference module models/seg_hrnet2.py: 1-channel crops -> 11 heatmaps
(models/seg_hrnet2.py:265,324) — the SPEED 11-keypoint configuration BASELINE.json names."""
from .hrnet import HighResolutionNet as _Base


class HighResolutionNet(_Base):
    CIN, NUM_KEYPOINTS = 1, 11


def get_seg_model(cfg, **kwargs):
    model = HighResolutionNet(cfg, **kwargs)
    model.init_weights(cfg.MODEL.PRETRAINED)
    return model
