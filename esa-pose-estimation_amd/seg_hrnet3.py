"""Drop-in for the reference module models/seg_hrnet3.py — the network val.py:380 / demo.py:411 /
main.py:244 actually instantiate: 1-channel crops -> 30 heatmaps, CBAM channel + spatial attention in
every BasicBlock (seg_hrnet3.py:32-61, 90-91) and on the 64-channel pre-BN stem skip (:516-517), 3x3
last_layer[0] (:363), output_layer over [heat-maps, skip] (:381-382).  SURVEY.md §8a row a18."""
from .hrnet import HighResolutionNet as _Base


class HighResolutionNet(_Base):
    CIN, NUM_KEYPOINTS, VARIANT = 1, 30, 1


def get_seg_model(cfg, **kwargs):
    model = HighResolutionNet(cfg, **kwargs)
    model.init_weights(cfg.MODEL.PRETRAINED)
    return model
