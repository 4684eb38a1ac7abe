"""Stand-in for the reference's yacs config (config/default.py:17-149; yacs is not a
dependency here).  Only the MODEL keys the HRNet constructor reads are reproduced
(config/default.py:35-74); they can be read both as items and as attributes, which is what
models/seg_hrnet.py:261,273,325 does."""
from __future__ import annotations

import copy


class CfgNode(dict):
    """dict with attribute access (the subset of yacs.CfgNode the model code relies on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)


def _stage(num_branches, num_blocks, num_channels, num_modules=1):
    return CfgNode(NUM_MODULES=num_modules, NUM_BRANCHES=num_branches, NUM_BLOCKS=list(num_blocks),
                   NUM_CHANNELS=list(num_channels), BLOCK="BASIC", FUSE_METHOD="SUM")


def make_config(widths=(32, 64, 128, 256), blocks=((2,), (2, 2), (2, 2, 2), (4, 4, 4, 4)),
                modules=(1, 1, 1, 1), pretrained: str = "") -> CfgNode:
    """Default = config/default.py:39-74 (the reference's 'W32': widths 32/64/128/256)."""
    hr = CfgNode(PRETRAINED_LAYERS=["*"], STEM_INPLANES=64, FINAL_CONV_KERNEL=1, WITH_HEAD=True)
    for i in range(4):
        nb = len(blocks[i])
        hr[f"STAGE{i + 1}"] = _stage(nb, blocks[i], widths[:nb], modules[i])
    return CfgNode(MODEL=CfgNode(NAME="seg_hrnet", PRETRAINED=pretrained,
                                 EXTRA=CfgNode(HIGH_RESOLUTION_NET=hr)))


config = make_config()   # `from config import config` idiom of val.py:17
