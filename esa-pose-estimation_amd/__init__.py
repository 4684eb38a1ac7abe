"""MI355X-native HRNet keypoint-heatmap inference path for the ESA/Kelvins SPEED pipeline.

Drop-in surface (mirrors the reference's flat modules):
    seg_hrnet / seg_hrnet2 / seg_hrnet3   get_seg_model(cfg) -> nn.Module   (models/seg_hrnet*.py)
    inference                heatmaps_to_keypoints, get_max_preds, get_final   (inference.py)
    config                   `config` CfgNode with the HRNet stage table        (config/default.py)
    parallel                 crop sharding + RCCL keypoint all-gather
    pnp                      host pose solve after the path: EPnP + RANSAC, peak-weighted LM (pnp.py, cpnp)
    synth                    seed-reproducible weights / crops for tests and bench
"""
__all__ = ["seg_hrnet", "seg_hrnet2", "seg_hrnet3", "inference", "config", "parallel", "pnp", "synth", "hrnet", "build"]
