"""CPU: pin the oracle (oracle/) against the fixtures produced by the REAL reference
(tests/golden/make_golden.py).  These are the 'is the checker itself right' tests."""
import os

import numpy as np
import pytest
import torch

import esa_pose_estimation_amd.synth as synth
from oracle import hrnet_ref, keypoints_ref

FULL = ["tiny_hrnet2_64", "tiny_hrnet_64", "w32_hrnet2_128", "w32_hrnet2_256", "w32_hrnet_256",
        "small_hrnet3_64", "w32_hrnet3_128",
        "w32_hrnet2_128_g1", "w32_hrnet2_256_g1"]      # weight gain 1.0: max|act| 66 / 79, SURVEY.md §8d


def _load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, tag + ".npz"), allow_pickle=False)


def cfg_from_fixture(g):
    flat = [int(v) for v in g["blocks_flat"]]
    blocks = (tuple(flat[0:1]), tuple(flat[1:3]), tuple(flat[3:6]), tuple(flat[6:10]))
    cin, k, variant = {"seg_hrnet": (3, 32, 0), "seg_hrnet2": (1, 11, 0), "seg_hrnet3": (1, 30, 1)}[str(g["variant"])]
    return hrnet_ref.default_cfg(cin=cin, num_keypoints=k, widths=tuple(int(v) for v in g["widths"]),
                                 blocks=blocks, variant=variant)


def state_from_fixture(g):
    shapes = {str(k): tuple(int(x) for x in s.split(",")) if s else ()
              for k, s in zip(g["state_keys"], g["state_shapes"])}
    return synth.make_state_dict(shapes, seed=int(g["seed"]), gain=float(g["gain"]) if "gain" in g.files else 0.5)


@pytest.mark.parametrize("tag", FULL)
def test_full_net_matches_reference(golden_dir, tag):
    g = _load(golden_dir, tag)
    cfg = cfg_from_fixture(g)
    sd = state_from_fixture(g)
    x = synth.make_crops(int(g["n"]), cfg["cin"], int(g["hw"]), int(g["hw"]), seed=int(g["seed"]))
    with torch.no_grad():
        y = hrnet_ref.forward(sd, cfg, x).numpy()
    s = int(g["subsample"])
    err = np.abs(y[:, :, ::s, ::s] - g["out"]).max()
    scale = max(1.0, float(g["out_absmax"]))     # (the gain-1.0 fixtures reach |out| ~ 20)
    assert err <= 1e-5 * scale, err              # same torch ops, fp32: expect ~1e-6 at most
    assert abs(float(np.abs(y).max()) - float(g["out_absmax"])) <= 1e-5 * scale
    flat = y.reshape(y.shape[0], y.shape[1], -1)
    assert np.array_equal(flat.argmax(-1), g["plane_argmax"])


@pytest.mark.parametrize("tag", FULL)
def test_enumeration_matches_reference_state_dict(golden_dir, tag):
    """Every Conv2d/BatchNorm2d the reference registers is in the oracle's own enumeration."""
    g = _load(golden_dir, tag)
    cfg = cfg_from_fixture(g)
    keys = {str(k): s for k, s in zip(g["state_keys"], g["state_shapes"])}
    convs = hrnet_ref.enumerate_convs(cfg)
    want = set()
    for c in convs:
        want.add(c["name"] + ".weight")
        assert keys[c["name"] + ".weight"] == f"{c['cout']},{c['cin']},{c['k']},{c['k']}"
        if c["bias"]:
            want.add(c["name"] + ".bias")
        if c["bn"]:
            for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
                want.add(f"{c['bn']}.{s}")
    if cfg["variant"] == 1:      # CBAM parameters (seg_hrnet3.py:32-61): per BasicBlock and on the net itself
        owners = {""} | {c["name"][:-len(".conv1")] + "." for c in convs if c["name"].endswith(".conv1") and "." in c["name"]}
        for o in owners:
            want |= {o + "ca.fc.0.weight", o + "ca.fc.2.weight", o + "sa.conv1.weight"}
    assert want == set(keys)


def test_mac_count_matches_survey():
    """seg_hrnet2 W32 256^2 = 28.52 GFLOP, seg_hrnet = 30.16 GFLOP."""
    assert abs(hrnet_ref.conv_flops(hrnet_ref.default_cfg(1, 11), 256, 256) / 1e9 - 28.52) < 0.01
    assert abs(hrnet_ref.conv_flops(hrnet_ref.default_cfg(3, 32), 256, 256) / 1e9 - 30.16) < 0.01


@pytest.mark.parametrize("tag", ["hrmodule2", "hrmodule3", "hrmodule4"])
def test_hr_module_matches_reference(golden_dir, tag):
    g = _load(golden_dir, tag)
    nb, w, nblocks, hw, seed = int(g["nb"]), [int(v) for v in g["widths"]], \
        [int(v) for v in g["nblocks"]], int(g["hw"]), int(g["seed"])
    # rebuild the module's state_dict names/shapes from the oracle's enumeration
    cfg = hrnet_ref.default_cfg(widths=tuple(w) + (0,) * (4 - nb))
    shapes = {}
    for b in range(nb):
        for k in range(nblocks[b]):
            for c in ("conv1", "conv2"):
                shapes[f"branches.{b}.{k}.{c}.weight"] = (w[b], w[b], 3, 3)
            for bn in ("bn1", "bn2"):
                for s, shp in (("weight", (w[b],)), ("bias", (w[b],)), ("running_mean", (w[b],)),
                               ("running_var", (w[b],)), ("num_batches_tracked", ())):
                    shapes[f"branches.{b}.{k}.{bn}.{s}"] = shp

    def bn_shapes(p, c):
        for s, shp in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)),
                       ("running_var", (c,)), ("num_batches_tracked", ())):
            shapes[f"{p}.{s}"] = shp
    for i in range(nb):
        for j in range(nb):
            if j > i:
                shapes[f"fuse_layers.{i}.{j}.0.weight"] = (w[i], w[j], 1, 1)
                bn_shapes(f"fuse_layers.{i}.{j}.1", w[i])
            elif j < i:
                for k in range(i - j):
                    co = w[i] if k == i - j - 1 else w[j]
                    shapes[f"fuse_layers.{i}.{j}.{k}.0.weight"] = (co, w[j], 3, 3)
                    bn_shapes(f"fuse_layers.{i}.{j}.{k}.1", co)
    sd = synth.make_state_dict(shapes, seed=seed)
    sd = {"m." + k: v for k, v in sd.items()}
    xs = [torch.from_numpy(synth.normal(f"{tag}.x{b}", seed, (2, w[b], hw >> b, hw >> b)))
          for b in range(nb)]
    # NB: synth hashes by *name*; the fixture was generated with un-prefixed names
    sd = {"m." + k[2:]: v for k, v in sd.items()}
    with torch.no_grad():
        ys = hrnet_ref._hr_module(sd, "m", xs, nblocks)
    for b in range(nb):
        assert np.abs(ys[b].numpy() - g[f"y{b}"]).max() <= 1e-5


@pytest.mark.parametrize("tag", ["keypoints_gauss", "keypoints_adversarial", "keypoints_randn"])
def test_keypoints_match_reference(golden_dir, tag):
    g = _load(golden_dir, tag)
    hm = g["hm"]
    coords, maxvals = keypoints_ref.argmax_keypoints(hm)
    assert np.array_equal(coords, g["coords"])            # inference.get_max_preds
    assert np.array_equal(coords, g["caller_xy"])         # two-stage torch.max of demo.py
    assert np.array_equal(maxvals, g["maxvals"])
    assert np.array_equal(maxvals, g["caller_max"])
    refined = keypoints_ref.refine_keypoints(hm, coords)
    assert np.array_equal(refined, g["refined"])          # inference.get_final, bit-exact
    kp = keypoints_ref.heatmaps_to_keypoints(hm)
    assert kp.shape == hm.shape[:2] + (3,)


def test_adversarial_cases_do_what_the_survey_says(golden_dir):
    g = _load(golden_dir, "keypoints_adversarial")
    xy, ref = g["caller_xy"][0], g["refined"][0]
    assert tuple(xy[0]) == (1.0, 3.0)                     # 4-way tie -> first in row-major order
    assert tuple(xy[1]) == (0.0, 0.0)                     # all-equal plane
    for k in (2, 3, 4, 6, 10):                            # border / clamp / hxx==0: unrefined
        assert np.array_equal(xy[k], ref[k]), k
    assert ref[8][0] < xy[8][0] - 1.0                     # large NEGATIVE offset IS applied
    assert np.array_equal(xy[9], ref[9])                  # large positive offset rejected


def test_topk_and_backprojection():
    mv = np.array([0.9, 0.1, 0.85, 0.5, 0.95], np.float32)
    assert keypoints_ref.select_topk(mv, 0.8, 0) == [4, 0, 2]
    assert keypoints_ref.select_topk(mv, 0.8, 4) == [4, 0, 2, 3]
    p = keypoints_ref.crop_to_image(np.array([[10.0, 20.0]]), 0.5, 100.0, 200.0)
    assert np.allclose(p, [[120.0, 240.0]])
