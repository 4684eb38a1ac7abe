"""CPU: the host-side pose solve (esa-pose-estimation_amd/pnp.py, SURVEY.md §8f NEXT-1).
PARITY UNPINNED — cv2 / cpnp are absent and the reference holds no expected pose, so the published
algorithms are validated on synthetic projections with known (q, t), scored with the reference's own
SPEED metric (demo.py:297, 308).  Camera: the ESA intrinsics of lib/utils/base_utils.py:250-252."""
import numpy as np
import pytest

from esa_pose_estimation_amd import pnp as P
from esa_pose_estimation_amd import synth

K_ESA = np.array([[3003.41297, 0.0, 960.0], [0.0, 3003.41297, 600.0], [0.0, 0.0, 1.0]])


def _scene(seed, n_pts=30):
    pts = synth.uniform(f"pts{seed}", seed, (n_pts, 3), -0.6, 0.6).astype(np.float64)     # ~1.2 m satellite
    q = synth.normal(f"q{seed}", seed, (4,)).astype(np.float64)
    q /= np.linalg.norm(q)
    t = np.array([*synth.uniform(f"txy{seed}", seed, (2,), -0.4, 0.4), *synth.uniform(f"tz{seed}", seed, (1,), 4.0, 20.0)],
                 np.float64)
    R = P.quat_wxyz_to_rotation(q)
    return pts, q, t, R, P.project(pts, R, t, K_ESA)


def test_rotation_conversions_roundtrip():
    for seed in range(20):
        r = synth.normal(f"r{seed}", seed, (3,)).astype(np.float64)
        R = P.rodrigues(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        assert np.allclose(P.rodrigues(P.rodrigues_inv(R)), R, atol=1e-10)
        q = P.rotation_to_quat_wxyz(R)
        assert np.allclose(P.quat_wxyz_to_rotation(q), R, atol=1e-12)
    assert np.allclose(P.rodrigues(np.zeros(3)), np.eye(3))
    Rpi = P.rodrigues(np.array([np.pi, 0, 0]))
    assert np.allclose(P.rodrigues(P.rodrigues_inv(Rpi)), Rpi, atol=1e-6)


@pytest.mark.parametrize("seed", range(8))
def test_epnp_exact_on_noise_free_projections(seed):
    pts, q, t, R, uv = _scene(seed)
    Re, te = P.epnp(pts, uv, K_ESA)
    score, st, sr = P.speed_score(P.rotation_to_quat_wxyz(Re), te, q, t)
    assert score < 1e-5, (score, st, sr)
    Rt = P.pnp(pts, uv, K_ESA, P.SOLVEPNP_EPNP)               # pnp.py:46-90 contract: 3x4 [R|t]
    assert Rt.shape == (3, 4) and np.allclose(Rt[:, :3], R, atol=1e-5)


@pytest.mark.parametrize("seed", range(6))
def test_ransac_rejects_outliers_and_refinement_uses_peak_weights(seed):
    pts, q, t, R, uv = _scene(100 + seed)
    noisy = uv + 0.5 * synth.normal(f"noise{seed}", seed, uv.shape).astype(np.float64)   # 0.5 px keypoint noise
    noisy[[3, 11, 17]] += np.array([[60.0, -45.0], [-80.0, 30.0], [25.0, 90.0]])       # three gross outliers
    peaks = np.full(len(pts), 0.9)
    peaks[[3, 11, 17]] = 0.05                                                            # ... with low peaks
    Rr, tr, mask = P.solve_pnp_ransac(pts, noisy, K_ESA)
    assert not mask[[3, 11, 17]].any() and mask.sum() >= 24
    cam0 = np.concatenate([P.rodrigues_inv(Rr), tr])
    cam = P.cpnp_m(pts, noisy, peaks, K_ESA, cam0)
    s0 = P.speed_score(P.rotation_to_quat_wxyz(Rr), tr, q, t)[0]
    s1 = P.speed_score(P.rotation_to_quat_wxyz(P.rodrigues(cam[:3])), cam[3:], q, t)[0]
    assert s1 < 0.01, (s0, s1)              # reference's best synthetic SPEED score is 0.0193 (README.md:11)
    # the weighted refinement never makes the weighted reprojection cost worse than its start
    def cost(c):
        return np.sum((peaks[:, None] * (P.project(pts, P.rodrigues(c[:3]), c[3:], K_ESA) - noisy)) ** 2)
    assert cost(cam) <= cost(cam0) + 1e-9


def test_keypoints_to_pose_end_to_end():
    """val.py:172-224 on a row of the GPU path's output: top-k by peak, crop -> image, EPnP, refine."""
    pts, q, t, R, uv = _scene(7, n_pts=30)
    x0, y0, rate = 400.0, 200.0, 0.25                         # crop origin and scale (data_load_val.py:195)
    kp = np.concatenate([(uv - [x0, y0]) * rate, np.full((30, 1), 0.95)], 1)
    kp[5, 2] = 0.1                                            # one weak keypoint: dropped by the 0.8 rule ...
    kp[5, :2] += 40                                           # ... and wrong
    qe, te, Re = P.keypoints_to_pose(kp, pts, K_ESA, (x0, y0), rate, thresh=0.8, min_k=24)
    assert P.speed_score(qe, te, q, t)[0] < 1e-4


def test_speed_score_definition():
    q = np.array([1.0, 0, 0, 0])
    s, st, sr = P.speed_score(q, [0, 0, 10.0], -q, [0, 0, 8.0])     # q and -q are the same rotation
    assert abs(st - 0.25) < 1e-12 and sr < 1e-7
