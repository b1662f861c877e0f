"""Synthetic SURREAL-shaped scenes (skeleton pose, bounding cylinder, camera rays, net weights).

No dataset or checkpoint is available offline, so every parity test and every bench line runs on
inputs regenerated from numpy seeds on both the CPU container and the GPU box.  Recipe follows
SURVEY.md §8(d); the reference functions each piece restates are cited per function (paths are
relative to the reference repo).  `tests/golden/gen_golden.py` pins these against the reference.

Pure numpy; no torch, no HIP.
"""
import numpy as np

N_JOINTS = 24

# SMPL kinematic tree (parent of each joint) -- data from core/utils/skeleton_utils.py:98-104
SMPL_PARENTS = np.array([0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21])

# SMPL rest pose joint locations (data table, core/utils/skeleton_utils.py:259-282)
SMPL_REST_POSE = np.array([
    [0.00000000e+00, 2.30003661e-09, -9.86228770e-08],
    [1.63832515e-01, -2.17391014e-01, -2.89178602e-02],
    [-1.57855421e-01, -2.14761734e-01, -2.09642015e-02],
    [-7.04505108e-03, 2.50450850e-01, -4.11837511e-02],
    [2.42021069e-01, -1.08830070e+00, -3.14962119e-02],
    [-2.47206554e-01, -1.10715497e+00, -3.06970738e-02],
    [3.95125849e-03, 5.94849110e-01, -4.03754264e-02],
    [2.12680623e-01, -1.99382353e+00, -1.29327580e-01],
    [-2.10857525e-01, -2.01218796e+00, -1.23002514e-01],
    [9.39484313e-03, 7.19204426e-01, 2.06931755e-02],
    [2.63385147e-01, -2.12222481e+00, 1.46775618e-01],
    [-2.51970559e-01, -2.12153077e+00, 1.60450473e-01],
    [3.83779174e-03, 1.22592449e+00, -9.78838727e-02],
    [1.91201791e-01, 1.00385976e+00, -6.21964522e-02],
    [-1.77145526e-01, 9.96228695e-01, -7.55542740e-02],
    [1.68482102e-02, 1.38698268e+00, 2.44048554e-02],
    [4.01985168e-01, 1.07928419e+00, -7.47655183e-02],
    [-3.98825467e-01, 1.07523870e+00, -9.96334553e-02],
    [1.00236952e+00, 1.05217218e+00, -1.35129794e-01],
    [-9.86728609e-01, 1.04515052e+00, -1.40235111e-01],
    [1.56646240e+00, 1.06961894e+00, -1.37338534e-01],
    [-1.56946480e+00, 1.05935931e+00, -1.53905824e-01],
    [1.75282109e+00, 1.04682994e+00, -1.68231070e-01],
    [-1.75758195e+00, 1.04255080e+00, -1.77773550e-01]], dtype=np.float32)

# SURREAL rest-pose scale: 0.25 / 0.00035 * ext_scale(0.001)   (core/load_surreal.py:18,117,279)
SURREAL_SCALE = 0.25 / 0.00035 * 0.001
EXT_SCALE = 0.001


def rodrigues(rotvec):
    """Axis-angle -> 3x3 rotation (what scipy Rotation.from_rotvec(...).as_matrix() returns)."""
    rotvec = np.asarray(rotvec, dtype=np.float64)
    th = np.linalg.norm(rotvec)
    if th < 1e-12:
        return np.eye(3)
    k = rotvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def forward_kinematics(bones, rest_pose):
    """Local-to-world 4x4 per joint from per-joint axis-angle rotations.

    Restates get_smpl_l2ws (core/utils/skeleton_utils.py:334-378): root = [R0 | rest[0]], child =
    parent_l2w @ [R_j | rest[j] - rest[parent]].
    """
    rest_pose = np.asarray(rest_pose, dtype=np.float32)
    l2ws = np.zeros((N_JOINTS, 4, 4), dtype=np.float64)
    for j in range(N_JOINTS):
        loc = np.eye(4)
        loc[:3, :3] = rodrigues(bones[j])
        if j == 0:
            loc[:3, 3] = rest_pose[0]
            l2ws[0] = loc
        else:
            p = SMPL_PARENTS[j]
            loc[:3, 3] = rest_pose[j] - rest_pose[p]
            l2ws[j] = l2ws[p] @ loc
    return l2ws


def make_pose(seed, sigma=0.2, scale=SURREAL_SCALE):
    """Pose `seed`: bones ~ N(0, sigma^2) -> l2ws -> keypoints, world->bone matrices (skts)."""
    rs = np.random.RandomState(seed)
    bones = rs.normal(0.0, sigma, size=(N_JOINTS, 3)).astype(np.float32)
    l2ws = forward_kinematics(bones, SMPL_REST_POSE * np.float32(scale))
    skts = np.linalg.inv(l2ws)
    return {
        "bones": bones,
        "l2ws": l2ws.astype(np.float32),
        "kp": l2ws[:, :3, 3].astype(np.float32),
        "skts": skts.astype(np.float32),
    }


def bounding_cylinder(kp, ext_scale=EXT_SCALE, extend_mm=250.0, top_expand_ratio=1.60, bot_expand_ratio=1.10):
    """(cx, cz, radius, top, bot) of the x-z bounding cylinder, head direction '-y'.

    Restates get_kp_bounding_cylinder (core/utils/skeleton_utils.py:542-591) with the render-time
    ratios used by kp_to_valid_rays (core/utils/ray_utils.py:88-103).
    """
    kp = np.asarray(kp)
    root = kp[0]
    dist = np.linalg.norm(kp[:, [0, 2]] - root[[0, 2]], axis=-1)
    flip = -1.0
    max_h = (flip * kp[:, 1]).max()
    min_h = (flip * kp[:, 1]).min()
    ext = extend_mm * ext_scale
    return np.array([root[0], root[2], dist.max() + ext,
                     flip * (max_h + ext * top_expand_ratio),
                     flip * (min_h - ext * bot_expand_ratio)], dtype=np.float32)


def default_c2w(tz=3.0):
    c2w = np.eye(4, dtype=np.float32)
    c2w[2, 3] = tz
    return c2w


def camera_rays(H, W, focal, c2w):
    """Per-pixel rays (o, d) [H,W,3]; d is NOT normalised (core/utils/ray_utils.py:6-28)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    dirs = np.stack([(i - W * 0.5) / focal, -(j - H * 0.5) / focal, -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., None, :] * c2w[:3, :3], -1).astype(np.float32)
    rays_o = np.broadcast_to(c2w[:3, 3], rays_d.shape).astype(np.float32)
    return rays_o, rays_d


def cylinder_bbox(cyl, H, W, focal, c2w, center=None):
    """2-D pixel bbox (tl, br) of the projected cylinder caps.

    Restates cylinder_to_box_2d (core/utils/skeleton_utils.py:607-690) + nerf_c2w_to_extrinsic
    (:442): 50 points per cap, OpenCV-style extrinsic = inv(c2w with y,z columns negated).
    focal: scalar or (fx, fy) (focal_to_intrinsic_np); center: principal point (x, y) or None = image centre.
    """
    rads = np.linspace(0.0, 2 * np.pi, 50)
    x = cyl[0] + np.cos(rads) * cyl[2]
    z = cyl[1] + np.sin(rads) * cyl[2]
    pts = np.concatenate([np.stack([x, np.full_like(x, cyl[3]), z, np.ones_like(x)], -1),
                          np.stack([x, np.full_like(x, cyl[4]), z, np.ones_like(x)], -1)], 0)
    sw = np.concatenate([c2w[:, 0:1], -c2w[:, 1:2], -c2w[:, 2:3], c2w[:, 3:]], -1)
    w2c = np.linalg.inv(sw)
    cam = pts @ w2c.T
    fx, fy = (focal, focal) if np.ndim(focal) == 0 else (focal[0], focal[1])
    K = np.array([[fx, 0, 0], [0, fy, 0], [0, 0, 1]], dtype=np.float64)
    proj = cam[:, :3] @ K.T
    p2 = proj[:, :2] / proj[:, 2:3]
    off = np.array([int(W * .5), int(H * .5)]) if center is None else np.array([int(center[0]), int(center[1])])
    tl = np.floor(p2.min(0)).astype(np.int32) + off
    br = np.ceil(p2.max(0)).astype(np.int32) + off
    tl = np.array([np.clip(tl[0], 0, W - 1), np.clip(tl[1], 0, H - 1)])
    br = np.array([np.clip(br[0], 0, W - 1), np.clip(br[1], 0, H - 1)])
    return tl, br


def frame_rays(H, W, focal, cyl, c2w=None):
    """Rays of one frame restricted to the cylinder's 2-D bbox (kp_to_valid_rays, ray_utils.py:83-136).

    Returns rays_o [Nv,3], rays_d [Nv,3], valid_idx [Nv] (flat pixel index h*W+w).
    """
    c2w = default_c2w() if c2w is None else c2w
    ro, rd = camera_rays(H, W, focal, c2w)
    tl, br = cylinder_bbox(cyl, H, W, focal, c2w)
    hh, ww = np.meshgrid(np.arange(tl[1], br[1]), np.arange(tl[0], br[0]), indexing="ij")
    idx = (hh * W + ww).reshape(-1)
    return ro.reshape(-1, 3)[idx].copy(), rd.reshape(-1, 3)[idx].copy(), idx


# ---------------------------------------------------------------------------------------------
# network parameters (names and shapes = the reference's state_dict)
# ---------------------------------------------------------------------------------------------
def net_shapes(multires=7, multires_views=4, framecode_ch=0, W=256, D=8, skip=4, n_joints=N_JOINTS):
    in_v = n_joints * (1 + 2 * multires)
    in_b = n_joints * 3
    in_d = n_joints * 3 * (1 + 2 * multires_views)
    dnet = in_v + in_b
    shapes = {}
    for i in range(D):
        fan_in = dnet if i == 0 else (W + dnet if i == skip + 1 else W)
        shapes[f"pts_linears.{i}"] = (W, fan_in)
    shapes["alpha_linear"] = (1, W)
    shapes["feature_linear"] = (W, W)
    shapes["views_linears.0"] = (W // 2, W + in_d + framecode_ch)
    shapes["rgb_linear"] = (3, W // 2)
    return shapes


def make_net_params(seed, multires=7, multires_views=4, framecode_ch=0, n_codes=0, alpha_bias=1.0):
    """torch.nn.Linear-style init U(+-1/sqrt(fan_in)) from a numpy Generator; alpha bias forced to +1
    because the default init renders exactly zero density."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, (o, i) in net_shapes(multires, multires_views, framecode_ch).items():
        bound = 1.0 / np.sqrt(i)
        out[name + ".weight"] = rng.uniform(-bound, bound, size=(o, i)).astype(np.float32)
        out[name + ".bias"] = rng.uniform(-bound, bound, size=(o,)).astype(np.float32)
    if alpha_bias is not None:
        out["alpha_linear.bias"][:] = alpha_bias
    if framecode_ch > 0:
        std = np.sqrt(2.0 / (n_codes + framecode_ch))
        out["framecodes.codes.weight"] = (rng.standard_normal((n_codes, framecode_ch)) * std).astype(np.float32)
    return out


def make_scene(pose_seed=0, H=512, W=512, focal=600.0):
    """One synthetic frame: pose, cylinder and bbox-restricted rays (BASELINE config 2 geometry)."""
    pose = make_pose(pose_seed)
    cyl = bounding_cylinder(pose["kp"])
    ro, rd, idx = frame_rays(H, W, focal, cyl)
    return {"pose": pose, "cyl": cyl, "rays_o": ro, "rays_d": rd, "valid_idx": idx, "H": H, "W": W, "focal": focal}


def scene_batch(n_rays, pose_seeds, H=64, W=64, focal=75.0, ray_seed=0, per_ray_pose=False):
    """Seeded ray batch in the layout of the reference's collate (core/dataset.py:92-103, 813-820):
    n_rays rays picked from the bbox of pose_seeds[0]'s frame; per_ray_pose cycles the poses over rays.
    Returns rays_o, rays_d [n,3], kp [n,24,3], skts [n,24,4,4], bones [n,24,3], cyls [n,5], which [n]."""
    scenes = [make_scene(s, H, W, focal) for s in pose_seeds]
    sc = scenes[0]
    rs = np.random.RandomState(ray_seed)
    pick = rs.choice(len(sc["rays_o"]), size=n_rays, replace=len(sc["rays_o"]) < n_rays)
    ro, rd = sc["rays_o"][pick], sc["rays_d"][pick]
    which = np.arange(n_rays) % len(scenes) if per_ray_pose else np.zeros(n_rays, dtype=np.int64)
    kp = np.stack([scenes[w]["pose"]["kp"] for w in which])
    skts = np.stack([scenes[w]["pose"]["skts"] for w in which])
    bones = np.stack([scenes[w]["pose"]["bones"] for w in which])
    cyls = np.stack([scenes[w]["cyl"] for w in which])
    return ro, rd, kp, skts, bones, cyls, which
