"""Synthetic stand-ins for the datasets BASELINE.json names (no Replica / Redwood data and no network here).

SURVEY.md section 8(d) [D] defines them: an analytic primitive scene (room box interior + sphere + box) ray-cast
through a 640x480 pinhole camera moving on a circle.  Pure numpy, float64 maths, float32 / uint8 outputs, no RNG
unless a noise sigma is requested -- so the oracle and the HIP path always see byte-identical inputs.
"""
import numpy as np

# Camera(fu, fv, cu, cv, width, height)  -- conversions/image_conversions.cpp:27-32 argument order
REPLICA_LIKE_CAM = (320.0, 320.0, 319.5, 239.5, 640, 480)


class Scene:
    def __init__(self, room_min=(-3.0, -2.5, 0.0), room_max=(3.0, 2.5, 3.0),
                 sphere_c=(1.5, 1.0, 0.6), sphere_r=0.5,
                 box_min=(-2.0, -1.5, 0.0), box_max=(-1.2, -0.7, 0.9)):
        self.room_min = np.asarray(room_min, float); self.room_max = np.asarray(room_max, float)
        self.sphere_c = np.asarray(sphere_c, float); self.sphere_r = float(sphere_r)
        self.box_min = np.asarray(box_min, float); self.box_max = np.asarray(box_max, float)

    def raycast(self, o, d):
        """o: (3,), d: (...,3) directions (not normalised). Returns ray parameter t (inf where nothing is hit)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            # room interior: exit distance
            tpos = (np.where(d > 0, self.room_max, self.room_min) - o) / d
            tpos = np.where(d == 0, np.inf, tpos)
            t_room = tpos.min(axis=-1)
            # sphere
            oc = o - self.sphere_c
            a = (d * d).sum(-1); b = 2.0 * (d * oc).sum(-1); c = (oc * oc).sum() - self.sphere_r ** 2
            disc = b * b - 4 * a * c
            sq = np.sqrt(np.maximum(disc, 0.0))
            t_s = (-b - sq) / (2 * a)
            t_s = np.where((disc > 0) & (t_s > 0), t_s, np.inf)
            # box (slab)
            t1 = (self.box_min - o) / d; t2 = (self.box_max - o) / d
            tn = np.minimum(t1, t2); tf = np.maximum(t1, t2)
            tn = np.where(d == 0, -np.inf, tn); tf = np.where(d == 0, np.inf, tf)
            inside0 = (o >= self.box_min) & (o <= self.box_max)
            ok0 = np.where(d == 0, inside0, True).all(axis=-1)
            tnear = tn.max(axis=-1); tfar = tf.min(axis=-1)
            t_b = np.where(ok0 & (tnear <= tfar) & (tnear > 0), tnear, np.inf)
        return np.minimum(np.minimum(t_room, t_s), t_b)


def look_pose(position, yaw, pitch):
    """Row-major 4x4 T_L_C. Camera frame: z forward, x right, y down. yaw about +z, pitch>0 looks up."""
    f = np.array([np.cos(yaw) * np.cos(pitch), np.sin(yaw) * np.cos(pitch), np.sin(pitch)])
    up = np.array([0.0, 0.0, 1.0])
    r = np.cross(f, up); r /= np.linalg.norm(r)
    dn = np.cross(f, r)
    T = np.eye(4)
    T[:3, 0] = r; T[:3, 1] = dn; T[:3, 2] = f; T[:3, 3] = position
    return T.astype(np.float32)


def trajectory_pose(i, n_frames=200, radius=1.0, height=1.5, pitch_deg=-10.0, yaw_offset_deg=0.0, center=(0.0, 0.0)):
    th = 2.0 * np.pi * (i / float(n_frames)) + np.deg2rad(yaw_offset_deg)
    pos = np.array([center[0] + radius * np.cos(th), center[1] + radius * np.sin(th), height])
    return look_pose(pos, th, np.deg2rad(pitch_deg))


def pixel_rays(cam):
    fu, fv, cu, cv, w, h = cam
    w = int(w); h = int(h)
    u = (np.arange(w) + 0.5 - cu) / fu
    v = (np.arange(h) + 0.5 - cv) / fv
    rays = np.empty((h, w, 3)); rays[..., 0] = u[None, :]; rays[..., 1] = v[:, None]; rays[..., 2] = 1.0
    return rays


def render(scene, T_L_C, cam=REPLICA_LIKE_CAM, color=True, noise_sigma=0.0, rng=None, max_range=None):
    """Return (depth float32 [h,w] metres along camera z, rgb uint8 [h,w,3] or None)."""
    T = np.asarray(T_L_C, np.float64).reshape(4, 4)
    rays_c = pixel_rays(cam)
    d = rays_c @ T[:3, :3].T
    o = T[:3, 3]
    t = scene.raycast(o, d)
    depth = np.where(np.isfinite(t), t, 0.0)
    if max_range is not None:
        depth = np.where(depth > max_range, 0.0, depth)
    rgb = None
    if color:
        p = o + d * np.where(np.isfinite(t), t, 0.0)[..., None]
        q = np.floor(8.0 * p + 0.37).astype(np.int64) & 1
        rgb = np.where(q == 1, 192, 64).astype(np.uint8)
    if noise_sigma > 0.0:
        rng = rng or np.random.default_rng(0)
        depth = np.where(depth > 0, depth + rng.normal(0.0, noise_sigma, depth.shape), 0.0)
    return depth.astype(np.float32), rgb


def sequence(n, scene=None, cam=REPLICA_LIKE_CAM, n_frames_in_loop=200, yaw_offset_deg=0.0, color=True, start=0, **kw):
    """Yield (depth, rgb, T_L_C) for frames start..start+n-1 of the SURVEY 8(d) circle trajectory."""
    scene = scene or Scene()
    for i in range(start, start + n):
        T = trajectory_pose(i, n_frames_in_loop, yaw_offset_deg=yaw_offset_deg, **kw)
        depth, rgb = render(scene, T, cam, color=color)
        yield depth, rgb, T


def redwood_like_scene(frame, fps=30.0):
    """BASELINE.json configs[2] stand-in: room 8 x 6 x 2.8 m with one box translating at 0.5 m/s along +x
    (a dynamic object), frame index -> Scene."""
    x0 = -3.0 + 0.5 * (frame / fps)
    return Scene(room_min=(-4.0, -3.0, 0.0), room_max=(4.0, 3.0, 2.8), sphere_c=(2.0, -1.5, 0.5), sphere_r=0.5,
                 box_min=(x0, 1.0, 0.0), box_max=(x0 + 0.6, 1.6, 1.2))


# ------------------------------------------------------------------------------------------------ spinning LiDAR
# BASELINE.json configs[4] / SURVEY.md 8(d) [D]: Lidar(1024, 64, min_range 0.1, vfov 45 deg), ground plane + 40 boxes
# in a 300 x 300 m area (default_rng(1)), ranges beyond 200 m invalid.  Lidar tuple = (azimuth divisions, elevation
# divisions, min_valid_range_m, min_elevation_rad, max_elevation_rad) -- the C-ABI's nvbx_lidar.
SPINNING_LIDAR = (1024, 64, 0.1, -np.deg2rad(22.5), np.deg2rad(22.5))


class LidarScene:
    def __init__(self, n_boxes=40, extent=150.0, seed=1):
        rng = np.random.default_rng(seed)
        c = rng.uniform(-extent, extent, (n_boxes, 2))
        sz = rng.uniform(2.0, 12.0, (n_boxes, 2)); h = rng.uniform(2.0, 15.0, n_boxes)
        keep = np.hypot(c[:, 0], c[:, 1]) > 12.0          # keep the sensor's start area free
        c, sz, h = c[keep], sz[keep], h[keep]
        self.bmin = np.concatenate([c - sz / 2, np.zeros((len(c), 1))], 1)
        self.bmax = np.concatenate([c + sz / 2, h[:, None]], 1)

    def raycast(self, o, d):
        """o (3,), d (N,3) unit directions -> range (inf where nothing is hit)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(d[:, 2] < 0, -o[2] / d[:, 2], np.inf)         # ground plane z = 0
            for bmin, bmax in zip(self.bmin, self.bmax):
                t1 = (bmin - o) / d; t2 = (bmax - o) / d
                tn = np.where(d == 0, -np.inf, np.minimum(t1, t2)).max(axis=1)
                tf = np.where(d == 0, np.inf, np.maximum(t1, t2)).min(axis=1)
                t = np.minimum(t, np.where((tn <= tf) & (tn > 0), tn, np.inf))
        return t


def lidar_beam_dirs(lidar=SPINNING_LIDAR):
    """Unit beam directions [rows, cols, 3] in the sensor frame (x forward, z up); beam (k, j) = pixel centre (j+.5, k+.5)."""
    cols, rows, _, min_el, max_el = lidar
    el = max_el - np.arange(rows) * ((max_el - min_el) / (rows - 1))
    az = -np.pi + np.arange(cols) * (2.0 * np.pi / cols)
    d = np.empty((rows, cols, 3))
    d[..., 0] = np.cos(el)[:, None] * np.cos(az)[None, :]
    d[..., 1] = np.cos(el)[:, None] * np.sin(az)[None, :]
    d[..., 2] = np.sin(el)[:, None]
    return d


def lidar_pose(i, n_frames=200, radius=8.0, height=2.0):
    """Sensor frame = z up, x forward along the tangent of a slow circle (a vehicle driving a loop)."""
    th = 2.0 * np.pi * (i / float(n_frames))
    T = np.eye(4)
    c, s = np.cos(th + np.pi / 2), np.sin(th + np.pi / 2)
    T[:3, :3] = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    T[:3, 3] = [radius * np.cos(th), radius * np.sin(th), height]
    return T.astype(np.float32)


def render_lidar(scene, T_L_C, lidar=SPINNING_LIDAR, max_range=200.0):
    """Range image float32 [rows, cols] (0 = no return within max_range)."""
    T = np.asarray(T_L_C, np.float64).reshape(4, 4)
    dirs = lidar_beam_dirs(lidar)
    d = (dirs.reshape(-1, 3) @ T[:3, :3].T)
    t = scene.raycast(T[:3, 3], d).reshape(dirs.shape[:2])
    rng_img = np.where(np.isfinite(t) & (t <= max_range) & (t >= lidar[2]), t, 0.0)
    return rng_img.astype(np.float32)
