"""Deterministic KITTI-shaped synthetic worlds.

No dataset ships with the reference (no .pcd/.bag), so every config is exercised on a
synthetic street scene: analytic ground height field, axis-aligned building / parked-car /
moving-car / pedestrian boxes, pole cylinders, an HDL-64-like (or Ouster-128-like) ray-cast
sensor, SuMa-like poses, and a naively accumulated map in mapgen's semantics
(src/mapgen/mapgen.hpp:211-239: drop r<2.7 m, z += 1.73, pose transform, 0.2 m voxel).

Labels follow SemanticKITTI as the reference decodes them (erasor_utils.cpp:3,64-65):
intensity = float(sem | inst << 16), dynamic classes 252..259.

Pure numpy; used by tests/ and bench.py.  Nothing here is on the product's compute path.
"""
import numpy as np

LABEL_ROAD, LABEL_SIDEWALK, LABEL_BUILDING, LABEL_POLE, LABEL_PARKED = 40, 48, 50, 80, 10
LABEL_MOVING_CAR, LABEL_MOVING_PERSON = 252, 254
LIDAR_HEIGHT = 1.73  # config/seq_05.yaml:32
CAR_BODY_SIZE = 2.7  # mapgen.hpp:8


def yaw_pose(x, y, z, yaw):
    """7-vector pose x y z qx qy qz qw (geometry_msgs::Pose order used by geoPose2eigen)."""
    return np.array([x, y, z, 0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)], np.float64)


def pose_to_matrix(pose7):
    """float64 4x4 of a pose (tf formula); erasor_amd.geopose2eigen gives the float32 one the step consumes."""
    x, y, z, w = pose7[3:]
    s = 2.0 / (x * x + y * y + z * z + w * w)
    R = np.array([
        [1 - (y * y + z * z) * s, (x * y - w * z) * s, (x * z + w * y) * s],
        [(x * y + w * z) * s, 1 - (x * x + z * z) * s, (y * z - w * x) * s],
        [(x * z - w * y) * s, (y * z + w * x) * s, 1 - (x * x + y * y) * s]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = pose7[:3]
    return T


class Lidar:
    def __init__(self, beams=64, az_steps=2000, el_top_deg=2.0, el_bot_deg=-24.8, max_range=120.0, min_range=3.0,
                 noise=0.02):
        self.beams, self.az_steps = beams, az_steps
        el = np.deg2rad(np.linspace(el_top_deg, el_bot_deg, beams))
        az = np.linspace(-np.pi, np.pi, az_steps, endpoint=False)
        ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
        d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (beams, az_steps))], -1)
        self.dirs = d.reshape(-1, 3)
        self.max_range, self.min_range, self.noise = max_range, min_range, noise

    @staticmethod
    def hdl64(az_steps=2000):
        return Lidar(64, az_steps)

    @staticmethod
    def ouster128(az_steps=2048):
        # config/your_own_env_ouster.yaml shape: 128 beams, +-22.5 deg
        return Lidar(128, az_steps, 22.5, -22.5, max_range=100.0, min_range=1.0)


class World:
    """Parallel streets along +x, `n_streets` of them `street_gap` apart; the vehicle drives street 0."""

    def __init__(self, seed=20210305 + 5, length=400.0, n_streets=1, street_gap=50.0, n_moving=6, n_peds=4,
                 lane_speed=(5.0, 12.0)):
        rng = np.random.default_rng(seed)
        self.seed, self.length, self.n_streets, self.street_gap = seed, float(length), n_streets, float(street_gap)
        boxes, labels = [], []  # static AABBs: xmin ymin zmin xmax ymax zmax (z relative to local ground)
        poles = []
        self.street_y = np.array([k * street_gap * (1 if k % 2 == 0 else -1) * 0.5 * (2 if k else 0) for k in range(n_streets)])
        # street_y: 0, -gap, +gap*2/2 ... keep simple & symmetric:
        self.street_y = np.array([((k + 1) // 2) * street_gap * (1 if k % 2 else -1) for k in range(n_streets)], np.float64)
        for yc in self.street_y:
            for side in (-1.0, 1.0):
                x = -20.0 + rng.uniform(0, 5)
                while x < length + 20.0:
                    blen = rng.uniform(10.0, 25.0)
                    depth = rng.uniform(8.0, 12.0)
                    h = rng.uniform(6.0, 15.0)
                    y0 = yc + side * rng.uniform(9.0, 12.0)
                    y1 = y0 + side * depth
                    boxes.append([x, min(y0, y1), -0.5, x + blen, max(y0, y1), h])
                    labels.append(LABEL_BUILDING)
                    x += blen + rng.uniform(3.0, 8.0)
                # poles / trunks every ~15 m on the pavement
                xs = np.arange(0.0, length, 15.0) + rng.uniform(-3, 3, size=len(np.arange(0.0, length, 15.0)))
                for px in xs:
                    poles.append([px, yc + side * rng.uniform(6.0, 7.5), rng.uniform(0.15, 0.4), rng.uniform(3.0, 8.0)])
                # parked cars
                for px in np.arange(10.0, length - 10.0, 22.0):
                    if rng.uniform() < 0.6:
                        cx = px + rng.uniform(-4, 4)
                        cy = yc + side * 4.6
                        boxes.append([cx - 2.1, cy - 0.9, 0.0, cx + 2.1, cy + 0.9, 1.5])
                        labels.append(LABEL_PARKED)
        self.boxes = np.array(boxes, np.float64).reshape(-1, 6)
        self.box_labels = np.array(labels, np.int64)
        self.poles = np.array(poles, np.float64).reshape(-1, 4)  # x y r h
        # moving objects on street 0: x(t) = x0 + v t, lane y, size, label
        mov = []
        for k in range(n_moving):
            direction = 1.0 if k % 2 == 0 else -1.0
            v = direction * rng.uniform(*lane_speed)
            x0 = rng.uniform(0.1, 0.9) * length
            mov.append([x0, self.street_y[0] + (-1.75 if direction > 0 else 1.75), v, 4.2, 1.8, 1.5,
                        LABEL_MOVING_CAR | ((k + 1) << 16)])
        for k in range(n_peds):
            direction = 1.0 if k % 2 == 0 else -1.0
            mov.append([rng.uniform(0.1, 0.9) * length, self.street_y[0] + direction * rng.uniform(5.2, 5.8),
                        direction * rng.uniform(1.0, 1.6), 0.5, 0.5, 1.7, LABEL_MOVING_PERSON | ((100 + k) << 16)])
        self.moving = np.array(mov, np.float64).reshape(-1, 7)

    # ---- geometry ---------------------------------------------------------------------------
    @staticmethod
    def ground_h(x, y):
        with np.errstate(invalid="ignore"):
            return 0.3 * np.sin(x / 40.0) + 0.2 * np.sin(y / 25.0)

    def ground_label(self, y):
        d = np.min(np.abs(np.asarray(y)[..., None] - self.street_y[None, :]), axis=-1)
        return np.where(d < 5.0, LABEL_ROAD, LABEL_SIDEWALK)

    def moving_boxes(self, t):
        """AABBs of the moving objects at time t [s] (z relative to local ground)."""
        m = self.moving
        if len(m) == 0:
            return np.zeros((0, 6)), np.zeros(0, np.int64)
        cx = m[:, 0] + m[:, 2] * t
        # wrap around the street so objects stay in the scene
        cx = np.mod(cx, self.length)
        b = np.stack([cx - m[:, 3] / 2, m[:, 1] - m[:, 4] / 2, np.zeros(len(m)), cx + m[:, 3] / 2, m[:, 1] + m[:, 4] / 2,
                      m[:, 5]], 1)
        return b, m[:, 6].astype(np.int64)

    def pose(self, frame, step=1.0, x0=20.0, jitter_rng=None):
        """pose7 of frame (10 Hz, ~`step` m/frame) on street 0, gentle weave, SuMa-like jitter."""
        s = x0 + frame * step
        y = self.street_y[0] - 1.75 + 0.6 * np.sin(s / 60.0)
        yaw = np.arctan(0.6 / 60.0 * np.cos(s / 60.0))
        x = s
        if jitter_rng is not None:
            x += jitter_rng.normal(0, 0.01)
            y += jitter_rng.normal(0, 0.01)
            yaw += jitter_rng.normal(0, 0.0005)
        return yaw_pose(x, y, float(self.ground_h(x, y)), yaw)

    # ---- ray casting ------------------------------------------------------------------------
    def cast(self, pose7, lidar, frame, cull=130.0):
        """One scan in the LIDAR frame: (n,4) float32 xyzi.  t = frame * 0.1 s for moving objects."""
        rng = np.random.default_rng(self.seed * 1000 + frame)
        Tb = pose_to_matrix(pose7)
        o = Tb[:3, 3] + Tb[:3, :3] @ np.array([0, 0, LIDAR_HEIGHT])
        D = lidar.dirs @ Tb[:3, :3].T  # world directions
        n = len(D)
        best_t = np.full(n, np.inf)
        best_l = np.zeros(n, np.int64)
        # ground (fixed-point on the height field), downward rays only
        dz = D[:, 2]
        down = dz < -1e-3
        t = np.where(down, (self.ground_h(o[0], o[1]) - o[2]) / np.where(down, dz, -1.0), np.inf)
        for _ in range(4):
            px, py = o[0] + t * D[:, 0], o[1] + t * D[:, 1]
            t = np.where(down, (self.ground_h(px, py) - o[2]) / np.where(down, dz, -1.0), np.inf)
        ok = down & (t > 0)
        best_t = np.where(ok, t, best_t)
        best_l = np.where(ok, self.ground_label(o[1] + np.where(ok, t, 0) * D[:, 1]), best_l)
        # boxes (static + moving) and poles: only the azimuth columns a primitive can occupy are tested
        nb_, na_ = lidar.beams, lidar.az_steps
        yaw = np.arctan2(Tb[1, 0], Tb[0, 0])
        az0, daz = -np.pi, 2 * np.pi / na_

        def az_columns(px, py):
            """indices of the azimuth columns covering the xy points (world), in the sensor's frame"""
            ang = np.arctan2(py - o[1], px - o[0]) - yaw
            ang = np.mod(ang + np.pi, 2 * np.pi) - np.pi
            lo_, hi_ = ang.min(), ang.max()
            if hi_ - lo_ > np.pi:  # wraps through +-pi
                pos, neg = ang[ang >= 0], ang[ang < 0]
                i0 = int(np.floor((pos.min() - az0) / daz)) - 1
                i1 = int(np.ceil((neg.max() - az0) / daz)) + 1
                return np.concatenate([np.arange(max(i0, 0), na_), np.arange(0, min(i1 + 1, na_))])
            i0 = int(np.floor((lo_ - az0) / daz)) - 1
            i1 = int(np.ceil((hi_ - az0) / daz)) + 1
            return np.arange(max(i0, 0), min(i1 + 1, na_))

        rows = np.arange(nb_)[:, None] * na_
        mb, ml = self.moving_boxes(frame * 0.1)
        boxes = np.concatenate([self.boxes, mb], 0)
        labels = np.concatenate([self.box_labels, ml], 0)
        if len(boxes):
            c = 0.5 * (boxes[:, :2] + boxes[:, 3:5])
            near = (np.abs(c[:, 0] - o[0]) < cull) & (np.abs(c[:, 1] - o[1]) < cull)
            boxes, labels, c = boxes[near], labels[near], c[near]
            gz = self.ground_h(c[:, 0], c[:, 1])
            inv = 1.0 / np.where(np.abs(D) < 1e-12, 1e-12, D)
            for bi in range(len(boxes)):
                bx0, by0, bz0, bx1, by1, bz1 = boxes[bi]
                if bx0 <= o[0] <= bx1 and by0 <= o[1] <= by1:
                    continue
                cols = az_columns(np.array([bx0, bx0, bx1, bx1]), np.array([by0, by1, by0, by1]))
                ridx = (rows + cols[None, :]).ravel()
                lo = np.array([bx0, by0, bz0 + gz[bi]])
                hi = np.array([bx1, by1, bz1 + gz[bi]])
                t0 = (lo[None, :] - o[None, :]) * inv[ridx]
                t1 = (hi[None, :] - o[None, :]) * inv[ridx]
                tn = np.minimum(t0, t1).max(-1)
                tf = np.maximum(t0, t1).min(-1)
                hit = (tn <= tf) & (tn > 0) & (tn < best_t[ridx])
                hidx = ridx[hit]
                best_t[hidx] = tn[hit]
                best_l[hidx] = labels[bi]
        if len(self.poles):
            P = self.poles[(np.abs(self.poles[:, 0] - o[0]) < 80.0) & (np.abs(self.poles[:, 1] - o[1]) < 80.0)]
            for px, py, pr, ph in P:
                cols = az_columns(np.array([px - pr, px - pr, px + pr, px + pr]), np.array([py - pr, py + pr, py - pr, py + pr]))
                ridx = (rows + cols[None, :]).ravel()
                Dr = D[ridx]
                a = Dr[:, 0] ** 2 + Dr[:, 1] ** 2
                ox, oy = o[0] - px, o[1] - py
                b = ox * Dr[:, 0] + oy * Dr[:, 1]
                cc = ox * ox + oy * oy - pr * pr
                disc = b * b - a * cc
                okc = (disc > 0) & (a > 1e-9)
                tt = np.where(okc, (-b - np.sqrt(np.where(okc, disc, 0))) / np.where(a > 1e-9, a, 1), np.inf)
                zz = o[2] + tt * Dr[:, 2]
                g = self.ground_h(px, py)
                okc &= (tt > 0) & (zz > g) & (zz < g + ph) & (tt < best_t[ridx])
                hidx = ridx[okc]
                best_t[hidx] = tt[okc]
                best_l[hidx] = LABEL_POLE
        keep = np.isfinite(best_t) & (best_t < lidar.max_range) & (best_t > lidar.min_range)
        tk = best_t[keep] + rng.normal(0, lidar.noise, size=int(keep.sum()))
        pts = lidar.dirs[keep] * tk[:, None]  # lidar frame (rotation of the body, origin at the sensor)
        out = np.empty((len(pts), 4), np.float32)
        out[:, :3] = pts
        out[:, 3] = best_l[keep].astype(np.float32)
        return out

    # ---- maps ---------------------------------------------------------------------------------
    def accumulate_map(self, frames, lidar, step=1.0, voxel=0.2, jitter=True):
        """Naive accumulated map in mapgen semantics + the per-frame scans and poses."""
        jr = np.random.default_rng(self.seed + 77) if jitter else None
        scans, poses, clouds = [], [], []
        for f in frames:
            p7 = self.pose(f, step, jitter_rng=jr)
            sc = self.cast(p7, lidar, f)
            scans.append(sc)
            poses.append(p7)
            keep = (sc[:, 0].astype(np.float64) ** 2 + sc[:, 1].astype(np.float64) ** 2) >= CAR_BODY_SIZE ** 2
            body = sc[keep].astype(np.float64)
            body[:, 2] += LIDAR_HEIGHT
            T = pose_to_matrix(p7)
            w = body.copy()
            w[:, :3] = body[:, :3] @ T[:3, :3].T + T[:3, 3]
            clouds.append(voxel_downsample(w.astype(np.float32), voxel))
        m = voxel_downsample(np.concatenate(clouds, 0), voxel)
        return m, scans, poses

    def sample_map(self, spacing=0.2, frames=range(0), step=1.0, trail_range=50.0, x_range=None, noise=0.01):
        """Directly sampled "accumulated + voxelised" map (for the multi-million-point configs):
        jittered `spacing` grids on the ground, facades, car bodies and poles, plus the trails of the
        moving objects over `frames` (label 252/254).  Returns (N,4) float32 in the map frame."""
        rng = np.random.default_rng(self.seed + 991)
        x0, x1 = (0.0, self.length) if x_range is None else x_range
        parts = []

        def jit(a):
            return a + rng.uniform(-0.45 * spacing, 0.45 * spacing, size=a.shape)

        xs = np.arange(x0, x1, spacing)
        for yc in self.street_y:
            ys = np.arange(yc - 12.0, yc + 12.0, spacing)
            X, Y = np.meshgrid(xs, ys, indexing="ij")
            X, Y = jit(X).ravel(), jit(Y).ravel()
            Z = self.ground_h(X, Y) + rng.normal(0, noise, X.shape)
            parts.append(np.stack([X, Y, Z, self.ground_label(Y).astype(np.float64)], 1))
        for (bx0, by0, bz0, bx1, by1, bz1), lab in zip(self.boxes, self.box_labels):
            if bx1 < x0 or bx0 > x1:
                continue
            parts.append(self._box_surface(bx0, by0, bz0, bx1, by1, bz1, lab, spacing, rng, noise))
        for px, py, pr, ph in self.poles:
            if px < x0 or px > x1:
                continue
            nz = max(int(ph / spacing), 1)
            na = max(int(2 * np.pi * pr / spacing), 4)
            A, Zp = np.meshgrid(np.linspace(0, 2 * np.pi, na, endpoint=False), (np.arange(nz) + 0.5) * spacing, indexing="ij")
            g = self.ground_h(px, py)
            parts.append(np.stack([px + pr * np.cos(A).ravel(), py + pr * np.sin(A).ravel(), g + Zp.ravel(),
                                   np.full(A.size, float(LABEL_POLE))], 1))
        jr = np.random.default_rng(self.seed + 77)
        for f in frames:
            p7 = self.pose(f, step, jitter_rng=jr)
            mb, ml = self.moving_boxes(f * 0.1)
            for (bx0, by0, bz0, bx1, by1, bz1), lab in zip(mb, ml):
                cx, cy = 0.5 * (bx0 + bx1), 0.5 * (by0 + by1)
                if (cx - p7[0]) ** 2 + (cy - p7[1]) ** 2 > trail_range ** 2:
                    continue
                parts.append(self._box_surface(bx0, by0, 0.15, bx1, by1, bz1, lab, spacing, rng, noise))
        m = np.concatenate(parts, 0).astype(np.float32)
        return m

    def _box_surface(self, bx0, by0, bz0, bx1, by1, bz1, lab, spacing, rng, noise):
        g = float(self.ground_h(0.5 * (bx0 + bx1), 0.5 * (by0 + by1)))
        z0, z1 = g + max(bz0, 0.0), g + bz1
        out = []
        zs = np.arange(z0 + 0.5 * spacing, z1, spacing)
        for (xa, ya, xb, yb) in ((bx0, by0, bx1, by0), (bx0, by1, bx1, by1), (bx0, by0, bx0, by1), (bx1, by0, bx1, by1)):
            ln = max(abs(xb - xa), abs(yb - ya))
            ts = (np.arange(int(ln / spacing) + 1) + 0.5) * spacing / max(ln, 1e-9)
            ts = ts[ts < 1.0]
            T, Z = np.meshgrid(ts, zs, indexing="ij")
            T = T.ravel() + rng.uniform(-0.4, 0.4, T.size) * spacing / max(ln, 1e-9)
            Z = Z.ravel() + rng.uniform(-0.4, 0.4, Z.size) * spacing
            out.append(np.stack([xa + (xb - xa) * T + rng.normal(0, noise, T.size), ya + (yb - ya) * T + rng.normal(0, noise, T.size), Z], 1))
        if bz1 < 3.0:  # roof of cars / heads of pedestrians are visible from the lidar
            xs = np.arange(bx0 + 0.5 * spacing, bx1, spacing)
            ys = np.arange(by0 + 0.5 * spacing, by1, spacing)
            X, Y = np.meshgrid(xs, ys, indexing="ij")
            out.append(np.stack([X.ravel(), Y.ravel(), np.full(X.size, z1) + rng.normal(0, noise, X.size)], 1))
        p = np.concatenate(out, 0)
        return np.concatenate([p, np.full((len(p), 1), float(lab))], 1)


def voxel_downsample(cloud, leaf):
    """Plain numpy voxel filter for *data generation* (centroid per voxel, label of the first point).
    Not PCL-exact and not used by any parity check."""
    c = np.asarray(cloud, np.float32)
    if len(c) == 0:
        return c.reshape(0, 4)
    k = np.floor(c[:, :3].astype(np.float64) / leaf).astype(np.int64)
    k -= k.min(0)
    dims = k.max(0) + 1
    key = (k[:, 2] * dims[1] + k[:, 1]) * dims[0] + k[:, 0]
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    first = np.concatenate([[True], key_s[1:] != key_s[:-1]])
    starts = np.flatnonzero(first)
    cnt = np.diff(np.concatenate([starts, [len(c)]]))
    sums = np.add.reduceat(c[order, :3].astype(np.float64), starts, axis=0)
    out = np.empty((len(starts), 4), np.float32)
    out[:, :3] = (sums / cnt[:, None]).astype(np.float32)
    out[:, 3] = c[order[starts], 3]
    return out


def is_dynamic(intensity):
    sem = np.asarray(intensity).astype(np.uint32) & 0xFFFF
    return (sem >= 252) & (sem <= 259)


# ---- ready-made configurations (BASELINE.json configs; parameters from config/*.yaml) -------------

SEQ_PARAMS = {
    # name: max_range rings sectors min_h max_h th_bin_max_h srt min_pts gf_dist iter lpr seeds lowest query_voxel interval
    "05": dict(max_range=60.0, num_rings=15, num_sectors=60, min_h=-1.3, max_h=3.2, th_bin_max_h=0.05,
               scan_ratio_threshold=0.3, minimum_num_pts=10, gf_dist_thr=0.15, gf_iter=3, gf_num_lpr=10,
               gf_th_seeds_height=0.5, num_lowest_pts=5, query_voxel_size=0.2, removal_interval=8),
    "00": dict(max_range=80.0, num_rings=20, num_sectors=108, min_h=-1.3, max_h=3.0, th_bin_max_h=0.2,
               scan_ratio_threshold=0.1, minimum_num_pts=6, gf_dist_thr=0.15, gf_iter=3, gf_num_lpr=20,
               gf_th_seeds_height=0.5, num_lowest_pts=5, query_voxel_size=0.2, removal_interval=4),
    "01": dict(max_range=60.0, num_rings=15, num_sectors=108, min_h=-1.3, max_h=3.0, th_bin_max_h=0.2,
               scan_ratio_threshold=0.2, minimum_num_pts=6, gf_dist_thr=0.15, gf_iter=3, gf_num_lpr=10,
               gf_th_seeds_height=0.5, num_lowest_pts=5, query_voxel_size=0.2, removal_interval=1),
    "02": dict(max_range=60.0, num_rings=15, num_sectors=60, min_h=-1.3, max_h=3.2, th_bin_max_h=0.05,
               scan_ratio_threshold=0.13, minimum_num_pts=20, gf_dist_thr=0.15, gf_iter=3, gf_num_lpr=20,
               gf_th_seeds_height=0.5, num_lowest_pts=5, query_voxel_size=0.2, removal_interval=5),
    "07": dict(max_range=80.0, num_rings=20, num_sectors=108, min_h=-0.8, max_h=3.1, th_bin_max_h=0.2,
               scan_ratio_threshold=0.20, minimum_num_pts=6, gf_dist_thr=0.125, gf_iter=3, gf_num_lpr=10,
               gf_th_seeds_height=0.5, num_lowest_pts=1, query_voxel_size=0.2, removal_interval=5),
    "large_scale_05": dict(max_range=80.0, num_rings=20, num_sectors=108, min_h=-1.3, max_h=3.0, th_bin_max_h=0.2,
                           scan_ratio_threshold=0.2, minimum_num_pts=6, gf_dist_thr=0.25, gf_iter=3, gf_num_lpr=20,
                           gf_th_seeds_height=0.5, num_lowest_pts=5, query_voxel_size=0.2, removal_interval=4),
    "ouster": dict(max_range=20.0, num_rings=20, num_sectors=60, min_h=-2.0, max_h=1.3, th_bin_max_h=-1.0,
                   scan_ratio_threshold=0.2, minimum_num_pts=5, gf_dist_thr=0.075, gf_iter=3, gf_num_lpr=12,
                   gf_th_seeds_height=0.5, num_lowest_pts=5, query_voxel_size=0.2, removal_interval=4),
}


def apply_params(p, name, **over):
    """fill a Params ctypes struct from SEQ_PARAMS[name]"""
    d = dict(SEQ_PARAMS[name])
    d.update(over)
    for k, v in d.items():
        setattr(p, k, v)
    p.version = over.get("version", 3)
    p.map_voxel_size = over.get("map_voxel_size", 0.2)
    return p
