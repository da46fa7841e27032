"""Deterministic synthetic workloads (no datasets are available offline).

Everything is generated with numpy's legacy ``RandomState`` (MT19937, frozen stream), so
the same seed gives the same bytes on the build container and on the GPU box.

* ``synth_descriptors``  -- "DB-only" mode of SURVEY.md §8(d): Scan Context images made
  directly as smooth random height fields (low-frequency sinusoid mixtures in
  (ring, sector), clipped to [0, 12] m), consecutive keyframes perturbed slowly so ring
  keys cluster like a real trajectory, ~25 % of the sectors zeroed in contiguous wedges
  (exercises the zero-column rule of descriptor.h:1523), and a fraction of keyframes that
  are column-rotated noisy copies of older ones (planted loops with known shift).
* ``synth_scan``         -- a LiDAR-like cloud: ground plane + boxes, for descriptor
  construction (K3) and ICP.
"""
import numpy as np

SEEDS = {"C1": 1001, "C2": 1002, "C3": 1003, "C4": 1004, "C5": 1005}


def synth_descriptors(n, R, S, seed=1002, n_basis=24, zero_wedge_frac=0.25, revisit_frac=0.01,
                      revisit_gap=150, noise=0.02, return_truth=False):
    """Returns float32 (n, R, S) descriptors (wire layout = ring-major rows)."""
    rs = np.random.RandomState(seed)
    r = (np.arange(R, dtype=np.float64) + 0.5) / R
    s = (np.arange(S, dtype=np.float64) + 0.5) / S
    fr = rs.uniform(0.3, 2.5, size=n_basis)
    fs = rs.randint(0, 5, size=n_basis).astype(np.float64)       # integer: periodic in the sector axis
    ph_r = rs.uniform(0, 2 * np.pi, size=n_basis)
    ph_s = rs.uniform(0, 2 * np.pi, size=n_basis)
    basis = (np.sin(2 * np.pi * fr[:, None] * r[None, :] + ph_r[:, None])[:, :, None] *
             np.cos(2 * np.pi * fs[:, None] * s[None, :] + ph_s[:, None])[:, None, :])   # (M, R, S)
    basis = basis.reshape(n_basis, R * S).astype(np.float32)

    # slow random walk of the mixture weights: keyframe i+1 = keyframe i perturbed
    steps = rs.standard_normal(size=(n, n_basis)).astype(np.float32)
    w = np.empty((n, n_basis), dtype=np.float32)
    acc = np.zeros(n_basis, dtype=np.float32)
    for i in range(n):
        acc = 0.97 * acc + 0.35 * steps[i]
        w[i] = acc
    field = 4.0 + 0.55 * (w @ basis)                                           # (n, R*S)
    field += noise * rs.standard_normal(size=field.shape).astype(np.float32)
    np.clip(field, 0.0, 12.0, out=field)
    field = field.reshape(n, R, S)

    # contiguous zero wedges (occluded sectors); the wedge drifts slowly along the trajectory
    wedge_len = int(round(zero_wedge_frac * S))
    if wedge_len > 0:
        start = (np.cumsum(rs.randint(-2, 3, size=n)) + rs.randint(0, S)) % S
        cols = (start[:, None] + np.arange(wedge_len)[None, :]) % S             # (n, wedge)
        field[np.arange(n)[:, None], :, cols] = 0.0
    # sparse empty cells (no return in the bin)
    field[rs.random_sample(field.shape) < 0.03] = 0.0

    truth = []
    n_rev = int(revisit_frac * n)
    if n_rev > 0 and n > revisit_gap + 10:
        cur = rs.choice(np.arange(revisit_gap + 5, n), size=n_rev, replace=False)
        for c in np.sort(cur):
            old = int(rs.randint(0, c - revisit_gap))
            sh = int(rs.randint(0, S))
            rolled = np.roll(field[old], sh, axis=1)
            pert = rolled + (noise * rs.standard_normal(size=rolled.shape)).astype(np.float32) * (rolled > 0)
            field[c] = np.clip(pert, 0.0, 12.0)
            truth.append((int(c), old, sh))
    out = np.ascontiguousarray(field, dtype=np.float32)
    return (out, truth) if return_truth else out


def synth_scan(n_points, seed=0, max_range=95.0, n_boxes=60, stride_floats=8, lidar_height=1.65):
    """A LiDAR-like cloud as (n, stride_floats) float32 records (x, y, z, pad, intensity, pad...).

    Ground plane at z = -lidar_height plus axis-aligned boxes; ~5 % of the points fall beyond
    80 m (dropped by the range cut, descriptor.h:1429).
    """
    rs = np.random.RandomState(seed)
    ang = rs.uniform(0.0, 2 * np.pi, size=n_points)
    rad = max_range * np.sqrt(rs.uniform(0.0004, 1.0, size=n_points))
    x = rad * np.cos(ang)
    y = rad * np.sin(ang)
    z = np.full(n_points, -lidar_height) + 0.02 * rs.standard_normal(n_points)
    bx = rs.uniform(-80, 80, size=n_boxes); by = rs.uniform(-80, 80, size=n_boxes)
    bw = rs.uniform(2, 12, size=n_boxes); bh = rs.uniform(0.5, 12, size=n_boxes)
    for k in range(n_boxes):
        inside = (np.abs(x - bx[k]) < bw[k]) & (np.abs(y - by[k]) < bw[k])
        z[inside] = -lidar_height + bh[k] * rs.uniform(0.0, 1.0, size=int(inside.sum()))
    cloud = np.zeros((n_points, stride_floats), dtype=np.float32)
    cloud[:, 0] = x; cloud[:, 1] = y; cloud[:, 2] = z
    if stride_floats > 4:
        cloud[:, 4] = rs.uniform(0, 255, size=n_points)
    return cloud


def synth_structured_cloud(n_points, seed=0, extent=40.0, stride_floats=8):
    """Points on a few planes and boxes: well-conditioned for ICP (used by the geometry tests)."""
    rs = np.random.RandomState(seed)
    n_ground = n_points // 2
    n_walls = n_points - n_ground
    pts = np.empty((n_points, 3), dtype=np.float64)
    pts[:n_ground, 0] = rs.uniform(-extent, extent, n_ground)
    pts[:n_ground, 1] = rs.uniform(-extent, extent, n_ground)
    pts[:n_ground, 2] = 0.05 * np.sin(0.3 * pts[:n_ground, 0]) + 0.01 * rs.standard_normal(n_ground)
    k = n_walls // 4
    sizes = [k, k, k, n_walls - 3 * k]
    o = n_ground
    for w, m in enumerate(sizes):
        u = rs.uniform(-extent, extent, m); h = rs.uniform(0, 6, m)
        off = (w - 1.5) * extent / 2.5
        if w % 2 == 0:
            pts[o:o + m] = np.stack([u, np.full(m, off) + 0.3 * np.sin(0.2 * u), h], axis=1)
        else:
            pts[o:o + m] = np.stack([np.full(m, off) + 0.3 * np.cos(0.25 * u), u, h], axis=1)
        o += m
    cloud = np.zeros((n_points, stride_floats), dtype=np.float32)
    cloud[:, :3] = pts.astype(np.float32)
    if stride_floats > 4:
        cloud[:, 4] = rs.uniform(0, 255, size=n_points)
    return cloud


def rigid_transform(roll, pitch, yaw, tx, ty, tz):
    """4x4 = Translation * Rz(yaw) * Ry(pitch) * Rx(roll)  (pcl::getTransformation, DM.h:241)."""
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [tx, ty, tz]
    return T
