"""Synthetic KITTI-/nuScenes-shaped batches (there are no datasets in this environment).

Follows the sample-dict contract of the reference's large-range loader
(`src/kitti_odometry_corr_lidarnone_proj.py`: rgb not normalised :248,:757-760; random
mis-calibration = yaw about the camera y axis + x/z translation :292-303,:386-406; GT = inverse
of that perturbation :608-612; intensity feature :682) with the shapes BASELINE.json names
(375x1242 RGB + 8192-point cloud).  SURVEY.md §8d describes the distributions.
"""
import math

import torch


def _quat_from_yaw_y(yaw):
    """rotation about +y by `yaw` -> quaternion (w,x,y,z) [B,4]"""
    h = 0.5 * yaw
    z = torch.zeros_like(yaw)
    return torch.stack([torch.cos(h), z, torch.sin(h), z], -1)


def lidar_scan(B, N, generator=None, device="cpu", beams=64, fup=2.0, fdown=-24.8, zero_rows=0,
               layout="scan"):
    """HDL-64-like scan [B,N,3] in the sensor frame (x forward, y left, z up): `beams` rings at
    elevations fdown + (i+0.5)/beams*(fup-fdown), N/beams azimuth steps jittered by +-0.4 cell, range
    U[3,60) m, rows shuffled; the last `zero_rows` rows are zero padding (loader :699-711).
    layout="centre": one point on every level-1 centre cell (stride 4x8 of the 64x1800 image)
    first, remaining points scattered — keeps all 3600 level-1 queries live at N=8192."""
    g = generator
    n_real = N - zero_rows
    steps = max(n_real // beams, 1)
    ring = torch.arange(n_real, device=device) % beams
    step = (torch.arange(n_real, device=device) // beams).float()
    # rings sit strictly inside (fdown, fup) and off the projection's row-bin edges
    el = math.radians(fdown) + (math.radians(fup) - math.radians(fdown)) * (ring.float() + 0.5) / beams
    jit = (torch.rand(B, n_real, generator=g, device=device) - 0.5) * 0.8
    az = (step.unsqueeze(0) + 0.5 + jit) / steps * (2 * math.pi) - math.pi
    el = el.unsqueeze(0).expand(B, -1)
    if layout == "centre":
        # first 16*225 points: exactly the centre of every (4h, 8w) cell of a 64x1800 image
        H, W, sh, sw = 64, 1800, 4, 8
        hh = torch.arange(0, H, sh, device=device); ww = torch.arange(0, W, sw, device=device)
        vres = (math.radians(fup) - math.radians(fdown)) / (H - 1)
        # row r holds elevations with H - trunc(el/vres + off) == r; aim at the middle of the bin
        el_c = (H - hh.float() + 0.5) * vres + math.radians(fdown)
        az_c = math.pi - (ww.float() + 0.5) * (2 * math.pi / W)
        el_g, az_g = torch.meshgrid(el_c, az_c, indexing="ij")
        nc = el_g.numel()
        if nc <= n_real:
            el = el.clone(); az = az.clone()
            el[:, :nc] = el_g.reshape(1, -1); az[:, :nc] = az_g.reshape(1, -1)
    r = 3.0 + 57.0 * torch.rand(B, n_real, generator=g, device=device)
    pts = torch.stack([r * torch.cos(el) * torch.cos(az), r * torch.cos(el) * torch.sin(az), r * torch.sin(el)], -1)
    perm = torch.argsort(torch.rand(B, n_real, generator=g, device=device), dim=1)
    pts = torch.gather(pts, 1, perm.unsqueeze(-1).expand(-1, -1, 3))
    if zero_rows:
        pts = torch.cat([pts, torch.zeros(B, zero_rows, 3, device=device)], 1)
    return pts.contiguous()


def spherical_cells(xyz, H, W, fup=2.0, fdown=-24.8):
    """cell index row*W+col [B,N] of every point under the projection rule of
    src/projectPN/utils.py:144-155 (plain torch; used only to prepare synthetic clouds)."""
    az_res = 2 * math.pi / W
    vdown, vup = math.radians(fdown), math.radians(fup)
    vres = (vup - vdown) / (H - 1)
    r = torch.norm(xyz, p=2, dim=2)
    col = ((math.pi - torch.atan2(xyz[..., 1], xyz[..., 0])) / az_res).long().clamp(0, W - 1)
    row = (H - (torch.asin(xyz[..., 2] / r) / vres + (-vdown / vres)).long()).clamp(0, H - 1)
    return row * W + col


def drop_duplicate_cells(xyz, H, W, fup=2.0, fdown=-24.8):
    """zero every point whose range-image cell is already taken by an earlier point.  The
    reference scatters with index_put_, whose result under duplicate cells is unspecified
    (and not reproducible on CPU torch 2.10 either); fixtures therefore use duplicate-free clouds."""
    cells = spherical_cells(xyz, H, W, fup, fdown)
    out = xyz.clone()
    for b in range(xyz.shape[0]):
        order = torch.argsort(cells[b], stable=True)
        sc = cells[b][order]
        dup = torch.zeros_like(sc, dtype=torch.bool)
        dup[1:] = sc[1:] == sc[:-1]
        out[b, order[dup]] = 0.0
    return out


def above_fup_points(raw, count, fup=2.0):
    """overwrite the first `count` rows of every cloud with returns ABOVE the upper field-of-view limit (elevation fup + 1.5 ...
    fup + 2 degrees): the projection clamps them into image row 0 (utils.py:152-154), the row the NaN cell (0, W/2) of the zero
    padding rows lives in; a quarter of them look straight ahead (azimuth ~ 0 -> columns W/2 - 1, W/2) so that real points and
    padding rows compete for that cell in point order."""
    B = raw.shape[0]
    i = torch.arange(count, device=raw.device, dtype=torch.float32)
    el = math.radians(fup + 1.5) + math.radians(0.5) * (i / max(count - 1, 1))
    az = torch.where(i % 4 == 0, (i / count - 0.5) * 0.006, (i / count - 0.5) * 5.0)
    r = 5.0 + i
    pts = torch.stack([r * torch.cos(el) * torch.cos(az), r * torch.cos(el) * torch.sin(az), r * torch.sin(el)], -1)
    out = raw.clone()
    out[:, :count] = pts.unsqueeze(0).expand(B, -1, -1)
    return out


def drop_bin_edge_points(xyz, H, W, fup=2.0, fdown=-24.8, margin=2e-3):
    """zero every point whose fractional row / column coordinate under utils.py:144-155 lies within `margin` cells of a bin edge:
    host libm and the device's OCML differ by ulps in atan2 / asin, so such a point may land in either cell (SURVEY App. A.4:
    "keep edge-adjacent points out of exact-match fixtures").  The zeroed rows become padding rows in the MIDDLE of the cloud."""
    az_res = 2 * math.pi / W
    vdown, vup = math.radians(fdown), math.radians(fup)
    vres = (vup - vdown) / (H - 1)
    d = xyz.double()
    r = d.norm(dim=2).clamp_min(1e-30)
    fcol = (math.pi - torch.atan2(d[..., 1], d[..., 0])) / az_res
    frow = torch.asin((d[..., 2] / r).clamp(-1, 1)) / vres + (-vdown / vres)
    edge = ((fcol - fcol.round()).abs() < margin) | ((frow - frow.round()).abs() < margin)
    return xyz * (~edge).unsqueeze(-1)


def make_batch(B, N=8192, img_h=375, img_w=1242, seed=0, device="cpu", zero_rows=0, beams=64, fup=2.0,
               fdown=-24.8, layout="scan", unique_cells=None, edge_margin=None, above_fup=0):
    """-> dict with the reference loader's keys: rgb, lidar, raw_point_xyz, lidar_feats,
    init_intrinsic, init_extrinsic, decalib_real_gt (quat w,x,y,z), decalib_dual_gt (trans).
    `edge_margin` = (H, W, margin): drop_bin_edge_points; `above_fup` = count: above_fup_points — the loader-shaped fixtures
    (N = 150 000 with 30 000 padding rows, duplicate cells kept: kitti_odometry_corr_lidarnone_proj.py:264,699-711)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rgb = torch.rand(B, 3, img_h, img_w, generator=g, device=device) * 255.0
    raw = lidar_scan(B, N, g, device, beams=beams, fup=fup, fdown=fdown, zero_rows=zero_rows, layout=layout)
    if above_fup:
        raw = above_fup_points(raw, above_fup, fup)
    if edge_margin is not None:
        raw = drop_bin_edge_points(raw, edge_margin[0], edge_margin[1], fup, fdown, edge_margin[2])
    if unique_cells is not None:                       # (H, W) of the range image
        raw = drop_duplicate_cells(raw, unique_cells[0], unique_cells[1], fup, fdown)
    # velodyne -> camera axes: x_c = -y_v, y_c = -z_v, z_c = x_v
    Tr = torch.tensor([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]], device=device)
    yaw = (torch.rand(B, generator=g, device=device) * 2 - 1) * (2 * math.pi)
    tx = (torch.rand(B, generator=g, device=device) * 2 - 1) * 10.0
    tz = (torch.rand(B, generator=g, device=device) * 2 - 1) * 10.0
    c, s = torch.cos(yaw), torch.sin(yaw)
    z, o = torch.zeros_like(c), torch.ones_like(c)
    R = torch.stack([torch.stack([c, z, s], -1), torch.stack([z, o, z], -1), torch.stack([-s, z, c], -1)], -2)
    t = torch.stack([tx, z, tz], -1)
    cam = torch.einsum("ij,bnj->bni", Tr, raw)
    lidar = torch.einsum("bij,bnj->bni", R, cam) + t.unsqueeze(1)
    lidar = lidar * (raw != 0).any(-1, keepdim=True)      # padding rows stay all-zero in both frames
    ext = torch.eye(4, device=device).repeat(B, 1, 1)
    ext[:, :3, :3] = R @ Tr
    ext[:, :3, 3] = t
    # ground truth = inverse perturbation: q = quat(R^T), t = -R^T t
    q_gt = _quat_from_yaw_y(-yaw)
    t_gt = -torch.einsum("bji,bj->bi", R, t)
    K = torch.tensor([[718.856, 0.0, 607.19], [0.0, 718.856, 185.22], [0.0, 0.0, 1.0]], device=device)
    return {
        "rgb": rgb, "lidar": lidar.contiguous(), "raw_point_xyz": raw,
        "lidar_feats": torch.rand(B, N, 1, generator=g, device=device),
        "init_intrinsic": K.unsqueeze(0).repeat(B, 1, 1), "init_extrinsic": ext,
        "decalib_real_gt": q_gt, "decalib_dual_gt": t_gt,
    }


def write_kitti_tree(root, frames=16, seqs=(0,), seed=7, n_points=120000, img_h=376, img_w=1241):
    """A synthetic KITTI odometry tree in the layout the reference loader walks (kitti_odometry_corr_lidarnone_proj.py:38-77:
    velodyne .bin scans, colour frames as .npy, calib.txt, the SNR .npy it also opens) — `frames` frames per sequence with
    HDL-64-like scans of `n_points` points.  For `bench.py --data tree` (a loader-inclusive step) and the data-pipeline tests;
    seeded, so nothing has to be shipped."""
    import os
    import numpy as np
    rs = np.random.RandomState(seed)
    P2 = np.array([[718.856, 0, 607.1928, 45.38225], [0, 718.856, 185.2157, -0.1130887], [0, 0, 1, 0.003779761]])
    Tr = np.array([[4.276802e-04, -9.999672e-01, -8.084491e-03, -1.198459e-02], [-7.210626e-03, 8.081198e-03, -9.999413e-01, -5.403984e-02],
                   [9.999738e-01, 4.859485e-04, -7.206933e-03, -2.921968e-01]])
    for seq in seqs:
        vel = os.path.join(root, "data_odometry_velodyne", "dataset", "%02d" % seq, "velodyne")
        snr = os.path.join(root, "data_odometry_velodyne_deepi2p_new", "data_odometry_velodyne_NWU", "sequences", "%02d" % seq, "snr0.6")
        img = os.path.join(root, "kitti_processed_DeepI2P", "data_odometry_color_npy", "sequences", "%02d" % seq, "image_2")
        cal = os.path.join(root, "kitti_processed_DeepI2P", "data_odometry_calib", "dataset", "sequences", "%02d" % seq)
        for d in (vel, snr, img, cal):
            os.makedirs(d, exist_ok=True)
        for i in range(frames):
            n = n_points + 13 * i
            az = rs.rand(n) * 2 * np.pi
            el = np.deg2rad(-24.8 + 26.8 * rs.randint(0, 64, n) / 63.0)
            r = 3.0 + 57.0 * rs.rand(n)
            scan = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el), rs.rand(n)], 1).astype(np.float32)
            scan.tofile(os.path.join(vel, "%06d.bin" % i))
            np.save(os.path.join(snr, "%06d.npy" % i), np.zeros((7, 1), np.float32))
            np.save(os.path.join(img, "%06d.npy" % i), rs.randint(0, 256, (img_h, img_w, 3)).astype(np.uint8))
        with open(os.path.join(cal, "calib.txt"), "w") as f:
            for k, m in (("P0", P2), ("P1", P2), ("P2", P2), ("P3", P2), ("Tr", Tr)):
                f.write(k + ": " + " ".join("%.9e" % v for v in m.reshape(-1)) + "\n")
