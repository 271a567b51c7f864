"""Seeded synthetic inputs shared by the CPU and GPU tests."""
import numpy as np
import torch


def range_image(B, H, W, seed, empty_frac=0.2, lattice=False, scale=20.0):
    """[B,H,W,3] f32 range-image-like tensor with `empty_frac` all-zero cells.
    lattice=True puts points on a coarse integer lattice so that equal distances (ties) are
    everywhere — the case the reference resolves by selection-sort swap history."""
    g = torch.Generator().manual_seed(seed)
    if lattice:
        img = torch.randint(-3, 4, (B, H, W, 3), generator=g).float()
    else:
        # smooth surface + noise so neighbours are close in 3-D like a LiDAR scan
        az = torch.linspace(-np.pi, np.pi, W).view(1, 1, W)
        el = torch.linspace(-0.4, 0.05, H).view(1, H, 1)
        r = scale * (0.5 + torch.rand(B, H, W, generator=g))
        img = torch.stack([r * torch.cos(el) * torch.cos(az), r * torch.cos(el) * torch.sin(az),
                           r * torch.sin(el).expand(B, H, W)], -1)
    keep = (torch.rand(B, H, W, 1, generator=g) >= empty_frac).float()
    return (img * keep).contiguous()


def run_fcsk(backend, xyz1, xyz2, idx_n2, kH, kW, K, flag, distance, stride_h, stride_w, random_hw=None, init=0):
    B, H, W, _ = xyz1.shape
    sh, sw = xyz2.shape[1:3]
    N = idx_n2.shape[1]
    dev = xyz1.device
    if random_hw is None:
        random_hw = torch.arange(kH * kW, dtype=torch.int32, device=dev)
    sb = torch.full((B, N, K, 1), init, dtype=torch.long, device=dev)
    sh_ = sb.clone(); sw_ = sb.clone()
    v1 = torch.zeros(B, N, kH * kW, 1, device=dev); v2 = torch.zeros_like(v1)
    m = torch.full((B, N, K, 1), float(init), device=dev)
    backend.fused_conv_select_k(xyz1, xyz2, idx_n2, random_hw, H, W, N, kH, kW, K, flag, distance,
                                stride_h, stride_w, sb, sh_, sw_, v1, v2, m, sh, sw)
    return sb, sh_, sw_, m, v1, v2


def stride_grid(B, out_h, out_w, sh, sw):
    h = torch.arange(0, out_h * sh, sh, dtype=torch.int32)
    w = torch.arange(0, out_w * sw, sw, dtype=torch.int32)
    g = torch.stack(torch.meshgrid(h, w, indexing="ij"), -1).reshape(1, -1, 2)
    return g.expand(B, -1, -1).contiguous()


def cloud(B, N, seed, dup_frac=0.0, zero_frac=0.0):
    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * 40.0
    if dup_frac > 0:      # exact duplicates -> FPS ties
        nd = int(N * dup_frac)
        src = torch.randint(0, N, (nd,), generator=g)
        dst = torch.randint(0, N, (nd,), generator=g)
        pts[:, dst] = pts[:, src]
    if zero_frac > 0:
        nz = int(N * zero_frac)
        pts[:, N - nz:] = 0.0
    return pts.contiguous()


def synthetic_state(shapes, seed=0):
    """Deterministic weights keyed by parameter NAME (independent of module construction
    order), so the reference model (tools/gen_golden.py) and ours load identical values
    without shipping a 3.4 MB state_dict.  `shapes`: list of (key, shape)."""
    import hashlib
    out = {}
    for key, shape in shapes:
        h = int.from_bytes(hashlib.sha256(f"{seed}:{key}".encode()).digest()[:4], "little")
        g = torch.Generator().manual_seed(h)
        shape = tuple(shape)
        if key.endswith("num_batches_tracked"):
            t = torch.zeros(shape, dtype=torch.long)
        elif key in ("sq", "sx"):
            t = torch.tensor([-2.5 if key == "sq" else 0.0])
        elif key.endswith("running_var"):
            t = 1.0 + 0.2 * torch.rand(shape, generator=g)
        elif key.endswith("running_mean"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) >= 2:                                  # conv weights
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) * (1.5 / fan_in) ** 0.5
        elif "bn" in key.split(".")[-2] or key.split(".")[-2].isdigit() and key.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)    # BN gamma
        else:
            t = 0.1 * torch.randn(shape, generator=g)          # biases / BN beta
        out[key] = t
    return out


def make_kitti_tree(root, seed=7, n_points=3000, img_h=372, img_w=1030):
    """A minimal synthetic KITTI tree in the layout of `make_dataset` (kitti_odometry_corr_lidarnone_proj.py:38-77), val
    split (sequences 09 and 10, one frame each): velodyne .bin, colour .npy, calib.txt, the SNR .npy the loader also opens.
    Seeded, so the test regenerates the same files instead of shipping them."""
    import os
    import numpy as np
    rs = np.random.RandomState(seed)
    for seq in (9, 10):
        vel = os.path.join(root, "data_odometry_velodyne", "dataset", "%02d" % seq, "velodyne")
        snr = os.path.join(root, "data_odometry_velodyne_deepi2p_new", "data_odometry_velodyne_NWU", "sequences", "%02d" % seq, "snr0.6")
        img = os.path.join(root, "kitti_processed_DeepI2P", "data_odometry_color_npy", "sequences", "%02d" % seq, "image_2")
        cal = os.path.join(root, "kitti_processed_DeepI2P", "data_odometry_calib", "dataset", "sequences", "%02d" % seq)
        for d in (vel, snr, img, cal):
            os.makedirs(d, exist_ok=True)
        n = n_points + 17 * seq
        scan = np.concatenate([(rs.rand(n, 3) - 0.5) * np.array([80.0, 80.0, 6.0]), rs.rand(n, 1)], 1).astype(np.float32)
        scan.tofile(os.path.join(vel, "000000.bin"))
        np.save(os.path.join(snr, "000000.npy"), rs.randn(7, n).astype(np.float32))
        np.save(os.path.join(img, "000000.npy"), rs.randint(0, 256, (img_h, img_w, 3)).astype(np.uint8))
        P2 = np.array([[718.856, 0, 607.1928, 45.38225], [0, 718.856, 185.2157, -0.1130887], [0, 0, 1, 0.003779761]])
        Tr = np.array([[4.276802e-04, -9.999672e-01, -8.084491e-03, -1.198459e-02], [-7.210626e-03, 8.081198e-03, -9.999413e-01, -5.403984e-02],
                       [9.999738e-01, 4.859485e-04, -7.206933e-03, -2.921968e-01]])
        with open(os.path.join(cal, "calib.txt"), "w") as f:
            for k, m in (("P0", P2), ("P1", P2), ("P2", P2), ("P3", P2), ("Tr", Tr)):
                f.write(k + ": " + " ".join("%.9e" % v for v in m.reshape(-1)) + "\n")




def make_nuscenes_tree(root, split_dir, seed=11, n_points=4000, n_samples=2, img_h=900, img_w=1600):
    """A minimal synthetic nuScenes tree in the layout `src/nuscenes_loader_proj_nolidar.py` reads: `<root>/trainval/`
    with LIDAR_TOP `.pcd.bin` sweeps (float32 x 5 per point) and camera images (PNG: lossless), plus the pickled split
    lists `[((lidar file, camera file), K, Tr, night_tag)]` under `split_dir` (the reference opens
    ./nuScenes_datasplit/<mode>_dataset_randominfo_proj_day.list relative to the working directory).  Seeded."""
    import os
    import pickle
    import numpy as np
    from PIL import Image
    rs = np.random.RandomState(seed)
    os.makedirs(split_dir, exist_ok=True)
    K = np.array([[1266.417203046554, 0.0, 816.2670197447984], [0.0, 1266.417203046554, 491.50706579294757], [0.0, 0.0, 1.0]])
    entries = {"train": [], "val": [], "test": []}
    for i in range(n_samples):
        for mode, sub in (("val", "trainval"), ("train", "trainval"), ("test", "test")):
            if mode != "val" and i > 0:
                continue
            ld = os.path.join(root, sub, "samples", "LIDAR_TOP"); cd = os.path.join(root, sub, "samples", "CAM_FRONT")
            os.makedirs(ld, exist_ok=True); os.makedirs(cd, exist_ok=True)
            n = n_points + 31 * i
            xyz = (rs.rand(n, 3) - 0.5) * np.array([100.0, 100.0, 10.0]) + np.array([0.0, 0.0, -1.0])
            xyz[: n // 10] *= np.array([0.02, 0.06, 1.0])                    # a cluster on the ego vehicle (filtered out)
            scan = np.concatenate([xyz, rs.rand(n, 1) * 255.0, rs.randint(0, 32, (n, 1))], 1).astype(np.float32)
            lf = os.path.join("samples", "LIDAR_TOP", f"{mode}_{i:03d}.pcd.bin"); cf = os.path.join("samples", "CAM_FRONT", f"{mode}_{i:03d}.png")
            scan.tofile(os.path.join(root, sub, lf))
            Image.fromarray(rs.randint(0, 256, (img_h, img_w, 3)).astype(np.uint8)).save(os.path.join(root, sub, cf))
            a = 0.02 * (i + 1)
            R = np.array([[np.cos(a), -np.sin(a), 0.0], [0.0, 0.0, -1.0], [np.sin(a), np.cos(a), 0.0]])
            Tr = np.identity(4); Tr[:3, :3] = R; Tr[:3, 3] = [0.01 * i, -0.32, -0.75]
            entries[mode].append(((lf, cf), K.copy(), Tr, False))
    for mode, lst in entries.items():
        with open(os.path.join(split_dir, f"{mode}_dataset_randominfo_proj_day.list"), "wb") as f:
            pickle.dump(lst, f)
