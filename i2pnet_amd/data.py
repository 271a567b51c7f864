"""Real-data input pipeline of the large-range model: KITTI odometry velodyne `.bin` + colour image
`.npy` + `calib.txt` -> the sample dict `RegNet_v2` / `Trainer.step` consume, with everything per point or per pixel done
ON THE DEVICE.

Reference: `src/kitti_odometry_corr_lidarnone_proj.py` — directory layout `make_dataset` (:38-77), calibration parsing
`read_calib` (:205-229), the random mis-calibration `generate_random_transform` / `angles2rotation_matrix` (:80-91,
:386-406), its inverse as ground truth (:608-612, `utils.extrinsic_to_dual_quat` / `rotmat_to_quat` utils.py:245-322),
point shuffle (:532-533), training jitter (:332-343, :622-626), `init_extrinsic @ [pc;1]` (:649-656), zero padding to
150 000 rows (:699-711), image crop of the top 50 rows, x0.5 resize, 160x512 crop and the intrinsic bookkeeping
(:713-747), keys of the sample (:770-787).

What the reference does in numpy / cv2 on DataLoader workers per sample (150 000-row float64 matmuls, concatenations, a
cv2 resize) happens here in a handful of device kernels per BATCH; the host only reads the two files and draws the six
random numbers of the perturbation.  `Prefetcher` overlaps file reading (a thread), the pinned host->device copies (a
copy stream) and the device-side build with the training step of the previous batch.
"""
import math
import os
import random
import threading
from queue import Queue

import numpy as np
import torch


# --------------------------------------------------------------------------------------------------------------
# host side: files, calibration, the perturbation and its ground truth (a few dozen flops per sample)
# --------------------------------------------------------------------------------------------------------------
def read_calib(calib_file_path):
    """-> (Tr [3,4] f32 velodyne->cam0, intrinsic [3,3] of camera 2, P [4,4] cam0->cam2 translation)
    kitti_odometry_corr_lidarnone_proj.py:205-229"""
    Tr = intrinsic = P = None
    with open(calib_file_path, "r") as f:
        for line in f.readlines():
            key = line[0:2]
            if key not in ("Tr", "P2"):
                continue
            mat = np.array(line[4:].split(), dtype=np.float64).reshape(3, 4).astype(np.float32)
            if key == "Tr":
                Tr = mat
            else:
                K = mat[0:3, 0:3]
                fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
                intrinsic = np.asarray([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
                tz = mat[2, 3]
                tx = (mat[0, 3] - cx * tz) / fx
                ty = (mat[1, 3] - cy * tz) / fy
                P = np.identity(4)
                P[0:3, 3] = np.asarray([tx, ty, tz])
    return Tr, intrinsic, P


def angles2rotation_matrix(angles):
    """Rz @ Ry @ Rx (:80-91)"""
    cx, sx = np.cos(angles[0]), np.sin(angles[0])
    cy, sy = np.cos(angles[1]), np.sin(angles[1])
    cz, sz = np.cos(angles[2]), np.sin(angles[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return np.dot(Rz, np.dot(Ry, Rx))


def rotmat_to_quat(R):
    """(w, x, y, z) with the branch structure of the reference (utils.py:245-273)"""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    return np.asarray(q)


def random_transform(rng, tx=10.0, ty=0.0, tz=10.0, rx=0.0, ry=2 * math.pi, rz=0.0):
    """the mis-calibration Pr [4,4] (`generate_random_transform`, :386-406; amplitudes :292-303): `rng` = `random`-like"""
    t = [rng.uniform(-tx, tx), rng.uniform(-ty, ty), rng.uniform(-tz, tz)]
    angles = [rng.uniform(-rx, rx), rng.uniform(-ry, ry), rng.uniform(-rz, rz)]
    Pr = np.identity(4, dtype=np.float64)
    Pr[0:3, 0:3] = angles2rotation_matrix(angles)
    Pr[0:3, 3] = t
    return Pr


def kitti_file_list(root_path, mode):
    """[(velodyne .bin, image .npy, calib.txt, seq, frame)] in the directory layout of `make_dataset` (:38-77)"""
    seqs = {"train": list(range(9)), "val": [9, 10], "test": [7, 8]}[mode]
    out = []
    for seq in seqs:
        pc = os.path.join(root_path, "data_odometry_velodyne", "dataset", "%02d" % seq, "velodyne")
        img = os.path.join(root_path, "kitti_processed_DeepI2P", "data_odometry_color_npy", "sequences", "%02d" % seq, "image_2")
        cal = os.path.join(root_path, "kitti_processed_DeepI2P", "data_odometry_calib", "dataset", "sequences", "%02d" % seq, "calib.txt")
        if not os.path.isdir(pc):
            continue
        for name in sorted(os.listdir(pc)):
            if name.endswith(".bin"):
                i = int(name[:-4])
                out.append((os.path.join(pc, name), os.path.join(img, "%06d.npy" % i), cal, seq, i))
    return out


class KittiOdometryFiles(torch.utils.data.Dataset):
    """Host half of a sample: the raw scan [N,4] f32, the raw image [H,W,3] u8 and the calibration — no per-point work."""

    def __init__(self, root_path, mode="train"):
        self.mode = mode
        self.items = kitti_file_list(root_path, mode)
        self._calib = {}

    def __len__(self):
        return len(self.items)

    def __getitem__(self, index):
        pc_path, img_path, calib_path, seq, i = self.items[index]
        if calib_path not in self._calib:
            self._calib[calib_path] = read_calib(calib_path)
        Tr, K, P2 = self._calib[calib_path]
        scan = np.fromfile(pc_path, np.float32).reshape(-1, 4)                       # :524
        rgb = np.load(img_path)                                                      # :617
        return {"scan": torch.from_numpy(scan), "image": torch.from_numpy(np.ascontiguousarray(rgb)), "Tr": Tr, "K": K, "P2": P2,
                "index": index, "path_info": "%02d %06d %06d" % (seq, i, i)}


# --------------------------------------------------------------------------------------------------------------
# device side
# --------------------------------------------------------------------------------------------------------------
def resize_half_u8(img):
    """cv2.resize(img, (round(w/2), round(h/2)), INTER_LINEAR) of a uint8 image [H,W,3] (on any device): at scale 1/2 the
    bilinear sample point of output pixel x is 2x + 0.5, i.e. the mean of source pixels 2x, 2x+1 (both axes), which
    cv2's fixed-point path rounds half up."""
    H, W, _ = img.shape
    oh, ow = int(round(H * 0.5)), int(round(W * 0.5))
    # (an odd trailing row / column is sampled with its predecessor, like cv2's border clamp of the +1 neighbour)
    ys = torch.arange(oh, device=img.device) * 2
    xs = torch.arange(ow, device=img.device) * 2
    y1 = (ys + 1).clamp_max(H - 1); x1 = (xs + 1).clamp_max(W - 1)
    v = img.to(torch.int32)
    s = v[ys][:, xs] + v[ys][:, x1] + v[y1][:, xs] + v[y1][:, x1]
    return ((s + 2) // 4).to(torch.uint8)


class DeviceSampleBuilder:
    """Turns a list of host half-samples (`KittiOdometryFiles`) into ONE batched sample dict on `device`.

    Padding rows stay exactly zero in `lidar`, `raw_point_xyz` and `lidar_feats` (:699-711); `rgb` is float [B,3,160,512]
    in 0..255 (the reference does not normalise, :757-760)."""

    def __init__(self, device, mode="train", sample_point=150000, img_H=160, img_W=512, img_scale=0.5, crop_top=50,
                 rng=None, jitter=True):
        self.device, self.mode = torch.device(device), mode
        self.sample_point, self.img_H, self.img_W, self.img_scale, self.crop_top = sample_point, img_H, img_W, img_scale, crop_top
        self.rng = rng or random
        self.jitter = jitter and mode == "train"

    def perturbation(self, host):
        """host-side part of a sample: Pr, init_extrinsic, ground truth (float64 4x4 algebra, :583-612)"""
        Tr = np.vstack((host["Tr"], [0, 0, 0, 1]))
        Pc = np.dot(host["P2"], Tr)
        Pr = host.get("Pr")
        if Pr is None:
            Pr = random_transform(self.rng)
        calib_extrinsic = np.linalg.inv(Pr)[:3, :]
        q = rotmat_to_quat(calib_extrinsic[:3, :3])
        t = calib_extrinsic[:, 3]
        init_extrinsic = np.dot(Pr, Pc)[:3, :]
        return Pr, init_extrinsic, q, t

    @torch.no_grad()
    def __call__(self, hosts, stream=None):
        dev = self.device
        B = len(hosts)
        SP = self.sample_point
        lidar = torch.zeros(B, SP, 3, device=dev)
        raw = torch.zeros(B, SP, 3, device=dev)
        feats = torch.zeros(B, SP, 1, device=dev)
        rgb = torch.empty(B, 3, self.img_H, self.img_W, device=dev)
        ext = torch.empty(B, 3, 4, device=dev); intr = torch.empty(B, 3, 3, device=dev)
        q_gt = torch.empty(B, 4, device=dev); t_gt = torch.empty(B, 3, device=dev)
        Ks, paths, idxs = [], [], []
        for b, host in enumerate(hosts):
            Pr, E, q, t = self.perturbation(host)
            scan = host["scan"].to(dev, non_blocking=True)                            # [N,4]
            n = min(scan.shape[0], SP)
            perm = host.get("perm")
            perm = torch.randperm(scan.shape[0], device=dev) if perm is None else torch.as_tensor(perm, device=dev)   # :532-533
            scan = scan[perm][:n]
            pc = scan[:, :3]
            if self.jitter:                                                           # :332-343: N(0, 0.01^2) clipped at 5 cm
                pc = pc + torch.clamp(0.01 * torch.randn_like(pc), -0.05, 0.05)
            Ed = torch.as_tensor(E, dtype=torch.float64, device=dev)
            cam = (pc.double() @ Ed[:, :3].t() + Ed[:, 3]).float()                    # :654-656 (float64 product, float32 result)
            lidar[b, :n] = cam; raw[b, :n] = pc; feats[b, :n, 0] = scan[:, 3]
            # image: drop the top rows, halve, crop (:713-747)
            img = host["image"].to(dev, non_blocking=True)[self.crop_top:]
            K = np.copy(host["K"]).astype(np.float64); K[1, 2] -= self.crop_top
            if self.img_scale == 0.5:
                img = resize_half_u8(img)
            else:
                h0, w0, _ = img.shape
                img = torch.nn.functional.interpolate(img.permute(2, 0, 1)[None].float(), size=(int(round(h0 * self.img_scale)), int(round(w0 * self.img_scale))),
                                                      mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
            K[0, 0] *= self.img_scale; K[0, 2] *= self.img_scale; K[1, 1] *= self.img_scale; K[1, 2] *= self.img_scale
            h, w, _ = img.shape
            if self.mode == "train":
                dx, dy = self.rng.randint(0, w - self.img_W), self.rng.randint(0, h - self.img_H)
            else:
                dx, dy = int((w - self.img_W) / 2), int((h - self.img_H) / 2)
            dx, dy = host.get("crop", (dx, dy))
            rgb[b] = img[dy:dy + self.img_H, dx:dx + self.img_W].permute(2, 0, 1).float()
            K[0, 2] -= dx; K[1, 2] -= dy
            Ks.append(K)
            ext[b] = torch.as_tensor(E, dtype=torch.float32); q_gt[b] = torch.as_tensor(q, dtype=torch.float32)
            t_gt[b] = torch.as_tensor(t, dtype=torch.float32)
            paths.append(host["path_info"]); idxs.append(host["index"])
        intr.copy_(torch.as_tensor(np.stack(Ks), dtype=torch.float32))
        return {"rgb": rgb, "lidar": lidar, "raw_point_xyz": raw, "lidar_feats": feats, "init_extrinsic": ext, "init_intrinsic": intr,
                "decalib_real_gt": q_gt, "decalib_dual_gt": t_gt, "path_info": paths, "index": idxs,
                "resize_img": torch.tensor([[self.img_scale, self.img_scale]] * B)}


class Prefetcher:
    """Iterates device sample dicts: a reader thread pulls half-samples from `dataset` (file I/O, pinned memory), the
    device build of batch i+1 runs on a side stream under the step of batch i; `__next__` makes the current stream wait
    for that side stream, no host synchronisation."""

    def __init__(self, dataset, builder, batch_size, indices=None, depth=2):
        self.dataset, self.builder, self.batch_size = dataset, builder, batch_size
        self.indices = list(range(len(dataset))) if indices is None else list(indices)
        self.depth = depth
        self.cuda = builder.device.type == "cuda"
        self.stream = torch.cuda.Stream(builder.device) if self.cuda else None

    def _reader(self, q):
        try:
            for s in range(0, len(self.indices) - self.batch_size + 1, self.batch_size):
                hosts = [self.dataset[i] for i in self.indices[s:s + self.batch_size]]
                if self.cuda:
                    for h in hosts:
                        h["scan"] = h["scan"].pin_memory(); h["image"] = h["image"].pin_memory()
                q.put(hosts)
        finally:
            q.put(None)

    def __iter__(self):
        q = Queue(maxsize=self.depth)
        threading.Thread(target=self._reader, args=(q,), daemon=True).start()
        pending = None

        def build(hosts):
            if not self.cuda:
                return self.builder(hosts), None
            with torch.cuda.stream(self.stream):
                out = self.builder(hosts)
                ev = torch.cuda.Event(); ev.record(self.stream)
            return out, ev
        hosts = q.get()
        if hosts is not None:
            pending = build(hosts)
        while pending is not None:
            out, ev = pending
            hosts = q.get()
            pending = build(hosts) if hosts is not None else None           # next batch's copies / kernels start now
            if ev is not None:
                torch.cuda.current_stream(self.builder.device).wait_event(ev)
                for v in out.values():
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(torch.cuda.current_stream(self.builder.device))
            yield out
