"""Real-data input pipeline of the large-range model: KITTI odometry velodyne `.bin` + colour image
`.npy` + `calib.txt` -> the sample dict `RegNet_v2` / `Trainer.step` consume, with everything per point or per pixel done
ON THE DEVICE.

Reference: `src/kitti_odometry_corr_lidarnone_proj.py` — directory layout `make_dataset` (:38-77), calibration parsing
`read_calib` (:205-229), the random mis-calibration `generate_random_transform` / `angles2rotation_matrix` (:80-91,
:386-406), its inverse as ground truth (:608-612, `utils.extrinsic_to_dual_quat` / `rotmat_to_quat` utils.py:245-322),
point shuffle (:532-533), training jitter (:332-343, :622-626), `init_extrinsic @ [pc;1]` (:649-656), zero padding to
150 000 rows (:699-711), image crop of the top 50 rows, x0.5 resize, 160x512 crop and the intrinsic bookkeeping
(:713-747), keys of the sample (:770-787).

nuScenes half (same sample dict, `src/nuscenes_loader_proj_nolidar.py:94-387`): `NuScenesFiles` reads the pickled split
lists the reference reads (`[((lidar file, camera file), K, Tr, night_tag)]`, :132-166), the `.pcd.bin` sweeps (float32
x 5 per point, what the devkit's `LidarPointCloud.from_file` does) and the camera images; `NuScenesSampleBuilder` does
the ego-vehicle / elevation filters (:244-279), the axis swap of `raw_point_xyz` (:337-341), the 100-row crop and the
0.32 x 0.2 resize (:287-296) on the device.  `nuscenes_pair_from_tables` derives one list entry (K, Tr) from the raw
nuScenes JSON tables without the devkit.

What the reference does in numpy / cv2 on DataLoader workers per sample (150 000-row float64 matmuls, concatenations, a
cv2 resize) happens here in a handful of device kernels per BATCH; the host only reads the two files and draws the six
random numbers of the perturbation.  `Prefetcher` overlaps file reading (a thread), the pinned host->device copies (a
copy stream) and the device-side build with the training step of the previous batch.
"""
import math
import os
import random
import atexit
import threading
import weakref
from queue import Full, Queue

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
_COEF_BITS = 11                      # cv2's INTER_RESIZE_COEF_BITS: bilinear weights as 11-bit fixed point


def _linear_taps(n_out, n_in, clamp_weight, device):
    """source index and the two fixed-point weights of every output coordinate along one axis, as cv2 computes them
    (resize.cpp, `resize` linear branch): f = float((d + 0.5) * scale - 0.5) with scale = n_in / n_out in double,
    s = floor(f), f -= s in float32, weights = cvRound((1 - f) * 2048), cvRound(f * 2048) (round half to even).
    Along x (`clamp_weight`) an out-of-range tap is replaced by the border pixel with weight 1; along y the weights stay
    and the ROW indices are clamped."""
    scale = float(n_in) / float(n_out)
    d = torch.arange(n_out, dtype=torch.float64)
    f = ((d + 0.5) * scale - 0.5).to(torch.float32)
    s = torch.floor(f)
    f = f - s
    s = s.to(torch.int64)
    if clamp_weight:
        lo, hi = s < 0, s >= n_in - 1
        f = torch.where(lo | hi, torch.zeros_like(f), f)
        s = torch.where(lo, torch.zeros_like(s), torch.where(hi, torch.full_like(s, n_in - 1), s))
    one = torch.ones_like(f)
    w0 = torch.round((one - f) * float(1 << _COEF_BITS)).to(torch.int64)
    w1 = torch.round(f * float(1 << _COEF_BITS)).to(torch.int64)
    i0 = s.clamp(0, n_in - 1); i1 = (s + 1).clamp(0, n_in - 1)
    return i0.to(device), i1.to(device), w0.to(device), w1.to(device)


_RESIZE_PLANS = {}


def _resize_plan(H, W, oh, ow, device):
    """taps of both axes, the source rows the output touches and the tap rows renumbered within them — built on the host once
    per (sizes, device) and kept on the device: a dataset has a handful of frame sizes, and building the plan per image cost
    eight blocking copies and a `torch.unique` (a sort, and a host read of its size)"""
    key = (H, W, oh, ow, str(device))
    plan = _RESIZE_PLANS.get(key)
    if plan is None:
        x0, x1, a0, a1 = _linear_taps(ow, W, True, "cpu")
        y0, y1, b0, b1 = _linear_taps(oh, H, False, "cpu")
        rows = torch.unique(torch.cat([y0, y1]))
        remap = torch.zeros(H, dtype=torch.int64); remap[rows] = torch.arange(rows.numel())
        plan = tuple(t.to(device) for t in (x0, x1, a0, a1, remap[y0], remap[y1], b0, b1, rows))
        _RESIZE_PLANS[key] = plan
    return plan


# --------------------------------------------------------------------------------------------------------------
# colour jitter of the training images (torchvision.transforms.ColorJitter on a PIL image, as the reference loaders call it)
# --------------------------------------------------------------------------------------------------------------
# nuScenes (nuscenes_loader_proj_nolidar.py:172-187, :308-310): ColorJitter(brightness (0.8, 1.2), contrast (0.8, 1.2), saturation
# (0.8, 1.2), hue (-0.1, 0.1)) on the cropped uint8 image in train mode.  KITTI (kitti_odometry_corr_lidarnone_proj.py:499-514,
# :746-750): `augment_img` builds `transforms.ColorJitter()` WITHOUT ranges — the identity — and discards the result of
# `get_params`, so the reference's default path (crop=False, :256) changes no pixel; only `augment_img_crop` (crop=True) jitters.
# Hence: NuScenesSampleBuilder jitters by default, DeviceSampleBuilder (KITTI) does not (`color_jitter=` overrides both).
#
# The four operations run on the device on the uint8 image with PIL's own integer / float32 arithmetic (libImaging Blend.c,
# Convert.c rgb2hsv / hsv2rgb, ImageEnhance's degenerate images), so that a given (order, factors) draw reproduces what
# torchvision's PIL path returns bit for bit (tests/test_data_pipeline.py compares against PIL where it is importable).
def _cj_blend(deg, img, f):
    """PIL Image.blend(deg, img, f) on uint8: float32 arithmetic, clip to [0, 255], truncate"""
    # (a Python scalar times a float32 tensor is computed in float32; a 0-d device tensor built from it would be a blocking
    #  host-to-device copy per call)
    import numpy as np
    t = deg.float() + float(np.float32(f)) * (img.float() - deg.float())
    return torch.floor(t.clamp(0.0, 255.0)).to(torch.uint8)


def _cj_gray(img):
    """PIL convert("L") of RGB: (R*19595 + G*38470 + B*7471 + 0x8000) >> 16"""
    i = img.to(torch.int64)
    return ((i[..., 0] * 19595 + i[..., 1] * 38470 + i[..., 2] * 7471 + 0x8000) >> 16).to(torch.uint8)


def _cj_hue(img, f):
    """torchvision adjust_hue on a PIL image: RGB -> HSV (uint8, libImaging rgb2hsv_row), H += uint8(f * 255) with wrap-around,
    HSV -> RGB (hsv2rgb_row); float32 / double roundings as in the C code"""
    r, g, b = [img[..., k].float() for k in range(3)]
    maxc = torch.maximum(r, torch.maximum(g, b)); minc = torch.minimum(r, torch.minimum(g, b))
    cr = maxc - minc
    ok = cr > 0
    crs = torch.where(ok, cr, torch.ones_like(cr))
    s = cr / torch.where(ok, maxc, torch.ones_like(maxc))
    rc, gc, bc = (maxc - r) / crs, (maxc - g) / crs, (maxc - b) / crs
    h = torch.where(r == maxc, (bc - gc).double(), torch.where(g == maxc, 2.0 + rc.double() - bc.double(), 4.0 + gc.double() - rc.double())).float()
    h = torch.fmod(h.double() / 6.0 + 1.0, 1.0).float().double()
    zero = torch.zeros_like(h, dtype=torch.int64)
    uh = torch.where(ok, (h * 255.0).to(torch.int64).clamp(0, 255), zero)
    us = torch.where(ok, (s.double() * 255.0).to(torch.int64).clamp(0, 255), zero)
    v = maxc.double()
    uh = (uh + int(f * 255) % 256) % 256
    h6 = uh.double() * 6.0 / 255.0
    fs = us.double() / 255.0
    i = torch.floor(h6)
    fr = h6 - i
    rnd = lambda x: torch.floor(x + 0.5)
    p, q, t = rnd(v * (1.0 - fs)), rnd(v * (1.0 - fs * fr)), rnd(v * (1.0 - fs * (1.0 - fr)))
    i = i.to(torch.int64) % 6
    pick = lambda c0, c1, c2, c3, c4, c5: torch.where(i == 0, c0, torch.where(i == 1, c1, torch.where(i == 2, c2, torch.where(i == 3, c3, torch.where(i == 4, c4, c5)))))
    out = torch.stack([pick(v, q, p, p, t, v), pick(t, v, v, q, p, p), pick(p, p, t, v, v, q)], -1)
    out = torch.where((us == 0).unsqueeze(-1), v.unsqueeze(-1).expand_as(out), out)
    return out.clamp(0, 255).to(torch.uint8)


def color_jitter_u8(img, order, brightness, contrast, saturation, hue):
    """img uint8 [H,W,3] (any device) -> jittered uint8 image.  `order`: permutation of (0 brightness, 1 contrast, 2 saturation,
    3 hue) = torchvision's `fn_idx`; factors as torchvision samples them.  No host synchronisation."""
    for k in order:
        if k == 0:
            img = _cj_blend(torch.zeros_like(img), img, brightness)
        elif k == 1:
            mean = torch.floor(_cj_gray(img).double().mean() + 0.5)          # int(ImageStat.Stat(L).mean[0] + 0.5), stays on the device
            img = _cj_blend(mean.float().expand(img.shape), img, contrast)
        elif k == 2:
            img = _cj_blend(_cj_gray(img).unsqueeze(-1).expand_as(img), img, saturation)
        else:
            img = _cj_hue(img, hue)
    return img


def draw_color_jitter(rng, brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-0.1, 0.1)):
    """one draw of ColorJitter.get_params: (random order of the four operations, four uniform factors)"""
    order = list(range(4))
    rng.shuffle(order)
    return (tuple(order), rng.uniform(*brightness), rng.uniform(*contrast), rng.uniform(*saturation), rng.uniform(*hue))


def resize_linear_u8(img, oh, ow):
    """`cv2.resize(img, (ow, oh), interpolation=cv2.INTER_LINEAR)` of a uint8 image [H,W,C], on any device.

    cv2 (opencv-python 4.x; absent from this image) is restated from its published algorithm, modules/imgproc/src/resize.cpp:
    * an exact 2x shrink on both axes is routed to INTER_AREA's fast path: the 2x2 mean `(a+b+c+d+2) >> 2`;
    * otherwise the 8-bit bilinear path: 11-bit fixed-point weights (`_linear_taps`), a horizontal pass into int32
      `S[x0]*a0 + S[x1]*a1`, then the vertical pass `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`.
    The reference loaders pass dsize = (round(w*s), round(h*s)); for odd sizes (a 1241-wide KITTI frame -> 620) the scale
    is not exactly 2 and the second path runs."""
    H, W, _ = img.shape
    dev = img.device
    v = img.to(torch.int64)
    if H == 2 * oh and W == 2 * ow:
        s = v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2]
        return ((s + 2) >> 2).to(torch.uint8)
    x0, x1, a0, a1, y0r, y1r, b0, b1, rows = _resize_plan(H, W, oh, ow, dev)
    src = v[rows]                                                  # horizontal pass only on the rows the output touches
    hz = src[:, x0] * a0.view(1, -1, 1) + src[:, x1] * a1.view(1, -1, 1)                       # [rows, ow, C]
    s0, s1 = hz[y0r] >> 4, hz[y1r] >> 4
    out = (((b0.view(-1, 1, 1) * s0) >> 16) + ((b1.view(-1, 1, 1) * s1) >> 16) + 2) >> 2
    return out.to(torch.uint8)


def resize_half_u8(img):
    """cv2.resize(img, (round(w/2), round(h/2)), INTER_LINEAR) — the KITTI loader's call (:716-720)"""
    H, W, _ = img.shape
    return resize_linear_u8(img, int(round(H * 0.5)), int(round(W * 0.5)))


def affine_f64(pc, E):
    """Points [N,3] float32 through the host 3x4 float64 matrix `E` (R | t) in float64 arithmetic, float32 result — the
    loaders' `np.dot(E, [pc; 1])` (:654-656).  Written out per coordinate with the twelve entries as scalars: a
    `pc.double() @ E.T` goes to a rocBLAS kernel, and that kernel on the Prefetcher's stream next to the training step's
    resident-grid chain kernels (mlp_chain.hip) made grid barriers time out (tools/chain_cotenancy.py); the scalars
    also save the synchronous copy of `E` to the device."""
    x, y, z = pc[:, 0].double(), pc[:, 1].double(), pc[:, 2].double()
    E = np.asarray(E, dtype=np.float64)
    return torch.stack([x * float(E[r, 0]) + y * float(E[r, 1]) + z * float(E[r, 2]) + float(E[r, 3]) for r in range(3)], 1).float()


class DeviceSampleBuilder:
    """Turns a list of host half-samples (`KittiOdometryFiles`) into ONE batched sample dict on `device`.

    Padding rows stay exactly zero in `lidar`, `raw_point_xyz` and `lidar_feats` (:699-711); `rgb` is float [B,3,160,512]
    in 0..255 (the reference does not normalise, :757-760)."""

    COLOR_JITTER_DEFAULT = False     # the reference's KITTI augment_img is the identity (see color_jitter_u8 above)

    def __init__(self, device, mode="train", sample_point=150000, img_H=160, img_W=512, img_scale=0.5, crop_top=50,
                 rng=None, jitter=True, color_jitter=None, fused=True, perm_seed=None):
        """`jitter`: per-point N(0, 0.01^2) noise of the cloud in train mode; `color_jitter`: ColorJitter of the cropped image
        in train mode (None = the reference loader's effective behaviour: off for KITTI, on for nuScenes).  The random draws
        (perturbation, crop offsets, colour jitter) come from `rng` in this order per sample; the reference draws from the
        global `random` / numpy / torch generators in its own order, so equal seeds do not reproduce its samples — the
        golden vectors pin the arithmetic by passing the reference's draws in (`Pr`, `perm`, `crop`, `jitter_params`).
        `perm_seed`: seed of the host-drawn point shuffles (`draw_perm`: numpy streams keyed by (perm_seed, sample number), NOT
        `rng` and not torch's generator — `torch.manual_seed` alone does not fix them).  None: one draw from
        `rng.getrandbits(62)` when the rng has it (this consumes one value of its stream before the first sample), else from
        its `random()`."""
        self.device, self.mode = torch.device(device), mode
        self.sample_point, self.img_H, self.img_W, self.img_scale, self.crop_top = sample_point, img_H, img_W, img_scale, crop_top
        self.fused = fused                 # HIP device: the batch build in two launches (csrc/loader_build.hip); False = the torch path
        self.rng = rng or random
        if perm_seed is None:
            perm_seed = self.rng.getrandbits(62) if hasattr(self.rng, "getrandbits") else int(self.rng.random() * (1 << 62))
        self.perm_seed = int(perm_seed)
        self.perm_gen = np.random.default_rng(self.perm_seed)
        self._small = [[None, None] for _ in range(12)]     # pinned blocks (+ the event of their last copy) of `_upload`
        self._uploads = 0
        self.jitter = jitter and mode == "train"
        self.color_jitter = (self.COLOR_JITTER_DEFAULT if color_jitter is None else bool(color_jitter)) and mode == "train"

    def draw_perm(self, n, key=None):
        """the point shuffle of a sample (:532-533), drawn on the host: `torch.randperm` on the device is a radix sort whose blocks
        wait on one another, which next to the training step's resident-grid kernels ended in grid-barrier time-outs
        (tools/chain_cotenancy.py); the Prefetcher's reader thread draws it ahead and stages it in pinned memory.  numpy, not
        `torch.randperm`: a multi-threaded torch CPU op called from a thread other than the main one cost 90 ms per call on the
        256-thread GPU box (an OpenMP team per call) and stalled the main thread's launches with it (tools/time_loader.py).
        `key` (the Prefetcher passes the running number of the sample): an independent stream per sample, so that the pool's workers
        can draw in parallel (1.7 ms per 120 000-point scan) and the result does not depend on which worker ran first."""
        if key is not None:
            return np.random.default_rng([self.perm_seed, int(key)]).permutation(n)
        return self.perm_gen.permutation(n)

    def _upload_bytes(self, raw):
        """host bytes (a contiguous numpy array of any dtype) -> device uint8 tensor without blocking the host: through a ring of
        pinned blocks (a pageable source makes the copy synchronous, i.e. the host waits for everything queued on the stream)"""
        raw = np.ascontiguousarray(raw).view(np.uint8).ravel()
        if self.device.type != "cuda":
            return torch.from_numpy(raw.copy())
        slot = self._small[self._uploads % len(self._small)]
        self._uploads += 1
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < raw.size:
            slot[0] = torch.empty(max(4096, raw.size), dtype=torch.uint8, pin_memory=True)
        slot[0].numpy()[:raw.size] = raw
        out = slot[0][:raw.size].to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event(); slot[1].record()
        return out

    def _upload(self, values):
        """float32 [n] host values -> float32 device tensor (see _upload_bytes)"""
        values = np.ascontiguousarray(values, dtype=np.float32).ravel()
        return self._upload_bytes(values).view(torch.float32)

    def _upload_blocks(self, arrays):
        """several small host arrays -> float32 device tensors of the same shapes through ONE copy (each starts 16-byte aligned)"""
        sizes = [(a.size + 3) // 4 * 4 for a in arrays]
        flat = np.zeros(sum(sizes), dtype=np.float32)
        offs = np.cumsum([0] + sizes)
        for a, o in zip(arrays, offs):
            flat[o:o + a.size] = np.ravel(a)                 # float64 -> float32 here, as `torch.as_tensor(a, dtype=float32)` rounds
        dev_flat = self._upload(flat)
        return [dev_flat[o:o + a.size].view(a.shape) for a, o in zip(arrays, offs)]

    def _crop_rgb(self, img, dx, dy, host):
        """crop -> (train mode) colour jitter on the uint8 crop -> float [3,H,W] in 0..255"""
        crop = img[dy:dy + self.img_H, dx:dx + self.img_W]
        if self.color_jitter:
            params = host.get("jitter_params") or draw_color_jitter(self.rng)
            crop = color_jitter_u8(crop, *params)
        return crop.permute(2, 0, 1).float()

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
        """No step of this blocks the host when scan / image / perm arrive in pinned memory or on the device (the Prefetcher): the
        per-sample matrices enter the kernels as scalars, the small outputs go up in one pinned block."""
        dev = self.device
        B = len(hosts)
        SP = self.sample_point
        if self._fused_ok(hosts):
            return self._call_fused(hosts)
        lidar = torch.zeros(B, SP, 3, device=dev)
        raw = torch.zeros(B, SP, 3, device=dev)
        feats = torch.zeros(B, SP, 1, device=dev)
        rgb = torch.empty(B, 3, self.img_H, self.img_W, device=dev)
        Es, qs, ts, Ks, paths, idxs = [], [], [], [], [], []
        for b, host in enumerate(hosts):
            Pr, E, q, t = self.perturbation(host)
            scan = host["scan"].to(dev, non_blocking=True)                            # [N,4]
            n = min(scan.shape[0], SP)
            perm = host.get("perm")
            perm = torch.as_tensor(self.draw_perm(scan.shape[0]) if perm is None else perm).to(dev, non_blocking=True)   # :532-533
            scan = scan[perm][:n]
            pc = scan[:, :3]
            if self.jitter:                                                           # :332-343: N(0, 0.01^2) clipped at 5 cm
                pc = pc + torch.clamp(0.01 * torch.randn_like(pc), -0.05, 0.05)
            lidar[b, :n] = affine_f64(pc, E); raw[b, :n] = pc; feats[b, :n, 0] = scan[:, 3]   # :654-656
            # image: drop the top rows, halve, crop (:713-747)
            img = host["image"].to(dev, non_blocking=True)[self.crop_top:]
            K = np.copy(host["K"]).astype(np.float64); K[1, 2] -= self.crop_top
            h0, w0, _ = img.shape
            img = resize_linear_u8(img, int(round(h0 * self.img_scale)), int(round(w0 * self.img_scale)))
            K[0, 0] *= self.img_scale; K[0, 2] *= self.img_scale; K[1, 1] *= self.img_scale; K[1, 2] *= self.img_scale
            h, w, _ = img.shape
            if self.mode == "train":
                dx, dy = self.rng.randint(0, w - self.img_W), self.rng.randint(0, h - self.img_H)
            else:
                dx, dy = int((w - self.img_W) / 2), int((h - self.img_H) / 2)
            dx, dy = host.get("crop", (dx, dy))
            rgb[b] = self._crop_rgb(img, dx, dy, host)
            K[0, 2] -= dx; K[1, 2] -= dy
            Es.append(E); qs.append(q); ts.append(t); Ks.append(K)
            paths.append(host["path_info"]); idxs.append(host["index"])
        ext, q_gt, t_gt, intr = self._upload_blocks([np.stack(Es), np.stack(qs), np.stack(ts), np.stack(Ks)])
        return {"rgb": rgb, "lidar": lidar, "raw_point_xyz": raw, "lidar_feats": feats, "init_extrinsic": ext, "init_intrinsic": intr,
                "decalib_real_gt": q_gt, "decalib_dual_gt": t_gt, "path_info": paths, "index": idxs,
                "resize_img": torch.tensor([[self.img_scale, self.img_scale]] * B)}


    # ---- the same build in two launches per batch (csrc/loader_build.hip) ---------------------------------------------------
    def _fused_ok(self, hosts):
        """HIP device, KITTI builder proper (the nuScenes subclass has data-dependent filters), no colour jitter: the batch goes
        through i2p_kitti_points_build / i2p_kitti_image_build — the arithmetic of the torch path below, operation for operation
        (bit-identical: tests/test_data_pipeline.py), without its ~50 launches per sample"""
        return (self.fused and self.device.type == "cuda" and type(self) is DeviceSampleBuilder and not self.color_jitter
                and all(h["scan"].dtype == torch.float32 and h["image"].dtype == torch.uint8 for h in hosts))

    @torch.no_grad()
    def _call_fused(self, hosts):
        from . import ops
        be = ops.hip_backend()                                   # raises without the device library
        dev, B, SP = self.device, len(hosts), self.sample_point
        ptab = np.zeros((B, 16), dtype=np.int64)                 # i2p_kitti_points_build's table rows (include/i2p_ops.h)
        itab = np.zeros((B, 8), dtype=np.int64)
        keep, Es, qs, ts, Ks, paths, idxs = [], [], [], [], [], [], []
        for b, host in enumerate(hosts):
            Pr, E, q, t = self.perturbation(host)
            scan = host["scan"].to(dev, non_blocking=True).contiguous()
            perm = host.get("perm")
            perm = torch.as_tensor(self.draw_perm(scan.shape[0]) if perm is None else perm).to(dev, non_blocking=True).contiguous()
            if perm.dtype != torch.int64:
                perm = perm.long()
            ptab[b, 0], ptab[b, 1], ptab[b, 2] = scan.data_ptr(), perm.data_ptr(), min(scan.shape[0], SP)
            ptab[b, 4:] = np.ascontiguousarray(E, dtype=np.float64).ravel().view(np.int64)
            img = host["image"].to(dev, non_blocking=True).contiguous()
            H0, W0, _ = img.shape
            h0 = H0 - self.crop_top
            K = np.copy(host["K"]).astype(np.float64); K[1, 2] -= self.crop_top
            oh, ow = int(round(h0 * self.img_scale)), int(round(W0 * self.img_scale))
            K[0, 0] *= self.img_scale; K[0, 2] *= self.img_scale; K[1, 1] *= self.img_scale; K[1, 2] *= self.img_scale
            if self.mode == "train":
                dx, dy = self.rng.randint(0, ow - self.img_W), self.rng.randint(0, oh - self.img_H)
            else:
                dx, dy = int((ow - self.img_W) / 2), int((oh - self.img_H) / 2)
            dx, dy = host.get("crop", (dx, dy))
            if not (0 <= dx <= ow - self.img_W and 0 <= dy <= oh - self.img_H):
                raise ValueError("crop (%d, %d) of a %dx%d window leaves the %dx%d resized image" % (dx, dy, self.img_W, self.img_H, ow, oh))
            itab[b] = (img.data_ptr() + self.crop_top * W0 * 3, h0, W0, oh, ow, dx, dy, int(h0 == 2 * oh and W0 == 2 * ow))
            K[0, 2] -= dx; K[1, 2] -= dy
            keep += [scan, perm, img]
            Es.append(E); qs.append(q); ts.append(t); Ks.append(K)
            paths.append(host["path_info"]); idxs.append(host["index"])
        tables = self._upload_bytes(np.concatenate([ptab.ravel(), itab.ravel()]))
        noise = torch.randn(B, SP, 3, device=dev) if self.jitter else None         # :332-343: one draw per batch
        lidar, raw, feats = be.kitti_points_build(tables[:B * 128], B, SP, noise)
        rgb = be.kitti_image_build(tables[B * 128:], B, self.img_H, self.img_W)
        ext, q_gt, t_gt, intr = self._upload_blocks([np.stack(Es), np.stack(qs), np.stack(ts), np.stack(Ks)])
        del keep                                                 # (the kernels are queued on this stream: the allocator orders the reuse)
        return {"rgb": rgb, "lidar": lidar, "raw_point_xyz": raw, "lidar_feats": feats, "init_extrinsic": ext, "init_intrinsic": intr,
                "decalib_real_gt": q_gt, "decalib_dual_gt": t_gt, "path_info": paths, "index": idxs,
                "resize_img": torch.tensor([[self.img_scale, self.img_scale]] * B)}


# --------------------------------------------------------------------------------------------------------------
# nuScenes (src/nuscenes_loader_proj_nolidar.py)
# --------------------------------------------------------------------------------------------------------------
def read_pcd_bin(path):
    """nuScenes LIDAR_TOP sweep `.pcd.bin`: float32 (x, y, z, intensity, ring) per point -> [N,4] (the devkit's
    `LidarPointCloud.from_file` keeps the first four of the five columns; nuscenes_loader_proj_nolidar.py:237)"""
    return np.fromfile(path, dtype=np.float32).reshape(-1, 5)[:, :4]


def _quat_wxyz_to_matrix(q):
    w, x, y, z = [float(v) for v in q]
    n = math.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def _pose(rotation_wxyz, translation):
    P = np.identity(4)
    P[:3, :3] = _quat_wxyz_to_matrix(rotation_wxyz); P[:3, 3] = np.asarray(translation, dtype=np.float64)
    return P


def nuscenes_pair_from_tables(lidar_sd, cam_sd, calibrated_sensor, ego_pose, night_tag=False):
    """One entry `((lidar file, camera file), K, Tr, night_tag)` of the reference's split lists from the raw nuScenes
    JSON tables (no devkit): `lidar_sd` / `cam_sd` = rows of sample_data.json, `calibrated_sensor` / `ego_pose` = dicts
    token -> row.  Tr [4,4] maps LIDAR_TOP sensor coordinates to the camera frame through the two ego poses
    (sensor -> ego(t_lidar) -> global -> ego(t_cam) -> camera), K = the camera's `camera_intrinsic`."""
    cs_l, cs_c = calibrated_sensor[lidar_sd["calibrated_sensor_token"]], calibrated_sensor[cam_sd["calibrated_sensor_token"]]
    ep_l, ep_c = ego_pose[lidar_sd["ego_pose_token"]], ego_pose[cam_sd["ego_pose_token"]]
    lidar_to_ego = _pose(cs_l["rotation"], cs_l["translation"]); ego_l_to_global = _pose(ep_l["rotation"], ep_l["translation"])
    cam_to_ego = _pose(cs_c["rotation"], cs_c["translation"]); ego_c_to_global = _pose(ep_c["rotation"], ep_c["translation"])
    Tr = np.linalg.inv(cam_to_ego) @ np.linalg.inv(ego_c_to_global) @ ego_l_to_global @ lidar_to_ego
    K = np.asarray(cs_c["camera_intrinsic"], dtype=np.float64)
    return ((lidar_sd["filename"], cam_sd["filename"]), K, Tr, bool(night_tag))


class NuScenesFiles(torch.utils.data.Dataset):
    """Host half of a nuScenes sample: the sweep [N,4] f32, the camera image [900,1600,3] u8, K and Tr of the list entry.
    Split lists and sub-directories as in the reference (:130-166): `train` = train + val lists under `<root>/trainval`,
    `val` under `<root>/trainval`, `test` under `<root>/test`; the lists are looked up in `split_dir` (the reference
    hard-codes ./nuScenes_datasplit relative to the working directory)."""

    def __init__(self, root_path, mode="train", split_dir="./nuScenes_datasplit", random_car=True, skip=1):
        import pickle
        info = "randominfo" if random_car else "info"
        lst = lambda m: os.path.join(split_dir, f"{m}_dataset_{info}_proj_day.list")
        if mode not in ("train", "val", "test"):
            raise NotImplementedError(mode)
        self.mode, self.skip = mode, skip
        self.root = os.path.join(root_path, "test" if mode == "test" else "trainval")
        self.items = []
        for m in {"train": ("train", "val"), "val": ("val",), "test": ("test",)}[mode]:
            with open(lst(m), "rb") as f:
                self.items.extend(pickle.load(f))

    def __len__(self):
        return int(np.ceil(len(self.items) / self.skip))

    def __getitem__(self, index):
        from PIL import Image
        (lp, cp), K, Tr, _night = self.items[index * self.skip]
        scan = read_pcd_bin(os.path.join(self.root, lp))
        img = np.array(Image.open(os.path.join(self.root, cp)), np.uint8)
        return {"scan": torch.from_numpy(np.ascontiguousarray(scan)), "image": torch.from_numpy(np.ascontiguousarray(img)),
                "K": np.asarray(K, dtype=np.float64).copy(), "Tr": np.asarray(Tr, dtype=np.float64), "index": index,
                "path_info": "%d" % (index * self.skip)}


class NuScenesSampleBuilder(DeviceSampleBuilder):
    """nuScenes half-samples -> one batched sample dict on the device (nuscenes_loader_proj_nolidar.py:230-387): point
    shuffle, ego-vehicle box and elevation filters, training jitter, `init_extrinsic @ [pc;1]`, `raw_point_xyz` =
    (y, -x, z) of the sensor-frame points, zero padding to 150 000 rows; image: top 100 rows dropped, resized by
    (0.32, 0.2) with cv2's bilinear rule, 160x512 crop, intrinsics adjusted alike."""
    TAN_UP, TAN_DOWN = 0.03492076949, -0.4620648698           # +2 deg / -24.8 deg elevation window (:271-275)
    COLOR_JITTER_DEFAULT = True      # nuscenes_loader_proj_nolidar.py:172-187, :308-310

    def __init__(self, device, mode="train", sample_point=150000, img_H=160, img_W=512, img_scale_H=0.2, img_scale_W=0.32,
                 crop_top=100, rng=None, jitter=True, color_jitter=None):
        super().__init__(device, mode, sample_point, img_H, img_W, img_scale_H, crop_top, rng, jitter, color_jitter)
        self.img_scale_H, self.img_scale_W = img_scale_H, img_scale_W

    def perturbation(self, host):
        Pr = host.get("Pr")
        if Pr is None:
            Pr = random_transform(self.rng).astype(np.float32)               # (:230: float32 perturbation)
        Pr = np.asarray(Pr)
        calib_extrinsic = np.linalg.inv(Pr)[:3, :]
        q = rotmat_to_quat(calib_extrinsic[:3, :3])
        t = calib_extrinsic[:, 3]
        init_extrinsic = np.dot(Pr, host["Tr"])[:3]
        return Pr, init_extrinsic, q, t

    @torch.no_grad()
    def __call__(self, hosts, stream=None):
        dev = self.device
        B, SP = len(hosts), self.sample_point
        lidar = torch.zeros(B, SP, 3, device=dev); raw = torch.zeros(B, SP, 3, device=dev); feats = torch.zeros(B, SP, 1, device=dev)
        rgb = torch.empty(B, 3, self.img_H, self.img_W, device=dev)
        ext = torch.empty(B, 3, 4, device=dev); q_gt = torch.empty(B, 4, device=dev); t_gt = torch.empty(B, 3, device=dev)
        Ks, raw_Ks, paths, idxs, stats, counts = [], [], [], [], [], []
        for b, host in enumerate(hosts):
            Pr, E, q, t = self.perturbation(host)
            scan = host["scan"].to(dev, non_blocking=True)                            # [N,4]
            perm = host.get("perm")
            perm = torch.as_tensor(self.draw_perm(scan.shape[0]) if perm is None else perm).to(dev, non_blocking=True)   # :238
            scan = scan[perm]
            x, y = scan[:, 0], scan[:, 1]
            inside = (x < 0.8) & (x > -0.8) & (y < 2.7) & (y > -2.7)                   # the ego vehicle, :243-247
            scan = scan[~inside]
            # 5 / 95 percentiles of (y, -x, z) after the ego filter (`pc_stat`, :253-259; diagnostic only)
            pq = torch.tensor([0.05, 0.95], dtype=torch.float64, device=dev)
            stats.append(torch.stack([torch.quantile(c.double(), pq) for c in (scan[:, 1], -scan[:, 0], scan[:, 2])]))
            ratio = scan[:, 2] / torch.sqrt(scan[:, 0] * scan[:, 0] + scan[:, 1] * scan[:, 1])
            scan = scan[(ratio < self.TAN_UP) & (ratio > self.TAN_DOWN)]                # :266-278
            n = min(scan.shape[0], SP)
            scan = scan[:n]
            pc = scan[:, :3]
            if self.jitter:
                pc = pc + torch.clamp(0.01 * torch.randn_like(pc), -0.05, 0.05)
            lidar[b, :n] = affine_f64(pc, E)                                            # :343-348
            raw[b, :n, 0] = pc[:, 1]; raw[b, :n, 1] = -pc[:, 0]; raw[b, :n, 2] = pc[:, 2]   # :337-341
            feats[b, :n, 0] = scan[:, 3]
            counts.append(n)
            img = host["image"].to(dev, non_blocking=True)[self.crop_top:]
            K = np.copy(host["K"]).astype(np.float64); raw_Ks.append(np.copy(K)); K[1, 2] -= self.crop_top
            h0, w0, _ = img.shape
            img = resize_linear_u8(img, int(round(h0 * self.img_scale_H)), int(round(w0 * self.img_scale_W)))
            K[0, 0] *= self.img_scale_W; K[0, 2] *= self.img_scale_W; K[1, 1] *= self.img_scale_H; K[1, 2] *= self.img_scale_H
            h, w, _ = img.shape
            if self.mode == "train":
                dx, dy = self.rng.randint(0, w - self.img_W), self.rng.randint(0, h - self.img_H)
            else:
                dx, dy = int((w - self.img_W) / 2), int((h - self.img_H) / 2)
            dx, dy = host.get("crop", (dx, dy))
            rgb[b] = self._crop_rgb(img, dx, dy, host)
            K[0, 2] -= dx; K[1, 2] -= dy
            Ks.append(K)
            ext[b] = torch.as_tensor(E, dtype=torch.float32); q_gt[b] = torch.as_tensor(q, dtype=torch.float32)
            t_gt[b] = torch.as_tensor(t, dtype=torch.float32)
            paths.append(host["path_info"]); idxs.append(host["index"])
        intr = torch.as_tensor(np.stack(Ks), dtype=torch.float32).to(dev)
        return {"rgb": rgb, "lidar": lidar, "raw_point_xyz": raw, "lidar_feats": feats, "init_extrinsic": ext, "init_intrinsic": intr,
                "raw_intrinsic": torch.as_tensor(np.stack(raw_Ks), dtype=torch.float32).to(dev),
                "decalib_real_gt": q_gt, "decalib_dual_gt": t_gt, "path_info": paths, "index": idxs, "n_points": counts,
                "pc_stat": torch.stack(stats), "resize_img": torch.tensor([[self.img_scale_H, self.img_scale_W]] * B)}


_LIVE = weakref.WeakSet()        # Prefetchers with a running reader thread


@atexit.register
def _close_prefetchers():
    # a daemon thread still inside the interpreter at finalisation is torn down in C++ frames ("terminate called without an
    # active exception"): stop the readers first
    for p in list(_LIVE):
        p.close()


class Prefetcher:
    """Iterates device sample dicts: a reader thread pulls half-samples from `dataset` (file reads on a small pool, copies into
    pinned staging), the device build of batch i+1 runs on a side stream under the step of batch i; `__next__` makes the current
    stream wait for that side stream, no host synchronisation.

    The reader thread and its pool live as long as the Prefetcher (one `__iter__` = one epoch request to them): a thread's first
    multi-threaded torch op creates an OpenMP team of its own, 60 - 190 ms on the 256-thread GPU box, and a reader started per
    epoch paid that at every epoch start (tools/time_loader.py).

    mode "copy" (default): the side stream carries only the host-to-device copies of scan, image and point shuffle (DMA engines);
    the build's kernels are issued in `__next__` on the consumer's stream, in order with the training steps.  mode "side": the
    whole build on the side stream, concurrent with the step.  That is NOT safe next to the step's chain kernels
    (mlp_chain.hip): their grid barrier needs every block resident, and a side-stream kernel whose own blocks wait on one
    another (rocPRIM's sort / select / scan look-back, rocBLAS GEMMs) can hold the slot the last chain block needs while it
    waits itself — the barrier then times out (ChainBarrierTimeout; measured in tools/chain_cotenancy.py,
    profiles/r05_chain_cotenancy.txt).  Use it only with I2P_NO_CHAIN=1 or a build made of element-wise kernels."""

    STAGED = ("scan", "image", "perm")

    def __init__(self, dataset, builder, batch_size, indices=None, depth=2, workers=8, mode="copy"):
        self.dataset, self.builder, self.batch_size, self.mode = dataset, builder, batch_size, mode
        self.indices = list(range(len(dataset))) if indices is None else list(indices)
        self.depth, self.workers = depth, max(1, int(workers))
        self.cuda = builder.device.type == "cuda"
        self.stream = torch.cuda.Stream(builder.device) if self.cuda else None
        # pinned staging, allocated ONCE per (slot, sample, tensor) and re-used: `tensor.pin_memory()` per sample allocates page-locked
        # memory every time — measured 8.7 ms per sample on the MI355X box, 70 ms per batch of 8 against a 5 ms step.  depth + 3
        # slots; a slot is written again only after the device work that read it has finished (its event), so an H2D copy never
        # reads a buffer the reader thread is refilling.
        self._slots = [{"bufs": {}, "event": None} for _ in range(depth + 3)] if self.cuda else []
        self._batches = 0                      # batches staged so far (slot rotation goes on across epochs)
        self._requests, self._thread, self._pool = None, None, None
        self._active = None                    # (queue, cancel event) of the epoch request the reader is serving
        self.trace = None                      # a list here collects (batch, start, slot wait ms, load ms) from the reader

    def _stage(self, slot, j, key, t):
        buf = slot["bufs"].get((j, key))
        if buf is None or buf.numel() < t.numel() or buf.dtype != t.dtype:
            buf = torch.empty(int(t.numel() * 1.25) + 16, dtype=t.dtype, pin_memory=True)
            slot["bufs"][(j, key)] = buf
        view = buf[:t.numel()].view(t.shape)
        np.copyto(view.numpy(), t.numpy())     # plain memcpy (a torch copy_ of this size would go through the thread's OpenMP team)
        return view

    def _epoch(self, q, cancel, epochs=1):
        """reader thread: `epochs` passes (None: endless) over `indices` into `q`, ended by None"""
        import time

        def put(item):
            while not cancel.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except Full:
                    pass
            return False
        try:
            import itertools
            starts = ((e, s) for e in (itertools.count() if epochs is None else range(epochs))
                      for s in range(0, len(self.indices) - self.batch_size + 1, self.batch_size))
            for e, s in starts:
                if cancel.is_set():
                    break
                idx = self.indices
                t0 = time.perf_counter()
                slot = None
                if self.cuda:
                    slot = self._slots[self._batches % len(self._slots)]
                    if slot["event"] is not None:
                        slot["event"].synchronize()
                        slot["event"] = None
                t1 = time.perf_counter()
                serial0 = self._batches * self.batch_size
                self._batches += 1

                def load(jk, slot=slot, serial0=serial0):
                    # one sample on a pool worker: file reads, the point shuffle, the copies into the slot's pinned staging
                    # (numpy releases the GIL in all three)
                    j, k = jk
                    h = self.dataset[k]
                    if self.cuda:
                        if h.get("perm") is None and hasattr(self.builder, "draw_perm"):
                            h["perm"] = self.builder.draw_perm(h["scan"].shape[0], key=serial0 + j)
                        for key in self.STAGED:
                            if h.get(key) is not None:
                                h[key] = self._stage(slot, j, key, torch.as_tensor(h[key]))
                    return h
                hosts = list(self._pool.map(load, enumerate(idx[s:s + self.batch_size])))
                if self.trace is not None:
                    self.trace.append((s // self.batch_size, t0, 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)))
                if not put((hosts, slot)):
                    break
        except BaseException as e:             # surfaces in the consumer
            put(e)
        finally:
            put(None)

    def _serve(self, requests, pool):
        # the reader owns `pool` for its lifetime and shuts it down itself: close() never pulls it from under a running map()
        try:
            while True:
                req = requests.get()
                if req is None:
                    return
                self._epoch(*req)
        finally:
            pool.shutdown(wait=False)

    def close(self):
        """stop the reader thread (also done when the Prefetcher is collected): the epoch being served is cancelled first — an
        endless `cycle(None)` or a generator nobody closed would otherwise never look at the request queue — then the reader is
        asked to leave and joined; it shuts its pool down itself on the way out"""
        th = self._thread
        if th is not None:
            active = self._active
            if active is not None:
                active[1].set()
            self._requests.put(None)
            th.join(timeout=5)
            if th.is_alive():                  # still inside a file read: it leaves at the next cancel check; nothing is pulled away under it
                th.join(timeout=30)
            self._thread = self._pool = self._requests = self._active = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __iter__(self):
        return self._stream(1)

    def cycle(self, epochs=None):
        """batches of `epochs` passes (None: endless) as ONE stream: the reader runs on across the epoch boundary, so the first
        batch of the next pass is ready when the last one of this pass is consumed (an `__iter__` per epoch starts with an empty
        pipeline: two batch reads before the first yield)"""
        return self._stream(epochs)

    def _stream(self, epochs):
        from concurrent.futures import ThreadPoolExecutor
        if self._thread is None:
            self._requests, self._pool = Queue(), ThreadPoolExecutor(max_workers=self.workers)
            self._thread = threading.Thread(target=self._serve, args=(self._requests, self._pool), daemon=True)
            self._thread.start()
            _LIVE.add(self)
        q, cancel = Queue(maxsize=self.depth), threading.Event()
        self._active = (q, cancel)
        self._requests.put((q, cancel, epochs))
        dev = self.builder.device

        def take():
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            return item

        def build(item):
            hosts, slot = item
            if not self.cuda:
                return self.builder(hosts), None
            if self.mode == "side":
                with torch.cuda.stream(self.stream):
                    out = self.builder(hosts)
                    ev = torch.cuda.Event(); ev.record(self.stream)
                slot["event"] = ev                                           # the staging slot is free once this build has run
                return out, ev
            with torch.cuda.stream(self.stream):
                for h in hosts:
                    for key in self.STAGED:
                        if h.get(key) is not None:
                            h[key] = h[key].to(dev, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(self.stream)
            slot["event"] = ev
            return hosts, ev
        try:
            item = take()
            pending = build(item) if item is not None else None
            while pending is not None:
                out, ev = pending
                item = take()
                pending = build(item) if item is not None else None         # next batch's copies / kernels start now
                if ev is not None:
                    cur = torch.cuda.current_stream(dev)
                    cur.wait_event(ev)
                    if self.mode == "side":
                        for v in out.values():
                            if isinstance(v, torch.Tensor) and v.is_cuda:
                                v.record_stream(cur)
                    else:
                        for h in out:
                            for key in self.STAGED:
                                if h.get(key) is not None:
                                    h[key].record_stream(cur)
                        out = self.builder(out)
                yield out
        finally:
            cancel.set()                       # an epoch left early: the reader drops what it has and takes the next request
