"""Evaluator and registration metrics — mirror of the reference's `metric.py` helpers and of the
loop in `evaluation_proj.py:225-405`, restated for the device.

The reference pulls every batch's prediction to the host, runs numpy / scipy per batch and brackets the model
call with two `torch.cuda.synchronize()` (`evaluation_proj.py:238-263`).  Here the forward is one captured
hipGraph on static input buffers, the per-batch latency comes from HIP events, all pose algebra runs on the
device in float64, and nothing is synchronised until the sequence ends: one `.cpu()` per metric vector.

Functions keep the reference's names and argument meaning ([B,3,4] extrinsics, (w,x,y,z) quaternions, degrees).
"""
import math
import time

import numpy as np
import torch

_F64 = torch.float64


# --------------------------------------------------------------------------------------------------
# pose algebra (metric.py:9-104)
# --------------------------------------------------------------------------------------------------
def quat_to_rotmat_batch(q):
    """[B,4] (w,x,y,z) -> [B,3,3]; the polynomial form of metric.py:9-34 (NOT normalising: an unnormalised
    quaternion gives a scaled non-orthogonal matrix there too)."""
    q = q.to(_F64)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w,
                        2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w,
                        2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y], -1).reshape(-1, 3, 3)


def mult_extrinsic_batch(m1, m2):
    """[B,3,4] o [B,3,4] as homogeneous 4x4 products (metric.py:37-50) without building the padded matrices."""
    m1, m2 = m1.to(_F64), m2.to(_F64)
    R1, t1 = m1[:, :, :3], m1[:, :, 3:]
    return torch.cat([R1 @ m2[:, :, :3], R1 @ m2[:, :, 3:] + t1], -1)


def inv_extrinsic(m):
    """Inverse of [R|t; 0 1] (metric.py:53-57 uses a general 4x4 inverse; R need not be orthogonal, so R^-1 is a
    true inverse here too, not a transpose)."""
    m = m.to(_F64)
    Ri = torch.linalg.inv(m[:, :, :3])
    return torch.cat([Ri, -(Ri @ m[:, :, 3:])], -1)


def rotmat_to_euler(rotmat, out="rad"):
    """roll/pitch/yaw of metric.py:60-86 (ZYX convention with the sy < 1e-6 singular branch)."""
    R = rotmat.to(_F64)
    sy = torch.sqrt(R[:, 0, 0] ** 2 + R[:, 1, 0] ** 2)
    sing = sy < 1e-6
    roll = torch.where(sing, torch.atan2(-R[:, 1, 2], R[:, 1, 1]), torch.atan2(R[:, 2, 1], R[:, 2, 2]))
    pitch = torch.atan2(-R[:, 2, 0], sy)
    yaw = torch.where(sing, torch.zeros_like(sy), torch.atan2(R[:, 1, 0], R[:, 0, 0]))
    e = torch.stack([roll, pitch, yaw], -1)
    return e * (180.0 / math.pi) if out == "deg" else e


def _orthonormalise(R):
    """scipy's Rotation.from_matrix projects onto SO(3) (polar factor via SVD) before anything else."""
    U, _, Vt = torch.linalg.svd(R)
    d = torch.sign(torch.linalg.det(U @ Vt))
    D = torch.diag_embed(torch.stack([torch.ones_like(d), torch.ones_like(d), d], -1))
    return U @ D @ Vt


def euler_xzy_deg(R):
    """`Rotation.from_matrix(R).as_euler('xzy', degrees=True)` (metric.py:142-143,234-235): extrinsic x, z, y,
    i.e. R = Ry(c) Rz(b) Rx(a) -> (a, b, c).  Gimbal lock (|R10| = 1) follows scipy: third angle 0."""
    R = _orthonormalise(R.to(_F64))
    b = torch.asin(R[:, 1, 0].clamp(-1.0, 1.0))
    lock = (1.0 - R[:, 1, 0].abs()) < 1e-14
    a = torch.where(lock, torch.atan2(R[:, 2, 1], R[:, 2, 2]), torch.atan2(-R[:, 1, 2], R[:, 1, 1]))
    c = torch.where(lock, torch.zeros_like(b), torch.atan2(-R[:, 2, 0], R[:, 0, 0]))
    return torch.stack([a, b, c], -1) * (180.0 / math.pi)


def calibration_error_batch(e1, e2):
    """|roll|,|pitch|,|yaw| (deg) and |x|,|y|,|z| of inv(e1) o e2 (metric.py:89-102)."""
    err = mult_extrinsic_batch(inv_extrinsic(e1), e2)
    eu = rotmat_to_euler(err[:, :, :3], out="deg").abs()
    tr = err[:, :, 3].abs()
    return eu[:, 0], eu[:, 1], eu[:, 2], tr[:, 0], tr[:, 1], tr[:, 2]


def rre_rte(pred_extrinsic, gt_extrinsic):
    """Per-sample RRE (sum of |xzy Euler angles|, deg) and RTE (|t|) of inv(pred) o gt (metric.py:229-235)."""
    P = mult_extrinsic_batch(inv_extrinsic(pred_extrinsic), gt_extrinsic)
    return euler_xzy_deg(P[:, :, :3]).abs().sum(-1), torch.linalg.norm(P[:, :, 3], dim=-1)


def _decalib(q, t):
    return torch.cat([quat_to_rotmat_batch(q), t.to(_F64).reshape(-1, 3, 1)], -1)


def getExtrinsic(out3, data_valid, out_raw=False):
    """metric.py:105-126: prediction / ground-truth decalibration [R(q)|t] composed onto `init_extrinsic`."""
    dev = out3.device
    pred_raw = _decalib(out3[:, :4], out3[:, 4:])
    gt_raw = _decalib(data_valid["decalib_real_gt"].to(dev), data_valid["decalib_dual_gt"].to(dev))
    init = data_valid["init_extrinsic"].to(dev)[:, :3, :]            # [B,3,4] or homogeneous [B,4,4]
    pred, gt = mult_extrinsic_batch(pred_raw, init), mult_extrinsic_batch(gt_raw, init)
    return (pred, gt, pred_raw, gt_raw) if out_raw else (pred, gt)


def cal_rete_once(out3, data_valid):
    """Batch-mean (RRE, RTE) of the raw decalibration (metric.py:128-147); device scalars, no sync."""
    dev = out3.device
    r, t = rre_rte(_decalib(out3[:, :4], out3[:, 4:]),
                   _decalib(data_valid["decalib_real_gt"].to(dev), data_valid["decalib_dual_gt"].to(dev)))
    return r.mean(), t.mean()


class RteRreEval:
    """metric.py:205-274.  addBatch keeps device vectors; the host sees them once, in evalSeq / save_metric."""

    def __init__(self, threshold=False, rre_th=10.0, rte_th=5.0):
        self.threshold, self.rre_th, self.rte_th = threshold, rre_th, rte_th
        self._r, self._t = [], []

    def reset(self):
        self._r.clear(); self._t.clear()

    def addBatch(self, pred_extrinsic, gt_extrinsic):
        r, t = rre_rte(pred_extrinsic, gt_extrinsic)
        self._r.append(r); self._t.append(t)
        return r, t

    def _all(self):
        if not self._r:
            return np.zeros(0), np.zeros(0)
        return torch.cat(self._r).cpu().numpy(), torch.cat(self._t).cpu().numpy()

    def _kept(self):
        r, t = self._all()
        if self.threshold:
            m = np.logical_and(t < self.rte_th, r < self.rre_th)
            return r[m], t[m], len(r)
        return r, t, len(r)

    def get_recall(self):
        r, _, n = self._kept()
        return len(r) / n

    def evalSeq(self):
        r, t, _ = self._kept()
        return t.mean(), math.sqrt(np.var(t)), r.mean(), math.sqrt(np.var(r))

    def save_metric(self, path):
        r, t = self._all()
        np.savez(path, RRE=r, RTE=t)


# --------------------------------------------------------------------------------------------------
# checkpoints (evaluation_proj.py:134-139)
# --------------------------------------------------------------------------------------------------
def load_checkpoint(model, path, map_location="cpu"):
    """Load a reference checkpoint: `{"model_state_dict": ...}` (evaluation_proj.py:134-139), with or without the
    DistributedDataParallel `module.` prefix the training script saves under.  Strict: the parameter names of
    `i2pnet_amd.model.RegNet_v2` ARE the reference's."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt.get("model_state_dict", ckpt)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=True)
    return ckpt


# --------------------------------------------------------------------------------------------------
# the evaluation loop (evaluation_proj.py:225-405)
# --------------------------------------------------------------------------------------------------
_INPUT_KEYS = ("rgb", "lidar", "raw_point_xyz", "init_intrinsic", "lidar_feats")


class Evaluator:
    """Runs `model` (eval mode) over an iterable of sample dicts (keys of the reference loader:
    rgb, lidar, raw_point_xyz, init_intrinsic, lidar_feats, init_extrinsic, decalib_real_gt, decalib_dual_gt;
    `kitti_odometry_corr_lidarnone_proj.py:757-789`) and returns / writes the reference's metrics.

    The forward is captured once into a hipGraph reading static input buffers; every batch is an async copy into
    those buffers + a replay between two HIP events.  A batch of a different size (the last one) runs eagerly."""

    def __init__(self, model, cfg, device, use_graph=True, coarse=False):
        self.model, self.cfg, self.device = model.to(device).eval(), cfg, device
        self.use_graph, self.coarse = use_graph and device.type == "cuda", coarse
        self._static, self._graph, self._out, self._copy = None, None, None, None

    def _forward(self, s):
        o = self.model(s["rgb"], s["lidar"], s["raw_point_xyz"], None, s["init_intrinsic"], None, None, None,
                       s["lidar_feats"], cfg=self.cfg)
        return o[0], o[1]

    def _capture(self, sample):
        self._static = {k: sample[k].to(self.device).float().clone() for k in _INPUT_KEYS}
        side = torch.cuda.Stream(self.device); side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._forward(self._static)
        torch.cuda.current_stream(self.device).wait_stream(side)
        from .train import _quiesce_watchdog           # (a live process group's watchdog must have nothing to poll while the
        _quiesce_watchdog(self.device)                 #  two-stream forward is captured: train._CAPTURE_MODE)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):     # see train._CAPTURE_MODE
            self._out = self._forward(self._static)

    def _stage(self, sample):
        """Start the host->device copies of one sample dict on the copy stream (overlaps the previous forward)."""
        if not self.use_graph:
            return sample, None
        if self._copy is None:
            self._copy = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self._copy):
            dev = {k: sample[k].to(self.device, non_blocking=True).float() for k in _INPUT_KEYS}
            ev = torch.cuda.Event(); ev.record(self._copy)
        return {**sample, **dev}, ev

    def _run(self, staged, ready):
        """(out3, out4) for one staged sample dict; the static-buffer/graph path when the shapes match."""
        if self.use_graph:
            cur = torch.cuda.current_stream(self.device)
            if self._graph is None:
                self._copy.synchronize()
                self._capture(staged)
            cur.wait_event(ready)
            for k in _INPUT_KEYS:
                staged[k].record_stream(cur)
            if all(staged[k].shape == self._static[k].shape for k in _INPUT_KEYS):
                for k in _INPUT_KEYS:
                    self._static[k].copy_(staged[k], non_blocking=True)      # device->device, ~45 MB
                self._graph.replay()
                return self._out[0].clone(), self._out[1].clone()
            return self._forward(staged)
        return self._forward({k: staged[k].to(self.device).float() for k in _INPUT_KEYS})

    @torch.no_grad()
    def evaluate(self, loader, log_path=None, metric_path=None, rot_test=None):
        """Timing: `mean_time_ms` / `mean_FPS` are the reference's per-batch span (inputs handed to the model ->
        outputs ready, evaluation_proj.py:238-263) measured with HIP events; the host->device copy of batch i+1 runs
        on a copy stream under the forward of batch i, so it is outside that span — `wall_samples_per_s` (samples / wall
        clock of the whole loop, loader and copies included) is the end-to-end rate."""
        dev, cuda = self.device, self.device.type == "cuda"
        ev = RteRreEval()
        ev_coarse = RteRreEval() if self.coarse else None
        errs, events, host_t = [], [], []
        n_samples, wall0 = 0, None
        it = iter(loader)
        nxt = next(it, None)
        staged = self._stage(nxt) if nxt is not None else None
        while staged is not None:
            sample, ready = staged
            if cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                t0 = time.perf_counter()
            out3, out4 = self._run(sample, ready)
            if cuda:
                e1.record(); events.append((e0, e1))
            else:
                host_t.append(time.perf_counter() - t0)
            if wall0 is None:                                     # wall clock starts after the capturing batch
                if cuda:
                    torch.cuda.synchronize(dev)
                wall0, n0 = time.perf_counter(), n_samples + out3.shape[0]
            nxt = next(it, None)
            staged = self._stage(nxt) if nxt is not None else None          # copies overlap this batch's forward
            _, _, pred_raw, gt_raw = getExtrinsic(out3, sample, out_raw=True)
            errs.append(torch.stack(calibration_error_batch(pred_raw, gt_raw), -1))     # [B,6]
            ev.addBatch(pred_raw, gt_raw)
            if ev_coarse is not None:
                _, _, pc, _ = getExtrinsic(out4, sample, out_raw=True)
                ev_coarse.addBatch(pc, gt_raw)
            n_samples += out3.shape[0]
        if cuda:
            torch.cuda.synchronize(dev)
            host_t = [a.elapsed_time(b) * 1e-3 for a, b in events]
        wall = time.perf_counter() - wall0
        # the reference's AverageMeter over batches; the graph-capturing first batch is not representative
        steady = host_t[1:] if len(host_t) > 1 and self.use_graph else host_t
        mean_time = float(np.mean(steady))
        e = torch.cat(errs).mean(0).cpu().numpy()
        rte_mean, rte_std, rre_mean, rre_std = ev.evalSeq()
        res = {"samples": n_samples, "mean_time_ms": mean_time * 1e3, "mean_FPS": 1.0 / mean_time,
               "wall_samples_per_s": (n_samples - n0) / wall if n_samples > n0 else float("nan"),
               "mean_roll_error": e[0], "mean_pitch_error": e[1], "mean_yaw_error": e[2],
               "mean_x_error": e[3], "mean_y_error": e[4], "mean_z_error": e[5],
               "mean_rotation_error": float(e[:3].mean()), "mean_translation_error": float(e[3:].mean()),
               "RTE": rte_mean, "RTE_std": rte_std, "RRE": rre_mean, "RRE_std": rre_std}
        if ev_coarse is not None:
            res["RTE_coarse"], _, res["RRE_coarse"], _ = ev_coarse.evalSeq()
        if log_path is not None:
            self.write_log(log_path, res, rot_test)
        if metric_path is not None:
            ev.save_metric(metric_path)
        return res

    @staticmethod
    def write_log(path, res, rot_test=None):
        """The text block of evaluation_proj.py:362-402 (same keys, same formats)."""
        with open(path, "a") as f:
            if rot_test is not None:
                f.write("rot_test_set= {:3f}\n".format(rot_test))
            f.write("mean_FPS= {:3f}\n".format(res["mean_FPS"]))
            f.write("mean_time= {:3f} ms\n".format(res["mean_time_ms"]))
            for k in ("roll", "pitch", "yaw", "x", "y", "z"):
                f.write("mean_{}_error= {:3f}\n".format(k, res[f"mean_{k}_error"]))
            f.write("mean_rotation_error= {:3f}\n".format(res["mean_rotation_error"]))
            f.write("mean_translation_error= {:3f}\n".format(res["mean_translation_error"]))
            f.write("RTE %.2f +- %.2f, RRE %.2f +- %.2f\n" % (res["RTE"], res["RTE_std"], res["RRE"], res["RRE_std"]))


def main(argv=None):
    """`python -m i2pnet_amd.evaluate [--ckpt model.pth] [--batches 20] [--batch 8] [--log eval.txt]`: the
    evaluation loop over SYNTHETIC sample dicts (`i2pnet_amd.synth`; there is no dataset in this image) — the
    reference's evaluation_proj.py on its KITTI loader.  Prints the result dict as one JSON line."""
    import argparse
    import json
    from . import synth
    from .config import CONFIGS
    from .model import RegNet_v2, RegNet_v2_iter
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt"); ap.add_argument("--config", default="config_proj_lidarcenter", choices=sorted(CONFIGS))
    ap.add_argument("--batches", type=int, default=20); ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iterative", action="store_true"); ap.add_argument("--log"); ap.add_argument("--metrics")
    a = ap.parse_args(argv)
    cfg, dev = CONFIGS[a.config], torch.device("cuda", 0)
    torch.manual_seed(0)
    net = (RegNet_v2_iter if a.iterative else RegNet_v2)(cfg=cfg)
    if a.ckpt:
        load_checkpoint(net, a.ckpt)
    nus = a.config.endswith("_nus")
    def pinned(d):                                                # what DataLoader(pin_memory=True) hands over
        return {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in d.items()}
    loader = (pinned(synth.make_batch(a.batch, 16384 if nus else 8192, 160 if nus else 375, 512 if nus else 1242, seed=100 + i,
                                      device=torch.device("cpu"), beams=32 if nus else 64)) for i in range(a.batches))
    res = Evaluator(net, cfg, dev).evaluate(loader, log_path=a.log, metric_path=a.metrics)
    print(json.dumps({k: (float(v) if not isinstance(v, int) else v) for k, v in res.items()} | {"data": "synthetic"}))


if __name__ == "__main__":
    main()
