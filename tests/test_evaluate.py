"""Evaluator + metrics against vectors recorded from the reference's metric.py
(tools/gen_golden.py metrics -> tests/golden/metrics.npz) and against scipy directly."""
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from i2pnet_amd import evaluate as E

GOLD = Path(__file__).parent / "golden" / "metrics.npz"


def _load(dev="cpu"):
    z = np.load(GOLD)
    out3 = torch.from_numpy(z["out3"]).to(dev)
    dv = {"decalib_real_gt": torch.from_numpy(z["q_gt"]), "decalib_dual_gt": torch.from_numpy(z["t_gt"]),
          "init_extrinsic": torch.from_numpy(z["init"])}
    return z, out3, dv


def _check(z, out3, dv):
    pred, gt, pred_raw, gt_raw = E.getExtrinsic(out3, dv, out_raw=True)
    for name, mine in (("pred", pred), ("gt", gt), ("pred_raw", pred_raw), ("gt_raw", gt_raw)):
        np.testing.assert_allclose(mine.cpu().numpy(), z[name], rtol=1e-6, atol=1e-6, err_msg=name)
    np.testing.assert_allclose(E.inv_extrinsic(pred).cpu().numpy(), z["inv"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(E.rotmat_to_euler(gt[:, :, :3], out="deg").cpu().numpy(), z["euler"], rtol=1e-6, atol=1e-5)
    errs = torch.stack(E.calibration_error_batch(pred_raw, gt_raw), -1).cpu().numpy()
    np.testing.assert_allclose(errs, z["errs"], rtol=1e-5, atol=1e-5)
    r1, t1 = E.cal_rete_once(out3, dv)
    np.testing.assert_allclose([float(r1), float(t1)], z["rete_once"], rtol=1e-5, atol=1e-5)
    ev = E.RteRreEval()
    ev.addBatch(pred_raw[:20], gt_raw[:20]); ev.addBatch(pred_raw[20:], gt_raw[20:])
    r, t = ev._all()
    np.testing.assert_allclose(r, z["rre"], rtol=1e-5, atol=1e-4)          # degrees
    np.testing.assert_allclose(t, z["rte"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ev.evalSeq(), z["seq"], rtol=1e-5, atol=1e-4)
    evt = E.RteRreEval(threshold=True)
    evt.addBatch(pred_raw, gt_raw)
    assert evt.get_recall() == float(z["recall_th"])
    np.testing.assert_allclose(evt.evalSeq(), z["seq_th"], rtol=1e-5, atol=1e-4)


def test_metrics_against_reference_vectors():
    _check(*_load())


def test_euler_xzy_against_scipy():
    from scipy.spatial.transform import Rotation
    rot = Rotation.random(512, random_state=4)
    mine = E.euler_xzy_deg(torch.from_numpy(rot.as_matrix())).numpy()
    np.testing.assert_allclose(mine, rot.as_euler("xzy", degrees=True), rtol=1e-7, atol=1e-7)
    # non-orthogonal input: scipy projects onto SO(3) first
    noisy = rot.as_matrix()[:64] * 1.05 + np.random.default_rng(0).normal(0, 1e-2, (64, 3, 3))
    np.testing.assert_allclose(E.euler_xzy_deg(torch.from_numpy(noisy)).numpy(),
                               Rotation.from_matrix(noisy).as_euler("xzy", degrees=True), rtol=1e-6, atol=1e-6)


def test_checkpoint_formats(tmp_path):
    """evaluation_proj.py:134-139 loads ckpt["model_state_dict"]; the DDP training script saves under `module.`"""
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4))
    sd = {k: torch.randn_like(v.float()).to(v.dtype) for k, v in net.state_dict().items()}
    torch.save({"model_state_dict": {"module." + k: v for k, v in sd.items()}, "epoch": 7}, tmp_path / "a.pth")
    torch.save({"model_state_dict": sd}, tmp_path / "b.pth")
    for f in ("a.pth", "b.pth"):
        m = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4))
        ck = E.load_checkpoint(m, tmp_path / f)
        assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
    assert ck.get("epoch") is None
    bad = dict(sd); bad.pop("0.bias")
    torch.save({"model_state_dict": bad}, tmp_path / "c.pth")
    with pytest.raises(RuntimeError):
        E.load_checkpoint(torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4)), tmp_path / "c.pth")


def test_log_format(tmp_path):
    res = {"mean_FPS": 151.23456, "mean_time_ms": 6.6123, "mean_rotation_error": 0.5, "mean_translation_error": 0.25,
           "RTE": 1.234, "RTE_std": 0.5, "RRE": 2.345, "RRE_std": 0.75}
    res.update({f"mean_{k}_error": 0.125 for k in ("roll", "pitch", "yaw", "x", "y", "z")})
    E.Evaluator.write_log(tmp_path / "log.txt", res, rot_test=10.0)
    lines = (tmp_path / "log.txt").read_text().splitlines()
    assert lines[0] == "rot_test_set= 10.000000" and lines[1] == "mean_FPS= 151.234560" and lines[2] == "mean_time= 6.612300 ms"
    assert lines[-1] == "RTE 1.23 +- 0.50, RRE 2.35 +- 0.75" or lines[-1] == "RTE 1.23 +- 0.50, RRE 2.34 +- 0.75"


def test_evaluator_loop_cpu():
    """host logic of the loop (ragged last batch, accumulation, homogeneous init_extrinsic) with a stand-in network"""
    class Net(torch.nn.Module):
        def forward(self, rgb, lidar, raw, a, intr, b, c, d, feats, cfg=None):
            q = torch.nn.functional.normalize(torch.tensor([1.0, 0.02, -0.01, 0.03]) + 0 * rgb.mean((1, 2, 3)).unsqueeze(-1), dim=-1)
            o = torch.cat([q, lidar.mean(1)], -1)
            return o, o

    def loader():
        g = torch.Generator().manual_seed(0)
        for b in (2, 2, 1):
            yield {"rgb": torch.rand(b, 3, 4, 4, generator=g), "lidar": torch.rand(b, 10, 3, generator=g),
                   "raw_point_xyz": torch.rand(b, 10, 3), "init_intrinsic": torch.eye(3).repeat(b, 1, 1),
                   "lidar_feats": torch.rand(b, 10, 1), "init_extrinsic": torch.eye(4).repeat(b, 1, 1),
                   "decalib_real_gt": torch.tensor([[1.0, 0, 0, 0]]).repeat(b, 1), "decalib_dual_gt": torch.zeros(b, 3)}

    res = E.Evaluator(Net(), None, torch.device("cpu")).evaluate(loader())
    t = torch.cat([s["lidar"].mean(1) for s in loader()])
    assert res["samples"] == 5
    # RTE = |R^-1 (0 - t)| = |t| for a rotation
    assert abs(res["RTE"] - float(t.double().norm(dim=-1).mean())) < 1e-6
    from scipy.spatial.transform import Rotation
    q = torch.nn.functional.normalize(torch.tensor([1.0, 0.02, -0.01, 0.03]), dim=-1).double().numpy()
    want = np.abs(Rotation.from_quat(q[[1, 2, 3, 0]]).inv().as_euler("xzy", degrees=True)).sum()
    assert abs(res["RRE"] - want) < 1e-5 and res["RRE_std"] < 1e-6


@pytest.mark.gpu
def test_metrics_on_device():
    _check(*_load("cuda:0"))


@pytest.mark.gpu
def test_evaluator_graph_matches_eager(tmp_path):
    """The hipGraph / static-buffer loop gives the same metrics as eager per-batch forwards, handles a ragged last
    batch, and the metrics agree with the formulas applied to the raw outputs."""
    from i2pnet_amd import synth
    from i2pnet_amd.config import I2PNetConfig as cfg
    from i2pnet_amd.model import RegNet_v2
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = RegNet_v2(cfg=cfg).to(dev).eval()

    def batches():
        for i, b in enumerate((2, 2, 1)):
            s = synth.make_batch(b, 8192, 375, 1242, seed=10 + i, device=torch.device("cpu"))
            s["init_extrinsic"] = torch.eye(4)[:3].repeat(b, 1, 1)
            yield s

    res_g = E.Evaluator(net, cfg, dev, use_graph=True).evaluate(batches(), log_path=tmp_path / "g.txt",
                                                                 metric_path=tmp_path / "m.npz")
    res_e = E.Evaluator(net, cfg, dev, use_graph=False).evaluate(batches())
    assert res_g["samples"] == res_e["samples"] == 5
    for k in ("RRE", "RTE", "mean_roll_error", "mean_x_error", "mean_rotation_error"):
        assert math.isfinite(res_g[k])
        assert abs(res_g[k] - res_e[k]) <= 1e-3 * max(1.0, abs(res_e[k])), (k, res_g[k], res_e[k])
    saved = np.load(tmp_path / "m.npz")
    assert saved["RRE"].shape == (5,) and abs(saved["RTE"].mean() - res_g["RTE"]) < 1e-9
    assert res_g["mean_FPS"] > 0 and "mean_FPS=" in (tmp_path / "g.txt").read_text()
