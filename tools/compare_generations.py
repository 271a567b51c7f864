"""End-to-end cross-check of the layer-kernel generations: one seeded training step (forward, loss, backward) at full size
with the operands-in-registers kernels (default) and with I2P_NO_WREG=1 (second-generation kernels everywhere), in two
processes; prints the loss and the relative difference of the flat gradient.

    python tools/compare_generations.py [kitti|nus] [batch]
"""
import os, subprocess, sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tag, batch, out):
    from i2pnet_amd import synth
    from i2pnet_amd.config import I2PNetConfig, I2PNetConfigNuScenes
    from i2pnet_amd.train import Trainer
    cfg = I2PNetConfigNuScenes if tag == "nus" else I2PNetConfig
    dev = torch.device("cuda", 0)
    tr = Trainer(cfg=cfg, device=dev, seed=0)
    for m in tr.net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    if tag == "nus":
        b = synth.make_batch(batch, 16384, 160, 512, seed=5, beams=32, fup=cfg.fup, fdown=cfg.fdown, device=dev)
    else:
        b = synth.make_batch(batch, 8192, 375, 1242, seed=3, device=dev)
    loss = tr._forward_backward(b)
    loss = loss[0] if isinstance(loss, (tuple, list)) else loss
    torch.save({"loss": float(loss), "grad": tr.flat_grad.detach().cpu()}, out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4]); sys.exit(0)
    tag = sys.argv[1] if len(sys.argv) > 1 else "kitti"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    res = {}
    for name, env in (("wreg", {}), ("gen2", {"I2P_NO_WREG": "1"})):
        out = f"/tmp/cmpgen_{name}.pt"
        subprocess.run([sys.executable, __file__, "--child", tag, str(batch), out], check=True, env={**os.environ, **env})
        res[name] = torch.load(out)
    a, b = res["wreg"], res["gen2"]
    ga, gb = a["grad"].double(), b["grad"].double()
    print(f"{tag} batch {batch}: loss {a['loss']:.6f} vs {b['loss']:.6f}; |grad| {ga.norm():.6e} vs {gb.norm():.6e}; "
          f"|diff| / |grad| = {(ga - gb).norm() / gb.norm():.2e}; max |diff| / max |grad| = {(ga - gb).abs().max() / gb.abs().max():.2e}")
