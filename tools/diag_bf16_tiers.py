"""configs[2] (batch 16, bf16 storage) against the fp32 reference fixture with the first k blocks of the image encoder kept in fp32
(I2P_IMG_FP32_BLOCKS=k): pose / loss / worst activation deviation per k — the trade-off VERDICT r3 #4 asks for.
    python tools/diag_bf16_tiers.py [k ...]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402

import test_model_sized as T  # noqa: E402

ks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 5]
for k in ks:
    os.environ["I2P_IMG_FP32_BLOCKS"] = str(k)
    torch.manual_seed(0)
    gold, model, acts, out3, out4, loss = T._run_sized("kitti_b16", "cuda", precision="bf16")
    G = T._gen()
    worst = 0.0
    for name, t in acts.items():
        want = torch.as_tensor(gold[f"act.{name}.rows"]).double()
        _, rows = G.tensor_digest(t.detach().cpu(), name)
        worst = max(worst, float((torch.as_tensor(rows).double() - want).norm() / want.norm()))
    print(f"fp32 blocks {k:2d}: out3 {T._rel(out3.detach().cpu(), gold['out3']):.3e} out4 {T._rel(out4.detach().cpu(), gold['out4']):.3e} "
          f"loss {abs(loss.item() - gold['loss'][0]) / abs(gold['loss'][0]):.3e} worst activation L2 {worst:.3e}", flush=True)
    del model, acts, out3, out4, loss
    torch.cuda.empty_cache()
