"""The sized reference fixtures (tests/golden/model_kitti_b8.npz / _b16.npz) through the model with the encoder's first block on
csrc/image_first.hip against the same model with MIOpen's convolution there (I2P_NO_IMG_FIRST=1), in one process:
  miopen_first / fused_first   out3 / out4 / loss / activation errors against the reference fixture, and the gradient-norm check
  knn_fused_vs_miopen          neighbour sets of the fine cost volume that differ between the two passes, with the relative gap of the
                               exchanged squared distances (a flip is a legitimate fp32 outcome iff it is a near-tie)
  fused_first_given_knn        the fused pass with the MIOpen pass's neighbour sets: the fp32 contract given the integer decisions
usage: first_block_parity.py [kitti_b8|kitti_b16]"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "tools"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
import torch  # noqa: E402

import bench_mode_parity as BMP  # noqa: E402
import test_model_sized as T  # noqa: E402


def one(tag, env, knn_from=None):
    os.environ["I2P_NO_IMG_FIRST"] = env
    gold, model, rep, knn = BMP.run_pass(tag, knn_from=knn_from)
    worst, worst_key, checked = T._grad_norm_check(gold, model, 1e-3, 1.5e-2)
    rep["grad_worst_ratio_to_limit"] = worst
    rep["grad_worst_key"] = str(worst_key)
    del model
    torch.cuda.empty_cache()
    return rep, knn


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "kitti_b16"
    torch.backends.cudnn.benchmark = False
    res = {"tag": tag}
    res["miopen_first"], knn0 = one(tag, "1")
    res["fused_first"], knn1 = one(tag, "0")
    res["knn_fused_vs_miopen"] = BMP.flips(knn1, knn0)
    res["fused_first_given_knn"], _ = one(tag, "0", knn_from=knn0)
    print("FIRST_BLOCK_PARITY " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
