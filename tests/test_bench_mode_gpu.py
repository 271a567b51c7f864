"""The reference-pinned batch-8 fixture (BASELINE configs[1]) in the PROCESS CONFIGURATION `bench.py` measures (VERDICT r3 weak #1 /
next #1a): tools/bench_mode_parity.py runs in a fresh process — its own MIOpen find-db, `bench.process_setup()` = cudnn.benchmark /
exhaustive find, the solvers that find picks — and reports; this test asserts.

What is asserted, and why it is phrased this way.  The forward contract is 1e-4 on fp32 tensors and bit-exact integer outputs ON
IDENTICAL INPUTS.  Between the image features RF3 and the fine pose out3 the network takes one integer decision: the 32 nearest
pixels of every warped point (the fine cost volume's kNN, 1824 queries at batch 8).  A few of those queries sit at near-ties, so a
1e-5-level change of RF3 — another MIOpen solver (which one wins the find is timing noise between ConvHipImplicitGemmGroupFwdXdlops
and ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC), or a plain-torch CPU evaluation of the encoder — can exchange two neighbours at
equal distance to 1e-6, and a flipped neighbour set moves out3 by ~1e-3 (measured; the reference run on another BLAS would do the
same).  So under the bench's solvers:
  * everything upstream of that decision (the LiDAR pyramid, the coarse cost volume, the coarse pose out4, ...) holds 1e-4;
  * every flipped neighbour set must be a near-tie (relative squared-distance gap <= 1e-4) and they must be few (<= 1 %);
  * with the neighbour sets of the default-mode pass — which matches the reference to 1e-5, i.e. carries the reference's own
    decisions — the whole forward holds 1e-4 under the bench-mode solvers; with no flips it holds 1e-4 as is;
  * the gradient-norm check of tests/test_model_sized.py passes under the bench-mode backward solvers."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
TOL = 1e-4


@pytest.mark.gpu
def test_config1_batch8_fp32_bench_mode(hip_backend, tmp_path):
    env = dict(os.environ)
    env["MIOPEN_USER_DB_PATH"] = str(tmp_path / "udb")          # a fresh box's state, like the driver's bench run
    os.makedirs(env["MIOPEN_USER_DB_PATH"], exist_ok=True)
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "bench_mode_parity.py"), "kitti_b8"], capture_output=True, text=True,
                         env=env, cwd=str(ROOT), timeout=1500)
    lines = [l for l in out.stdout.splitlines() if l.startswith("BENCH_MODE_PARITY ")]
    assert out.returncode == 0 and lines, out.stderr[-3000:]
    r = json.loads(lines[-1].split(" ", 1)[1])
    print({k: r[k] for k in ("knn_bench_vs_default", "knn_cpu_rf3_vs_default", "rf3_vs_cpu_torch", "bench_mode_grad")})
    print("solvers:", sorted(set(r["solvers"].values())))
    assert r["cudnn_benchmark"] is True and r["chain_errors"] == 0
    # (blocks 1-5 of the encoder run on csrc/image_first.hip / image_conv16.hip: MIOpen keeps the 5 distinct shapes of blocks 6-15)
    assert len(r["solvers"]) >= 12, "the find-db of the process must name the solvers of the 5 + 5 + 5 convolution problems left on MIOpen"
    # the existing contract, same process, before the switch
    assert all(v <= TOL for v in r["default_mode"].values()), r["default_mode"]
    # bench mode: upstream of the integer decision
    bm = r["bench_mode"]
    assert all(bm[k] <= TOL for k in r["upstream_keys"]), {k: bm[k] for k in r["upstream_keys"]}
    assert r["rf3_vs_cpu_torch"]["l2_rel"] <= 1e-4
    # the decision itself
    fl = r["knn_bench_vs_default"]
    assert fl["flipped"] <= 0.01 * fl["queries"] and fl["worst_relative_distance_gap"] <= 1e-4, fl
    if fl["flipped"] == 0:
        assert all(v <= TOL for v in bm.values()), bm
    # given the decision: the whole forward under the bench-mode solvers
    assert all(v <= TOL for v in r["bench_mode_given_knn"].values()), r["bench_mode_given_knn"]
    # gradients under the bench-mode backward solvers (limits of test_model_sized.py: 1e-3 / RGB 1.5e-2 / 4x the reference's own floor)
    assert r["bench_mode_grad"]["checked"] > 100 and r["bench_mode_grad"]["worst_ratio_to_limit"] <= 1.0, r["bench_mode_grad"]
    # the diagnostic perturbation flips only near-ties as well
    assert r["knn_cpu_rf3_vs_default"]["worst_relative_distance_gap"] <= 1e-4, r["knn_cpu_rf3_vs_default"]
