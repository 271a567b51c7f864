"""Register / scratch / LDS table of every kernel in libi2p_ops.so: `python tools/resource_usage.py [file.hip ...] > profiles/rNN_resource_usage.txt`.

Compiles each source with the library's own flags plus `-Rpass-analysis=kernel-resource-usage` (objects go to a
temporary directory, the library is untouched) and prints one line per kernel; kernels with scratch are marked.
"""
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from i2pnet_amd import build as B  # noqa: E402

KEYS = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill",
        "LDS Size [bytes/block]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"\(anonymous namespace\)::", "", s).split("(")[0].replace("void ", "") for s in out.splitlines()]


def one(src, tmp):
    cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", str(B.CSRC / src),
                                                                            "-o", str(Path(tmp) / (src + ".o"))]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=str(B.CSRC)).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?):\s+(\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None and k in KEYS:
            cur[k] = int(v)
    for r, n in zip(rows, demangle([r["name"] for r in rows])):
        r["name"] = n
    return src, rows


def main():
    srcs = sys.argv[1:] or B._sources()
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(8) as ex:
        res = list(ex.map(lambda s: one(s, tmp), srcs))
    print(f"{'kernel':78s} {'sgpr':>5s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>8s} {'occ':>4s} {'lds':>7s}")
    spilled = 0
    for src, rows in res:
        print(f"# {src}")
        for r in rows:
            sc = r.get("ScratchSize [bytes/lane]", 0)
            spilled += sc > 0
            print(f"{r['name'][:78]:78s} {r.get('TotalSGPRs', 0):5d} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {sc:8d} "
                  f"{r.get('Occupancy [waves/SIMD]', 0):4d} {r.get('LDS Size [bytes/block]', 0):7d}{'   <-- scratch' if sc else ''}")
    print(f"# kernels with scratch: {spilled}")


if __name__ == "__main__":
    main()
