"""fp32 MFMA ceiling on this box: is the 0.68-0.70 of 157.3 TF of the weights-in-registers layer kernels the
design's limit (one wave per SIMD, 512 VGPRs), the power cap, or the HBM streams on top?  Compiles tools/mfma_ceiling.hip
(hipcc, gfx950) into tools/_build/ and times

    (i)   many waves per SIMD, register operands, independent accumulators (the guide's 155 TF pattern)
    (ii)  ONE wave per SIMD with 512 VGPRs, 256 back-to-back MFMAs per strip (16x16x4 and 32x32x2)
    (iii) (ii) + the HBM streams of a forward (8 loads + 8 stores of 16 B per lane per strip) / of a dgrad (24 + 8)
    (iv)  (iii) with the loads issued one strip ahead and the stores one strip behind (software pipelining inside the wave)

each with the engine clock / socket power sampled by rocm-smi while it runs.  Prints one JSON line per probe."""
import ctypes as C
import json
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
SO = HERE / "_build" / "libmfma_ceiling.so"
PEAK = 157.3


def build():
    src = HERE / "mfma_ceiling.hip"
    if not SO.exists() or SO.stat().st_mtime < src.stat().st_mtime:
        SO.parent.mkdir(exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=1000000", "-fPIC", "-shared", str(src), "-o", str(SO)], check=True)
    lib = C.CDLL(str(SO))
    lib.probe_many_waves.restype = C.c_double
    lib.probe_many_waves.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.probe_one_wave.restype = C.c_double
    lib.probe_one_wave.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


class Sampler:
    def __init__(self):
        self.stop, self.rows = False, []

    def run(self):
        while not self.stop:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            sclk = [l for l in out.splitlines() if "sclk" in l]
            pw = [l for l in out.splitlines() if "Power" in l]
            self.rows.append((sclk[0].split(":")[-1].strip() if sclk else "?", pw[0].split(":")[-1].strip() if pw else "?"))
            time.sleep(0.25)


def timed(name, launch, seconds=2.0):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        launch(st)
    torch.cuda.synchronize()
    samp = Sampler(); th = threading.Thread(target=samp.run); th.start()
    t_end = time.time() + seconds
    best, flop = 1e30, 0.0
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            flop = launch(st)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5 * 1e-3)
    samp.stop = True; th.join()
    tf = flop / best / 1e12
    print(json.dumps({"probe": name, "TFLOPs": round(tf, 1), "frac_of_157.3": round(tf / PEAK, 3), "best_ms": round(best * 1e3, 3),
                      "sclk_power_samples": samp.rows[-4:]}), flush=True)
    return tf


def main():
    lib = build()
    dev = torch.device("cuda", 0)
    out = torch.zeros(1 << 20, device=dev)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    for nacc, waves in ((4, 8), (8, 4), (4, 2), (4, 1)):
        blocks = ncu * waves
        timed(f"(i) many waves: {waves} waves/SIMD x {nacc} independent 16x16x4 accumulators, register operands",
              lambda st, a=nacc, b=blocks: lib.probe_many_waves(a, b, 4000, out.data_ptr(), st))
    strips = 400
    blocks = ncu
    rows_f4 = blocks * 4 * strips * 24 * 64                   # float4s the widest probe reads
    src = torch.empty(rows_f4 * 4, device=dev).normal_()
    dst = torch.empty(blocks * 4 * strips * 8 * 64 * 4, device=dev)
    names = {0: "(ii) one wave/SIMD, 512 VGPRs, 256 x 16x16x4 per strip, no memory traffic",
             1: "(iii) one wave/SIMD + forward streams (8 x 16 B loads + 8 x 16 B stores per lane per strip)",
             2: "(iii) one wave/SIMD + dgrad streams (24 loads + 8 stores per lane per strip)",
             3: "(ii') one wave/SIMD, 128 x 32x32x2 per strip, no memory traffic",
             4: "(iii') one wave/SIMD, 32x32x2 + forward streams"}
    names.update({5: "(iv) one wave/SIMD, 16x16x4 + forward streams, loads one strip ahead / stores one strip behind",
                  6: "(iv') one wave/SIMD, 32x32x2 + forward streams, loads one strip ahead / stores one strip behind",
                  7: "(iv) one wave/SIMD, 16x16x4 + dgrad streams (24 + 8), pipelined",
                  8: "(iv') one wave/SIMD, 32x32x2 + dgrad streams (24 + 8), pipelined"})
    for mode in (0, 3, 1, 4, 2, 5, 6, 7, 8):
        timed(names[mode], lambda st, m=mode: lib.probe_one_wave(m, blocks, strips, out.data_ptr(), src.data_ptr(), dst.data_ptr(), st))
        byts = blocks * 4 * strips * 64 * 16 * {0: 0, 3: 0, 1: 16, 4: 16, 2: 32, 5: 16, 6: 16, 7: 32, 8: 32}[mode]
        if byts:
            print(f"   (streams {byts / 1e9:.2f} GB per launch)")


if __name__ == "__main__":
    sys.exit(main())
