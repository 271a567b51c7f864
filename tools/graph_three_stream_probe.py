"""hipGraph capture with a stream forked from a forked stream (ROCm 7.0.2 / PyTorch 2.10): variants a, c, d, e, f join the inner stream
back into the stream it was forked from and crash in capture_end (segmentation fault inside the HIP runtime); variant g joins it into
the capturing stream instead and works.   python tools/graph_three_stream_probe.py <a|c|d|e|f|g>"""
import sys, torch
dev = torch.device("cuda", 0)
CYC = 20000
variant = sys.argv[1] if len(sys.argv) > 1 else "a"
side = torch.cuda.Stream(dev)
auxs = [torch.cuda.Stream(dev) for _ in range(5)]
n = 2 if variant == "f" else 5
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    torch.cuda._sleep(CYC)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for i in range(n):
            aux = auxs[i] if variant == "e" else auxs[0]
            torch.cuda._sleep(CYC)
            if variant != "c" or i == 0:
                aux.wait_stream(side)
            with torch.cuda.stream(aux):
                torch.cuda._sleep(CYC)
            torch.cuda._sleep(CYC)
            if variant == "d":
                side.wait_stream(aux)
        if variant == "g":
            torch.cuda._sleep(CYC)
            main.wait_stream(auxs[0])
        elif variant == "e":
            for a in auxs:
                side.wait_stream(a)
        else:
            side.wait_stream(auxs[0])
    torch.cuda._sleep(CYC)
    main.wait_stream(side)
    torch.cuda._sleep(CYC)
g.replay(); torch.cuda.synchronize()
print("three-stream capture ok", variant)
