import sys, subprocess
if len(sys.argv) > 1:
    import faulthandler; faulthandler.enable()
    import torch
    mode = sys.argv[1]
    dev = torch.device("cuda:0")
    x = torch.randn(1000, 3, device=dev, requires_grad=True)
    w = torch.randn(3, 3, device=dev, requires_grad=True)
    def it():
        y = (x @ w).relu().sum()
        y.backward()
        return y
    for _ in range(3):
        x.grad = None; w.grad = None
        it()
    torch.cuda.synchronize()
    keep = None
    if "live_graph" in mode:
        keep = (x * 2.0).sum()
    if "live_backwarded" in mode:
        x.grad = None; w.grad = None
        keep = it()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        x.grad = None; w.grad = None
        it()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    x.grad = None; w.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = it()
    print("captured", flush=True)
    g.replay(); torch.cuda.synchronize()
    print("replayed", float(y), flush=True)
else:
    for mode in ("plain", "live_graph", "live_backwarded"):
        r = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True, timeout=300)
        print(mode, "rc", r.returncode, "|", r.stdout.strip().replace("\n", " ; ")[:300], "|", [l for l in r.stderr.strip().splitlines() if "File" in l][:3] if r.returncode else "", flush=True)
