"""A hipMemsetAsync captured into a HIP graph between two kernels that use the same buffer (this image: torch 2.10 + ROCm 7.0).
Per replay: kernel fills `buf` with ones -> memset node zeroes it -> kernel sums it.  The sum must be 0 every time.
`python profiles/experiments/graph_memset_node_repro.py [MiB]` (default 160) prints the sums of five replays."""
import ctypes
import sys

import torch

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 160
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda:0")
n = mib * (1 << 20) // 4
buf = torch.zeros((n,), device=dev)
other = torch.zeros((n,), device=dev)
out = torch.zeros((2,), device=dev)


def body():
    buf.fill_(1.0)                                   # kernel: every element 1
    other.fill_(3.0)
    s = torch.cuda.current_stream().cuda_stream
    rc = hip.hipMemsetAsync(buf.data_ptr(), 0, n * 4, s)    # memset node
    assert rc == 0, rc
    out[0] = buf.sum()                               # kernels: must see zeros
    out[1] = other.sum()


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("eager: sum after the memset = %.1f (expected 0)" % float(out[0]))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
for rep in range(5):
    g.replay()
    torch.cuda.synchronize()
    print("replay %d: sum after the memset = %.1f (expected 0), the other buffer %.1f (expected %.1f)" % (rep, float(out[0]), float(out[1]), 3.0 * n))
