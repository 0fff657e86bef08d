"""Debug helper: dump the forward's per-pixel state / snapshots of one small scene (serial vs segmented forward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as Hh
from gftorf_amd import _lib, api
kw = eval(sys.argv[2]) if len(sys.argv) > 2 else dict(P=4000, W=48, H=32, scale_lo=0.03, scale_hi=0.2, z_lo=3.0, z_hi=3.05)
sc = Hh.small_scene(**kw)
dev = torch.device("cuda:0")
api.keep_last_buffers = True
out, grads, t = Hh.run_gpu(sc, dev)
b = api.last_call_buffers
P, W, H = b["P"], b["W"], b["H"]
L = _lib.get_layout(P, W, H, b["cap"])
T = ((W + 15) // 16) * ((H + 15) // 16)
img = b["img"]
f32 = lambda o, n: img[o:o + 4 * n].view(torch.float32).cpu().numpy().copy()
u32 = lambda o, n: img[o:o + 4 * n].view(torch.int32).cpu().numpy().copy()
N = W * H
res = dict(pix_state=f32(L.img_pix_state, 4 * N), pix_sums=f32(L.img_pix_sums, 8 * N), tile_max=u32(L.img_tile_max, 4 * T),
           front_len=u32(L.img_front_len, T), ranges=u32(L.img_ranges, 2 * T), unit_flag=u32(L.img_unit_flag, 4 * T),
           snaps=f32(L.img_snaps, 4 * T * 7 * 3 * 64 * 4))
for k, v in out.items():
    res["out_" + k] = v
for k, v in grads.items():
    if v is not None:
        res["g_" + k] = v
np.savez(sys.argv[1], **res)
print("saved", sys.argv[1], "tile_max", res["tile_max"], "front", res["front_len"], "ranges", res["ranges"])
