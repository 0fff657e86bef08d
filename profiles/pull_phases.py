"""Times the forward's binning stages of the fog frame with the whole-list build of k_tile_pull cut short (GFT_PULL_DBG: 1 =
behind pass A, 2 = behind the head / whole decision, 5 = behind pass B of every chunk, without the sorters; +16 = every tile
hinted, +32 = no depth gathers, +64 = no appearance marks, +128 = no accumulator clear, +256 = no pool atomic).  Results of
such runs are NOT valid; the tile_sort stage it prints includes the idle k_tail_build launch (~6 us).
`python profiles/pull_phases.py` on the GPU box."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch, bench
    from gftorf_amd import _lib
    dev = torch.device("cuda:0")
    scene = bench.build_scene(sys.argv[1], 0, 1)
    step, state, leaf = bench.gpu_step_fn(scene, dev)
    for _ in range(10): step()
    torch.cuda.synchronize()
    _lib.profile_reset(); _lib.profile_enable(True)
    for _ in range(30): step()
    torch.cuda.synchronize()
    p = _lib.profile_read()
    print(json.dumps({k: round(v / p["forward_calls"] * 1e3, 1) for k, v in p.items() if k.endswith("_ms")}))
else:
    for wl in ("fog",):
        for dbg in ("16", "17", "18", "21", str(16+5+32), str(16+5+32+64), str(16+5+32+64+128), str(16+5+256), str(16+5+32+64+256)):
            env = dict(os.environ, GFT_PULL_DBG=dbg)
            r = subprocess.run([sys.executable, __file__, wl], env=env, capture_output=True, text=True, timeout=300)
            print(wl, "dbg", dbg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
