import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = a[k].astype(np.float64), b[k].astype(np.float64)
    if k == "snaps":
        T4 = a["tile_max"].size
        x = x.reshape(T4, 7, 3, 64, 4); y = y.reshape(T4, 7, 3, 64, 4)
        tm = a["tile_max"]; fl = np.repeat(a["front_len"], 4)
        for v in range(T4):
            lim = min(tm[v], fl[v])
            ncut = min((lim - 1) // 256, 7) if lim > 0 else 0
            for c in range(ncut):
                d = np.abs(x[v, c] - y[v, c])
                rel = d.max() / (np.abs(x[v, c]).max() + 1e-30)
                if rel > 1e-5:
                    print("snap unit", v, "cut", c + 1, "rel", rel, "lim", lim, "argmax", np.unravel_index(d.argmax(), d.shape), x[v, c].flat[d.argmax()], y[v, c].flat[d.argmax()])
        continue
    d = np.abs(x - y)
    print(k, "max abs diff", d.max(), "scale", np.abs(x).max(), "n>1e-5rel", int((d > 1e-5 * (np.abs(x).max() + 1e-30)).sum()))
