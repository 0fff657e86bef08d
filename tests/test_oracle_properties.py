"""Domain properties of the oracle (size independent) and the edge cases the operator
must handle: empty / fully culled inputs, ragged image sizes, sorted keys, range cover."""
import numpy as np
import pytest

import helpers as Hh


def test_binning_invariants(oracle):
    sc = Hh.small_scene(P=800, W=90, H=70)
    f, _ = Hh.run_oracle(oracle, sc, backward=False)
    W, H = 90, 70
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    bit = oracle.get_higher_msb(T)
    keys = f.keys_sorted
    # sorted on the low 32+bit bits, stable: ties (same tile, same depth bits) keep index order
    assert (np.diff(keys.astype(np.uint64) & np.uint64((1 << (32 + bit)) - 1)) >= 0).all()
    same = keys[1:] == keys[:-1]
    assert (f.point_list[1:][same] > f.point_list[:-1][same]).all()
    # ranges partition the list by tile
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    for t in range(T):
        a, b = f.ranges[t]
        idx = np.nonzero(tiles == t)[0]
        if idx.size == 0:
            assert a == 0 and b == 0
        else:
            assert a == idx[0] and b == idx[-1] + 1
    assert f.num_rendered == int(f.geom["tiles_touched"].sum()) == keys.size
    # every instance's tile lies inside its Gaussian's rectangle
    np.testing.assert_array_equal(oracle.tiles_from_rect(W, H, f.geom["means2D"], f.radii), f.geom["tiles_touched"])
    # depth bits in the key are the view-space z of the value
    np.testing.assert_array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.uint32),
                                  f.geom["depths"][f.point_list].view(np.uint32))


def test_background_is_linear_in_final_transmittance(oracle):
    sc = Hh.small_scene(P=300)
    f1, _ = Hh.run_oracle(oracle, sc, backward=False)
    sc0 = dict(sc, bg=np.zeros_like(sc["bg"]))
    f0, _ = Hh.run_oracle(oracle, sc0, backward=False)
    T = f1.img["final_T"].reshape(sc["cfg"]["H"], sc["cfg"]["W"])
    np.testing.assert_allclose(f1.color - f0.color, T[None] * sc["bg"][:3], atol=2e-6)
    # phasor planes use the same background planes weighted by T (not T^2)
    np.testing.assert_allclose(f1.phasor - f0.phasor, T[None] * sc["bg"][:7], atol=2e-6)
    np.testing.assert_array_equal(f1.depth, f0.depth)
    assert (f1.acc <= 1.0 + 1e-5).all() and (f1.acc >= 0).all()
    np.testing.assert_allclose(f1.acc[0], 1.0 - T, atol=3e-6)    # sum of weights telescopes to 1 - T_final


def test_pixels_counts_and_first_hit(oracle):
    sc = Hh.small_scene(P=300)
    f, _ = Hh.run_oracle(oracle, sc, backward=False)
    assert f.pixels.sum() > 0 and (f.pixels == np.round(f.pixels)).all()
    assert (f.pixels[f.radii <= 0] == 0).all()
    hit = f.img["n_contrib"].reshape(f.H, f.W) > 0
    # distribution = (alpha, distance, amplitude/d^2) of the first blended splat
    assert (f.distribution[0][hit] >= 1.0 / 255.0 - 1e-7).all() and (f.distribution[0][hit] <= 0.99 + 1e-7).all()
    assert not f.distribution[:, ~hit].any()
    assert not f.normal.any() and not f.entropy.any() and not f.amp_distortion.any()


def test_order_independent_of_input_permutation(oracle):
    sc = Hh.small_scene(P=300)
    f, b = Hh.run_oracle(oracle, sc)
    perm = np.random.default_rng(0).permutation(300)
    g2 = {k: (v[perm] if v is not None else None) for k, v in sc["gaussians"].items()}
    f2, b2 = Hh.run_oracle(oracle, dict(sc, gaussians=g2))
    # depths are distinct here, so the per-tile order is the same and so is every sum
    Hh.assert_close("color", f.color, f2.color, rtol_max=1e-6)
    Hh.assert_close("phasor", f.phasor, f2.phasor, rtol_max=1e-6)
    np.testing.assert_array_equal(f.radii[perm], f2.radii)
    Hh.assert_close("dmeans", b["dL_dmeans3D"][perm], b2["dL_dmeans3D"], rtol_max=1e-5)


def test_empty_culled_and_ragged(oracle):
    sc = Hh.small_scene(P=50, W=33, H=17)
    f, b = Hh.run_oracle(oracle, sc)
    assert f.color.shape == (3, 17, 33) and np.isfinite(f.color).all()
    # P == 0
    e = dict(sc, gaussians={k: (v[:0] if v is not None else None) for k, v in sc["gaussians"].items()})
    f0, b0 = Hh.run_oracle(oracle, e)
    assert f0.num_rendered == 0 and not f0.color.any() and b0["dL_dmeans3D"].shape == (0, 3)
    # all culled by the depth range: background only
    far = dict(sc, gaussians=dict(sc["gaussians"], means3D=sc["gaussians"]["means3D"] + np.array([0, 0, 100], np.float32)))
    ff, bf = Hh.run_oracle(oracle, far)
    assert ff.num_rendered == 0 and not ff.radii.any()
    np.testing.assert_array_equal(ff.color, sc["bg"][:3])
    assert not bf["dL_dmeans3D"].any() and not bf["dL_dsh"].any()
    # prefiltered + culled point is an error (device trap in the reference)
    with pytest.raises(RuntimeError, match="prefiltered"):
        Hh.run_oracle(oracle, far, backward=False, prefiltered=True) if False else \
            oracle.forward(far["gaussians"]["means3D"], far["gaussians"]["opacities"], shs=far["gaussians"]["shs"],
                           shs_p=far["gaussians"]["shs_p"], scales=far["gaussians"]["scales"],
                           rotations=far["gaussians"]["rotations"], prefiltered=True, **Hh.oracle_kwargs(far))
    vis = oracle.mark_visible(far["gaussians"]["means3D"], sc["cam"]["viewmatrix"], sc["cam"]["projmatrix"],
                              sc["cam"]["znear"], sc["cam"]["zfar"])
    assert not vis.any()


def test_gradients_vanish_for_unused_outputs(oracle):
    """The caller renders a colour view and a ToF view and discards half of each
    (gaussian_renderer/__init__.py:107,117): zero upstream gradient -> zero contribution."""
    sc = Hh.small_scene(P=200)
    z = {k: np.zeros_like(v) for k, v in sc["grads"].items()}
    f, b = Hh.run_oracle(oracle, dict(sc, grads=z))
    for k in ["dL_dmeans3D", "dL_dsh", "dL_dsh_p", "dL_dscales", "dL_drotations", "dL_dopacity"]:
        assert not b[k].any(), k
    only_color = dict(z, color=sc["grads"]["color"])
    f, b = Hh.run_oracle(oracle, dict(sc, grads=only_color))
    assert b["dL_dsh"].any() and not b["dL_dsh_p"].any() and b["dL_dphase_offset"][0] == 0


def test_parallel_binning_stages_equal_the_serial_restatement(oracle):
    """The oracle's scan / sort / tile ranges run on all cores so that the CPU baseline is a fair one; the
    one-thread LSD radix restatement of cub::DeviceRadixSort::SortPairs (rasterizer_impl.cu:331-339) stays in the
    library as the definition: same bits on random keys with many ties, on a binned scene and for every thread count."""
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(5)
    threads = oracle.num_threads()
    try:
        for nthr in (1, 3, threads):
            oracle.set_num_threads(nthr)
            for R, T in ((40_000, 300), (200_000, 1200), (100_000, 8160), (70_000, 2)):
                bit = oracle.get_higher_msb(T)
                tiles = rng.integers(0, T, R).astype(np.uint64)
                depth = rng.integers(0, 50, R).astype(np.uint64) * np.uint64(0x01010101)     # many equal keys
                depth[::3] = rng.integers(0, 2 ** 32, depth[::3].size).astype(np.uint64)
                keys = (tiles << np.uint64(32)) | depth
                vals = rng.permutation(R).astype(np.uint32)
                out = []
                for fn in (L.gfto_sort_pairs_serial, L.gfto_sort_pairs):
                    ks, vs = np.zeros(R, np.uint64), np.zeros(R, np.uint32)
                    fn(C.c_uint32(R), C.c_void_p(keys.ctypes.data), C.c_void_p(vals.ctypes.data),
                       C.c_void_p(ks.ctypes.data), C.c_void_p(vs.ctypes.data), C.c_int(32 + bit))
                    out.append((ks, vs))
                np.testing.assert_array_equal(out[0][0], out[1][0])
                np.testing.assert_array_equal(out[0][1], out[1][1])
                ranges = np.zeros((T, 2), np.uint32)
                L.gfto_tile_ranges(C.c_uint32(R), C.c_void_p(out[1][0].ctypes.data), C.c_int(T), C.c_void_p(ranges.ctypes.data))
                st = (out[1][0] >> np.uint64(32)).astype(np.int64)
                cnt = np.bincount(st, minlength=T)
                first = np.searchsorted(st, np.arange(T))
                np.testing.assert_array_equal(ranges[:, 0], np.where(cnt > 0, first, 0))
                np.testing.assert_array_equal(ranges[:, 1], np.where(cnt > 0, first + cnt, 0))
            tt = rng.integers(0, 40, 300_001).astype(np.uint32)
            off = np.zeros_like(tt)
            L.gfto_scan.restype = C.c_uint32
            tot = L.gfto_scan(C.c_int(tt.size), C.c_void_p(tt.ctypes.data), C.c_void_p(off.ctypes.data))
            np.testing.assert_array_equal(off, np.cumsum(tt, dtype=np.uint64).astype(np.uint32))
            assert tot == int(off[-1])
    finally:
        oracle.set_num_threads(threads)


def test_kept_buffers_give_the_same_results(oracle):
    """bench.py times the oracle with its output arrays kept between iterations (oracle.reuse_buffers): same
    numbers as with fresh arrays, also when the second scene leaves rows untouched that the first one wrote."""
    a = Hh.small_scene(P=700, W=90, H=70, seed=8)
    b = Hh.small_scene(P=700, W=90, H=70, seed=9, z_lo=2.0)
    ref = []
    for sc in (a, b):
        f, g = Hh.run_oracle(oracle, sc)
        ref.append(({k: np.array(f[k]) for k in ("color", "phasor", "depth", "acc", "pixels", "radii", "point_list")},
                    {k: np.array(v) for k, v in g.items()}))
    oracle.reuse_buffers(True)
    try:
        for it in range(4):
            f, g = Hh.run_oracle(oracle, (a, b)[it % 2])
            rf, rg = ref[it % 2]
            for k, v in rf.items():
                np.testing.assert_array_equal(v, f[k], err_msg=k)
            for k, v in rg.items():
                np.testing.assert_allclose(v, g[k], rtol=1e-6, atol=1e-9, err_msg=k)   # (double sums in thread order)
    finally:
        oracle.reuse_buffers(False)
