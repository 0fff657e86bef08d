"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol the
header declares, the header is valid C and its struct layouts equal the ctypes mirrors,
and the operator API fails loudly instead of falling back when it cannot run on a HIP device."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gftorf_rast.h")


@pytest.fixture(scope="module")
def lib():
    from gftorf_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from gftorf_amd import build
        build.build()
    return _lib.load()


HEADERS = [HEADER, os.path.join(ROOT, "include", "gftorf_assemble.h"), os.path.join(ROOT, "include", "gftorf_knn.h"),
           os.path.join(ROOT, "include", "gftorf_optim.h"), os.path.join(ROOT, "include", "gftorf_deform.h"),
           os.path.join(ROOT, "include", "gftorf_densify.h"), os.path.join(ROOT, "include", "gftorf_loss.h")]


def declared_functions():
    names = set()
    for h in HEADERS:
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(gft_[a-z_0-9]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol(lib):
    from gftorf_amd import _lib
    names = declared_functions()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.gft_abi_version() == _lib.ABI_VERSION


def test_header_is_plain_c_and_matches_ctypes(tmp_path, lib):
    from gftorf_amd import _lib
    prog = tmp_path / "abi.c"
    fields = {"gft_config": [f[0] for f in _lib.Config._fields_],
              "gft_forward_io": _lib.FORWARD_FIELDS, "gft_backward_io": _lib.BACKWARD_FIELDS,
              "gft_layout": _lib.LAYOUT_FIELDS,
              "gft_profile": _lib.PROFILE_FIELDS + ["forward_calls", "backward_calls"],
              "gft_assemble_io": _lib.ASSEMBLE_FIELDS, "gft_assemble_bwd_io": _lib.ASSEMBLE_BWD_FIELDS,
              "gft_deform_params": _lib.DEFORM_FIELDS, "gft_deform_grads": _lib.DEFORM_FIELDS,
              "gft_forward_hints": [f[0] for f in _lib.ForwardHints._fields_],
              "gft_forward_report": [f[0] for f in _lib.ForwardReport._fields_]}
    body = ['#include <stdio.h>', '#include <stddef.h>', '#include "gftorf_rast.h"', '#include "gftorf_assemble.h"',
            '#include "gftorf_deform.h"', 'int main(void){']
    for s, fl in fields.items():
        body.append('printf("%s %%zu\\n", sizeof(%s));' % (s, s))
        for f in fl:
            body.append('printf("%s.%s %%zu\\n", offsetof(%s,%s));' % (s, f, s, f))
    body.append('printf("ACC %d\\n", GFT_ACC_STRIDE); return 0;}')
    prog.write_text("\n".join(body))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    mirrors = {"gft_config": _lib.Config, "gft_forward_io": _lib.ForwardIO, "gft_backward_io": _lib.BackwardIO,
               "gft_layout": _lib.Layout, "gft_profile": _lib.Profile,
               "gft_assemble_io": _lib.AssembleIO, "gft_assemble_bwd_io": _lib.AssembleBwdIO,
               "gft_deform_params": _lib.DeformParams, "gft_deform_grads": _lib.DeformParams,
               "gft_forward_hints": _lib.ForwardHints, "gft_forward_report": _lib.ForwardReport}
    for s, cls in mirrors.items():
        assert int(out[s]) == C.sizeof(cls), s
        for f in fields[s]:
            assert int(out["%s.%s" % (s, f)]) == getattr(cls, f).offset, (s, f)
    assert int(out["ACC"]) == _lib.ACC_STRIDE


def test_size_queries_and_layout_need_no_gpu(lib):
    from gftorf_amd import _lib
    assert lib.gft_geom_bytes(0) > 0
    g1, g2 = lib.gft_geom_bytes(1000), lib.gft_geom_bytes(2000)
    assert 81 * 1000 <= g1 < g2      # 32+32+4+4+8+1 bytes per Gaussian
    assert lib.gft_image_bytes(640, 480) >= 640 * 480 * 16 + 1200 * 12
    L = _lib.get_layout(1000, 640, 480, 5000)
    offs = [getattr(L, n) for n in _lib.LAYOUT_FIELDS]
    # every region is 256-B aligned, except those that directly follow the 64-B ctrl block
    # (the ctrl words, the tile counters, the tile cuts and the supertile tables are contiguous: one clear)
    assert all(o % 256 == 0 for n, o in zip(_lib.LAYOUT_FIELDS, offs) if n not in ("img_tile_cnt", "img_tile_cut", "img_super_tab"))
    assert L.img_tile_cnt == L.img_ctrl + 64 + 128 and L.img_tile_cut == L.img_tile_cnt + 4 * 1200      # ctrl words + ticket counters
    assert L.img_super_tab == L.img_tile_cut + 4 * 1200
    assert L.geom_rec_b >= 32 * 1000 and L.bin_point_list >= 8 * 5000 and L.bin_total >= 12 * 5000 + 1200 * 2048 * 4
    assert lib.gft_binning_bytes(0, 640, 480) >= 0
    # capacity <-> bytes (what the pybind-level backward reads back from the buffer it is handed)
    for cap in (0, 64, 5056, 3_638_528):
        assert lib.gft_binning_capacity(lib.gft_binning_bytes(cap, 640, 480), 640, 480) == cap


def test_argument_errors_are_reported(lib):
    from gftorf_amd import _lib
    c = _lib.Config()
    c.P, c.W, c.H, c.D, c.M = 4, 64, 64, 5, 16
    io = _lib.ForwardIO()
    R = C.c_int64(0)
    assert lib.gft_forward_preprocess(None, C.byref(c), C.byref(io), C.byref(R), None) != 0
    assert "sh_degree" in _lib.last_error()
    c.D, c.M = 3, 4
    assert lib.gft_forward_preprocess(None, C.byref(c), C.byref(io), C.byref(R), None) != 0
    assert "coefficients" in _lib.last_error()
    c.M = 16
    assert lib.gft_forward_preprocess(None, C.byref(c), C.byref(io), C.byref(R), None) != 0
    assert "NULL" in _lib.last_error()


def test_operator_api_surface_and_loud_failure():
    import diff_gaussian_rasterization_w_tof as d
    from gftorf_amd import GaussianRasterizationSettings, GaussianRasterizer
    assert d.GaussianRasterizer is GaussianRasterizer
    # 18 fields, same names/defaults as the reference NamedTuple (__init__.py:22-40)
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug", "near_n", "far_n", "depth_range",
        "use_view_dependent_phase", "optimize_phase_offset", "optimize_dc_offset")
    assert GaussianRasterizationSettings._field_defaults == dict(
        near_n=0.01, far_n=100.0, depth_range=100.0, use_view_dependent_phase=False,
        optimize_phase_offset=False, optimize_dc_offset=False)
    eye = torch.eye(4)
    s = GaussianRasterizationSettings(image_height=32, image_width=32, tanfovx=0.5, tanfovy=0.5,
                                      bg=torch.zeros(7, 32, 32), scale_modifier=1.0, viewmatrix=eye,
                                      projmatrix=eye, sh_degree=0, campos=torch.zeros(3), prefiltered=False, debug=False)
    r = GaussianRasterizer(raster_settings=s)
    m = torch.rand(8, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=m[:, :1], scales=m, rotations=torch.rand(8, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(means3D=m, means2D=m, opacities=m[:, :1], shs=torch.rand(8, 1, 3), scales=m)
    # CPU tensors: no silent fallback
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=m, means2D=m, opacities=m[:, :1], shs=torch.rand(8, 1, 3), scales=m, rotations=torch.rand(8, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.markVisible(m)


def test_missing_library_is_an_error(tmp_path, monkeypatch):
    from gftorf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gftorf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "gft_oracle" not in txt, f


def test_bench_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout)


def test_deform_size_queries_and_argument_errors(lib):
    from gftorf_amd import _lib
    # 2 x 96*256 (layer 0, encoding rows of layer 5) + 7*65536 + 16384 forward, 16384 + 7*65536 backward, 2112 biases
    # + three bf16 planes of both weight streams + two fp16 planes of both (999424 weights, 2 x 2 bytes each; the backward
    # stream's since round 6) + the range flag's four words
    assert lib.gft_deform_packed_bytes() == (1001536 + 999424 * 3 // 2 + 999424 + 4) * 4
    assert lib.gft_deform_inputs(10, 10) == 84 and lib.gft_deform_inputs(10, 6) == 76      # reference config / class default
    assert lib.gft_deform_inputs(10, 16) == 96 == _lib.DEFORM_MAX_INPUTS and lib.gft_deform_inputs(10, 17) == -1
    assert lib.gft_deform_inputs(-1, 6) == -1
    assert lib.gft_deform_saved_bytes(0) == 0 and lib.gft_deform_scratch_bytes(0) == 0
    # per point (padded to 192): 96 encoding + 8*256 activations + 8*8 words of ReLU sign bits
    assert lib.gft_deform_saved_bytes(1) == 192 * (96 + 2048 + 64) * 4
    assert lib.gft_deform_saved_bytes(193) == 384 * (96 + 2048 + 64) * 4
    assert lib.gft_deform_scratch_bytes(1000) > 1024 * (2048 + 64) * 4
    assert lib.gft_deform_pack(None, 10, 10, None, None) != 0
    assert "NULL" in _lib.last_error()
    assert lib.gft_deform_pack(None, 10, 17, C.byref(_lib.DeformParams()), C.c_void_p(64)) != 0
    assert "encoded inputs" in _lib.last_error()
    assert lib.gft_deform_forward(None, 10, 10, -1, None, None, 1, None, None, None, None) != 0
    assert lib.gft_deform_forward(None, 10, 10, 0, None, None, 1, None, None, None, None) == 0      # nothing to do
    assert lib.gft_deform_forward(None, 10, 10, 5, None, None, 1, None, None, None, None) != 0
    assert lib.gft_deform_forward(None, 11, 16, 0, None, None, 1, None, None, None, None) != 0      # 102 inputs
    assert lib.gft_deform_backward(None, 10, 10, 5, None, None, None, None, None, None) != 0
