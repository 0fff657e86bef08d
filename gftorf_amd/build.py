"""In-tree build of libgftorf_rast.so (hipcc, gfx950 only).

``python -m gftorf_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the resulting .so sits next to this file so it
travels with the repository snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libgftorf_rast.so")
SOURCES = ["gft_api.hip", "k_preprocess.hip", "k_binning.hip", "k_pull.hip", "k_render.hip", "k_assemble.hip", "k_knn.hip", "k_adam.hip", "k_deform.hip", "k_densify.hip", "k_loss.hip"]
ARCH = "gfx950"
# The SLP vectoriser packs the render kernels' scalar fp32 maths into v_pk_* ops that need extra
# v_mov to pair registers: measured +12 us per render kernel on the metric frame.
# (k_preprocess.hip: the vectoriser pairs scalar multiplies of the appearance maths into v_pk_mul_f32 behind four register moves
# each: preprocess_fwd stage 57.3 -> 54.7 us on the metric frame, 129 -> 126 fog, 132.6 -> 130.1 at 5 M @ 1080p without it)
# (k_pull.hip: it compiles the forward blend's walk too -- gft_render_walk.h -- and must do so exactly as k_render.hip does)
FILE_FLAGS = {"k_render.hip": ["-fno-slp-vectorize"], "k_preprocess.hip": ["-fno-slp-vectorize"], "k_pull.hip": ["-fno-slp-vectorize"]}


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libgftorf_rast.so cannot be built")


def flags():
    extra = os.environ.get("GFT_EXTRA_FLAGS", "").split()   # tuning experiments only
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function"] + extra


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, save_temps=False, tag=None, extra_flags=()):
    """tag=None: the product library.  tag="x": an experiment build (``extra_flags``, e.g. -DGFT_SOMETHING=0) with its own
    objects in _obj_x/ and its output in _abl/lib_x.so -- the product library and its objects are not touched
    (profiles/bench_with_lib.py runs the benches on such a library)."""
    OBJ, LIB = globals()["OBJ"], globals()["LIB"]
    if tag:
        OBJ = os.path.join(HERE, "_obj_" + tag)
        os.makedirs(os.path.join(HERE, "_abl"), exist_ok=True)
        LIB = os.path.join(HERE, "_abl", "lib_%s.so" % tag)
    os.makedirs(OBJ, exist_ok=True)
    import glob
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))) + [os.path.abspath(__file__)]
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [cc] + flags() + list(extra_flags) + FILE_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if save_temps:
                cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        logs = list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or force or _stale(LIB, objs):
        run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB, logs


if __name__ == "__main__":
    # python -m gftorf_amd.build [--force] [--save-temps] [--tag NAME -DFLAG ...]
    tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else None
    lib, logs = build(force="--force" in sys.argv, verbose=True, save_temps="--save-temps" in sys.argv, tag=tag,
                      extra_flags=[a for a in sys.argv[1:] if a.startswith("-D")])
    for l in logs:
        if l.strip():
            print(l)
    print("built", lib)
