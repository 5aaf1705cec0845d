"""Dev aid: time C3 (and optionally C2) with alternative builds of the library (knock-out experiments).
    python tools/gpu_variant.py gpurun_exp/librt_a.so gpurun_exp/librt_b.so ...
Results of such builds are NOT parity-checked; only kernel times are printed."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import subprocess

CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import raytrace_amd._lib as l
l.LIB_PATH = os.path.abspath(sys.argv[1])
import raytrace_amd as R
ctx = R.Context(0)
nx, ny, spp = 1920, 1080, int(os.environ.get('VAR_SPP', '100'))
ctx.upload_scene(R.Random_Scene(11, 0))
cam = R.book_camera(nx, ny)
cfg = R.make_config(nx, ny, spp, spp_chunk=R.RT_SPP_CHUNK_AUTO)
ts = []
for _ in range(4):
    res = ctx.render(cam, cfg)
    ts.append(res.kernel_ms)
print("%%-40s C3 kernel ms %%s  rays %%d" %% (os.path.basename(sys.argv[1]), " ".join("%%.2f" %% t for t in ts), res.rays), flush=True)
''' % ROOT

for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD, lib], timeout=300)
    if r.returncode != 0:
        print("variant", lib, "failed with", r.returncode, flush=True)
        break
