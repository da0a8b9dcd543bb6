"""Section counters / timers of the diagnostic build (make the library with -DVPT_COUNTERS, point VPT_HIP_LIB at it).
Run from the repo root on a GPU box: VPT_HIP_LIB=$PWD/volumetric-path-tracer_amd/libvpt_hip_cnt.so python profiles/tools/section_counters.py"""
import os, sys, ctypes
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, vpt_loader
vpt = vpt_loader.load()
lib = ctypes.CDLL(os.environ['VPT_HIP_LIB'])
scene = vpt.HostScene(sys.argv[1] if len(sys.argv) > 1 else 'tests/golden/scenes/03_volume/volume.json')
dev = vpt.DeviceScene(scene, 0)
p = vpt.PathtraceParams(resolution=1280, samples=1 << 30, shader='volpathtrace', bounces=64)
st = scene.make_state(p)
spp = 16
out = (ctypes.c_ulonglong * 64)()
lib.vpt_debug_counts(out, 1)
import time
dev.pathtrace_samples(st, p, 1)
lib.vpt_debug_counts(out, 1)
if hasattr(lib, "vpt_debug_hist"):
    lib.vpt_debug_hist(None, 1)
t0 = time.perf_counter()
dev.pathtrace_samples(st, p, spp)
dt = time.perf_counter() - t0
lib.vpt_debug_counts(out, 0)
hist = (ctypes.c_ulonglong * 24)()
if hasattr(lib, "vpt_debug_hist"):
    lib.vpt_debug_hist(hist, 0)
names = ['node step', 'prim test', 'instance entry', 'outer iteration', 'trip (query)', 'pop', 'miss', 'surface', 'volume', 'lights', 'generate', 'leaf',
         'group node phase (lanes = rays handed over)', 'group node step (lanes = 4 x rays)', 'group leaf step (lanes = rays)', 'wide node step (lanes = 32 x rays)']
nsamp = st.width * st.height * spp
slots = nsamp / 64
print(f"{'section':52s} {'wave-exec/sample-slot':>22s} {'lane-exec/sample':>18s} {'avg lanes':>10s}")
for k, n in enumerate(names):
    w, l = out[2 * k], out[2 * k + 1]
    if w: print(f"{n:52s} {w / slots:22.2f} {l / nsamp:18.2f} {l / w:10.1f}")

tn = ['A node loops', 'B prim phases', 'C instance entry', 'whole query', 'trip', 'lights pdf loop + MIS', 'sample_lights', 'surface event', 'volume event', 'generate', 'kernel', 'light CDF search', 'scatter eval (bsdf/phase)', 'medium distance sampling', 'surface: position+normal+material', 'surface: delta lobe']
tot = out[32 + 10]
print()
for k, n in enumerate(tn):
    if out[32 + k]: print(f"{n:24s} {100.0 * out[32 + k] / tot:6.2f} % of wave time")

print(f"host time of the launch {dt*1e3:.1f} ms; sum of wave times {tot} ticks; if ticks are 10 ns: mean waves in flight {tot*1e-8/dt:.0f} of {256*4*3}")

if sum(hist):
    tot_h = sum(hist)
    print("group-form node steps by rays taking part: " + ", ".join(f"{k}: {hist[k] / tot_h:.3f}" for k in range(17) if hist[k]))
