"""Occupancy timeline of one launch from per-wave start/end stamps (library built with -DVPT_WAVE_TIMES).
Run from the repo root on a GPU box: VPT_HIP_LIB=$PWD/volumetric-path-tracer_amd/libvpt_hip_wt.so python profiles/tools/wave_timeline.py"""
import os, sys, ctypes
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, vpt_loader
vpt = vpt_loader.load()
lib = ctypes.CDLL(os.environ['VPT_HIP_LIB'])
scene = vpt.HostScene('tests/golden/scenes/03_volume/volume.json')
dev = vpt.DeviceScene(scene, 0)
p = vpt.PathtraceParams(resolution=1280, samples=1 << 30, shader='volpathtrace', bounces=64)
st = scene.make_state(p)
spp = 64
dev.pathtrace_samples(st, p, spp)
dev.pathtrace_samples(st, p, spp)
nw = (st.width * st.height + 63) // 64   # upper bound; tile-major layout pads
nw = min(nw + 64, 65536)
buf = np.zeros(2 * nw, np.uint64)
lib.vpt_debug_wave_times(buf.ctypes.data_as(ctypes.c_void_p), nw)
t0, t1 = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
ok = (t0 > 0) & (t1 > t0)
t0, t1 = t0[ok], t1[ok]
base = t0.min(); end = t1.max()
dur = (t1 - t0)
print("waves", ok.sum(), "launch span (ticks)", end - base, "mean wave", dur.mean(), "min", dur.min(), "max", dur.max())
print("mean waves in flight", dur.sum() / (end - base), "of", 256 * 4 * 3)
# timeline in 20 slices
edges = np.linspace(base, end, 21)
for a, b in zip(edges[:-1], edges[1:]):
    inflight = (np.minimum(t1, b) - np.maximum(t0, a)).clip(0).sum() / (b - a)
    print(f"{(a-base)/(end-base):5.2f} {inflight:8.0f}")
# cost by dispatch order deciles
idx = np.nonzero(ok)[0]
for d in range(10):
    sel = slice(d * len(idx) // 10, (d + 1) * len(idx) // 10)
    print("decile", d, "mean duration", dur[sel].mean())
