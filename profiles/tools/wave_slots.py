"""Per-slot view of one K1 launch (library built with -DVPT_WAVE_TIMES): where do wave slots stand empty?
Every wave records start, end and the hardware slot it ran on (XCC, SE, CU, SIMD).  Run from the repo root on a GPU box:
  VPT_HIP_LIB=libvpt_hip_wt.so python3 profiles/tools/wave_slots.py [spp [scene file under tests/golden/scenes [shader [bounces [waves per SIMD]]]]]"""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import vpt_loader

vpt = vpt_loader.load()
lib = ctypes.CDLL(os.path.join(os.getcwd(), 'volumetric-path-tracer_amd', os.environ['VPT_HIP_LIB']))
scene_file = sys.argv[2] if len(sys.argv) > 2 else '03_volume/volume.json'
scene = vpt.HostScene('tests/golden/scenes/' + scene_file)
dev = vpt.DeviceScene(scene, 0)
shader = sys.argv[3] if len(sys.argv) > 3 else 'volpathtrace'
bounces = int(sys.argv[4]) if len(sys.argv) > 4 else 64
WPS = int(sys.argv[5]) if len(sys.argv) > 5 else 3
SLOTS = 1024 * WPS
p = vpt.PathtraceParams(resolution=1280, samples=1 << 30, shader=shader, bounces=bounces)
st = scene.make_state(p)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for _ in range(3):
    dev.pathtrace_samples(st, p, spp)
print("kernel ms", dev.last_kernel_ms())
nw = min((st.width * st.height + 63) // 64 + 4096, 65536)
buf = np.zeros(2 * nw, np.uint64)
hw = np.zeros(nw, np.uint32)
lib.vpt_debug_wave_times(buf.ctypes.data_as(ctypes.c_void_p), nw)
lib.vpt_debug_wave_hw(hw.ctypes.data_as(ctypes.c_void_p), nw)
t0, t1 = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
ok = (t0 > 0) & (t1 > t0)
t0, t1, hw = t0[ok], t1[ok], hw[ok]
base, end = t0.min(), t1.max()
span = end - base
t0, t1 = (t0 - base) / 1e5, (t1 - base) / 1e5   # ms
dur = t1 - t0
print(f"waves {ok.sum()}  span {span / 1e5:.2f} ms  longest {dur.max():.2f}  sum/slots {dur.sum() / SLOTS:.2f}  in flight {dur.sum() / (span / 1e5):.0f}")
xcc = (hw >> 16) & 15
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
print("distinct XCC", np.unique(xcc), "SE", np.unique(se), "SH", np.unique(sh), "CU", np.unique(cu), "SIMD", np.unique(simd))
print("XCC: waves, sum of wave ms / (128 SIMDs x waves per SIMD) slots, last end, first start")
for x in np.unique(xcc):
    m = xcc == x
    print(f"  {x}: {m.sum():6d} {dur[m].sum() / (128 * WPS):8.2f} {t1[m].max():8.2f} {t0[m].min():8.3f}")
# per SIMD (xcc, se, sh, cu, simd): busy time of its three slots
key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
ks, inv = np.unique(key, return_inverse=True)
busy = np.bincount(inv, weights=dur)
last = np.zeros(len(ks))
np.maximum.at(last, inv, t1)
print(f"SIMDs seen {len(ks)} (expected 1024); busy ms per SIMD / waves per SIMD: min {busy.min() / WPS:.1f} median {np.median(busy) / WPS:.1f} max {busy.max() / WPS:.1f}")
print(f"last end per SIMD: min {last.min():.1f} p10 {np.percentile(last, 10):.1f} median {np.median(last):.1f} max {last.max():.1f}")
# when does the queue run dry?  the latest start of any wave
print(f"latest wave start {t0.max():.2f} ms; waves started in the last 10 % of the span: {(t0 > 0.9 * span / 1e5).sum()}")
# slot idle before the queue ran dry: for each SIMD, waves in flight integrated up to the latest start
tq = t0.max()
inflight_before = np.minimum(t1, tq).sum() - np.minimum(t0, tq).sum()
print(f"mean waves in flight until the last dispatch: {inflight_before / tq:.0f} of {SLOTS}; after it: {(dur.sum() - inflight_before) / (span / 1e5 - tq):.0f}")
# duration of the waves by start time
for a in range(0, int(span / 1e5) + 1, 20):
    m = (t0 >= a) & (t0 < a + 20)
    if m.any():
        print(f"  started in [{a:3d},{a + 20:3d}) ms: {m.sum():5d} waves, mean {dur[m].mean():7.2f} max {dur[m].max():7.2f}  latest end {t1[m].max():7.2f}")
# exact slots: (xcc, se, sh, cu, simd, wave_id); gaps between consecutive waves of a slot
wid = hw & 15
slot = key * 16 + wid
order = np.lexsort((t0, slot))
s_sorted, a0, a1 = slot[order], t0[order], t1[order]
same = s_sorted[1:] == s_sorted[:-1]
gap = (a0[1:] - a1[:-1])[same]
print(f"slots seen {len(np.unique(slot))}; wave ids {np.unique(wid)}; replacements {same.sum()}")
print(f"gap between a wave's end and the next wave's start in the same slot (ms): min {gap.min():.4f} median {np.median(gap):.4f} mean {gap.mean():.4f} p90 {np.percentile(gap, 90):.4f} p99 {np.percentile(gap, 99):.4f} max {gap.max():.4f}; sum {gap.sum():.1f} slot-ms; negative {int((gap < 0).sum())}")
big = gap > 0.05
print(f"gaps > 50 us: {int(big.sum())}, their sum {gap[big].sum():.1f} slot-ms")
prev_end = a1[:-1][same]
for a in range(0, int(span / 1e5) + 1, 20):
    m = (prev_end >= a) & (prev_end < a + 20)
    if m.any():
        print(f"  wave ended in [{a:3d},{a + 20:3d}) ms: {int(m.sum()):5d} replacements, mean gap {gap[m].mean():7.3f} ms, max {gap[m].max():7.3f}")
# slots whose last wave ended early: idle until the end of the launch
last_slot_end = np.zeros(len(np.unique(slot)))
u, inv2 = np.unique(slot, return_inverse=True)
np.maximum.at(last_slot_end, inv2, t1)
print(f"idle after a slot's last wave: {(span / 1e5 - last_slot_end).sum():.0f} slot-ms = {(span / 1e5 - last_slot_end).sum() / SLOTS:.2f} ms per slot; slots never used {SLOTS - len(u)}")
