import os, sys, heapq
import numpy as np, torch
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import vpt_loader
vpt = vpt_loader.load()
scene = vpt.HostScene(os.path.join(ROOT,'tests/golden/scenes/06_gridsdf_full/gridsdf_full.json'))
dev = vpt.DeviceScene(scene, 0)
p = vpt.PathtraceParams(resolution=1280, samples=1<<20, shader='implicit', bounces=4)
host = scene.make_state(p)
lay = vpt.VptLayout(host.width, host.height, 8, 8, 0, 1)
slots = vpt.layout_slots(lay)
d = torch.device('cuda',0)
img = torch.zeros((slots,4),dtype=torch.float32,device=d); hit=torch.zeros((slots,),dtype=torch.int32,device=d); rng=torch.zeros((slots,2),dtype=torch.int64,device=d)
vpt.state_upload(lay, host, img.data_ptr(), hit.data_ptr(), rng.data_ptr())
for _ in range(4):
    dev.render_device(p, lay, 128, img.data_ptr(), hit.data_ptr(), rng.data_ptr(), 0); torch.cuda.synchronize()
ms = dev.last_kernel_ms()
c = dev.last_wave_costs().astype(np.float64)*1e-5
c = c[c>0]
def lpt(costs, m):
    costs = sorted(costs, reverse=True)
    h=[0.0]*m; heapq.heapify(h); span=0
    for x in costs:
        t=heapq.heappop(h)+x; span=max(span,t); heapq.heappush(h,t)
    return span
print('kernel ms', ms, 'waves', len(c), 'longest', c.max(), 'sum/5120', c.sum()/5120)
for m in (5120, 4608, 4096):
    print('ideal LPT on', m, 'slots:', lpt(c, m))
# per XCD: every 8th wave in sorted order on 640 slots
cs = np.sort(c)[::-1]
print('per-XCD LPT (640 slots, every 8th of the sorted list):', max(lpt(cs[k::8], 640) for k in range(8)))
print('quantiles', np.quantile(c,[0.01,0.1,0.25,0.5,0.75,0.9,0.99]))

# ---- what would cutting every tile at a sample boundary give?  Two pieces per tile (each half the samples, the second may only start when the first has ended;
# a piece dispatched before that spins in its slot), list scheduling in a given order on m slots
def simulate(first, second, order2, m):
    """first[i], second[i]: durations; dispatch: all first pieces longest first, then second pieces in order2; returns (span, spin slot-ms)"""
    n = len(first)
    o1 = np.argsort(-first)
    free = [0.0] * m
    heapq.heapify(free)
    end1 = np.zeros(n)
    t_dispatch = 0.0
    for i in o1:
        t = heapq.heappop(free)
        t_dispatch = max(t_dispatch, t)      # in-order dispatch
        end1[i] = t_dispatch + first[i]
        heapq.heappush(free, end1[i])
    span, spin = end1.max(), 0.0
    for i in order2:
        t = heapq.heappop(free)
        t_dispatch = max(t_dispatch, t)
        start = max(t_dispatch, end1[i])
        spin += start - t_dispatch
        e = start + second[i]
        span = max(span, e)
        heapq.heappush(free, e)
    return span, spin
half = c / 2
for name, o2 in (("second pieces longest first", np.argsort(-half)), ("second pieces by their first piece's cost, ascending", np.argsort(half)),
                 ("second pieces in the first pieces' dispatch order", np.argsort(-half))):
    s, sp = simulate(half, half, o2, 5120)
    print(f"two pieces per tile, {name}: span {s:.1f} ms, spinning {sp / 5120:.2f} ms per slot")
# K1-like check of the same idea is in the caller's hands: pass costs of a K1 launch

# ---- how repeatable is a wave's duration from one launch to the next (the order of launch n + 1 is the sorted cost of launch n)?
prev = dev.last_wave_costs().astype(np.float64) * 1e-5
dev.render_device(p, lay, 128, img.data_ptr(), hit.data_ptr(), rng.data_ptr(), 0); torch.cuda.synchronize()
cur = dev.last_wave_costs().astype(np.float64) * 1e-5
ok = (prev > 0) & (cur > 0)
rel = (cur[ok] - prev[ok]) / prev[ok]
print(f"launch to launch: relative change of a wave's duration: mean {rel.mean():+.4f}, std {rel.std():.4f}, |.| p50 {np.percentile(abs(rel), 50):.4f} p90 {np.percentile(abs(rel), 90):.4f} p99 {np.percentile(abs(rel), 99):.4f} max {abs(rel).max():.3f}")
big = prev[ok] > 60
print(f"  waves above 60 ms: std {rel[big].std():.4f}, p99 {np.percentile(abs(rel[big]), 99):.4f}, max {abs(rel[big]).max():.3f}")
# LPT with the PREVIOUS launch's order but THIS launch's durations
o = np.argsort(-prev[ok])
h = [0.0] * 5120; heapq.heapify(h); span = 0
for i in o:
    t = heapq.heappop(h) + cur[ok][i]; span = max(span, t); heapq.heappush(h, t)
print(f"list scheduling in the previous launch's order with this launch's durations: {span:.1f} ms (kernel {dev.last_kernel_ms():.1f} ms)")
