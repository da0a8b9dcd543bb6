#!/usr/bin/env python3
"""Tile splitting (vpt_capi.hip): one virtual rank of an N-GPU job on this GPU.
  split_calibration.py <nranks> <spp> [rank [scene file [shader [bounces [resolution]]]]]   -> kernel ms, longest wave, slot time of the steady-state launch, state hash
Environment: VPT_SPLIT (0 / 1), VPT_SPLIT_K (force every tile to 2^k waves), VPT_SPLIT_VERBOSE."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vpt_loader

vpt = vpt_loader.load()
nranks, spp = int(sys.argv[1]), int(sys.argv[2])
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
scene_file = sys.argv[4] if len(sys.argv) > 4 else "03_volume/volume.json"
shader = sys.argv[5] if len(sys.argv) > 5 else "volpathtrace"
bounces = int(sys.argv[6]) if len(sys.argv) > 6 else 64
res = int(sys.argv[7]) if len(sys.argv) > 7 else 1280
scene = vpt.HostScene(os.path.join(ROOT, "tests", "golden", "scenes", scene_file))
dev = vpt.DeviceScene(scene, 0)
p = vpt.PathtraceParams(resolution=res, samples=1 << 20, shader=shader, bounces=bounces)
host = scene.make_state(p)
lay = vpt.VptLayout(host.width, host.height, 8, 8, rank, nranks)
slots = vpt.layout_slots(lay)
d = torch.device("cuda", 0)
img = torch.zeros((slots, 4), dtype=torch.float32, device=d)
hit = torch.zeros((slots,), dtype=torch.int32, device=d)
rng = torch.zeros((slots, 2), dtype=torch.int64, device=d)
vpt.state_upload(lay, host, img.data_ptr(), hit.data_ptr(), rng.data_ptr())
ms = []
for _ in range(6):
    dev.render_device(p, lay, spp, img.data_ptr(), hit.data_ptr(), rng.data_ptr(), 0)
    torch.cuda.synchronize()
    ms.append(dev.last_kernel_ms())
costs = dev.last_wave_costs().astype(np.float64) * 1e-5
h = hashlib.sha1(img.cpu().numpy().tobytes() + rng.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"rank {rank} of {nranks}, {spp} spp: kernel ms {' '.join(f'{m:.1f}' for m in ms)} | waves {len(costs)} longest {costs.max():.1f} ms, slot time {costs.sum():.0f} ms "
      f"= {costs.sum() / 3072:.1f} ms per slot | state {h}")
