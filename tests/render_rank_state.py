"""helper of tests/test_tile_splitting.py: render one virtual rank of an N-rank job on GPU 0 through the device-state API,
several calls in a row, and save the rank's state (row-major, only its own pixels are touched) plus what the launches did.
usage: render_rank_state.py <scene.json> <resolution> <nranks> <rank> <spp per call> <calls> <out.npz> [shader [bounces]]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vpt_loader

vpt = vpt_loader.load()
scene_file, res, nranks, rank, spp, calls, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
scene = vpt.HostScene(scene_file)
dev = vpt.DeviceScene(scene, 0)
shader = sys.argv[8] if len(sys.argv) > 8 else "volpathtrace"
bounces = int(sys.argv[9]) if len(sys.argv) > 9 else 64
p = vpt.PathtraceParams(resolution=res, samples=1 << 20, shader=shader, bounces=bounces)
host = scene.make_state(p)
lay = vpt.VptLayout(host.width, host.height, 8, 8, rank, nranks)
slots = vpt.layout_slots(lay)
d = torch.device("cuda", 0)
img = torch.zeros((slots, 4), dtype=torch.float32, device=d)
hit = torch.zeros((slots,), dtype=torch.int32, device=d)
rng = torch.zeros((slots, 2), dtype=torch.int64, device=d)
vpt.state_upload(lay, host, img.data_ptr(), hit.data_ptr(), rng.data_ptr())
waves, ms = [], []
for _ in range(calls):
    dev.render_device(p, lay, spp, img.data_ptr(), hit.data_ptr(), rng.data_ptr(), 0)
    torch.cuda.synchronize()
    ms.append(dev.last_kernel_ms())
    waves.append(len(dev.last_wave_costs()))
back = host.copy()
vpt.state_download(lay, img.data_ptr(), hit.data_ptr(), rng.data_ptr(), back)
np.savez(out, image=back.image, hits=back.hits, rngs=back.rngs, waves=np.array(waves), ms=np.array(ms), tiles=slots // 64)
