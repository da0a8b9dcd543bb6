#!/usr/bin/env python3
"""Where the lanes of K2's waves are, trip by trip (diagnostic build -DVPT_K2_STATS of libvpt_hip.so):
    VPT_HIP_LIB=variants/libvpt_hip_k2stats.so python profiles/tools/k2_stats.py [scene.json] [spp]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vpt_loader  # noqa: E402

NAMES = ("trips scene_rounds scene_lanes light_rounds light_lanes shade_rounds shade_lanes done_lanes wait_lanes_at_march "
         "light_lanes_at_scene scene_lanes_at_shade clk_scene clk_light clk_shade clk_total scene_le8 scene_le16 scene_le32 scene_lanes_le16 scene_lanes_le32").split()


def main():
    vpt = vpt_loader.load()
    scene_file = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests/golden/scenes/06_gridsdf_full/gridsdf_full.json")
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    scene = vpt.HostScene(scene_file)
    dev = vpt.DeviceScene(scene, 0)
    p = vpt.PathtraceParams(resolution=1280, samples=1 << 20, shader="implicit", bounces=4)
    st = scene.make_state(p)
    dev.pathtrace_samples(st, p, spp)          # first launch: tile order
    out = (C.c_ulonglong * 24)()
    vpt.hip.vpt_debug_k2_stats(out, 1)
    dev.pathtrace_samples(st, p, spp)          # second launch: longest wave first
    vpt.hip.vpt_debug_k2_stats(out, 0)
    v = dict(zip(NAMES, list(out)))
    print({k: x for k, x in v.items()})
    print(f"lanes per scene round {v['scene_lanes'] / max(1, v['scene_rounds']):.1f}, per light round {v['light_lanes'] / max(1, v['light_rounds']):.1f}, "
          f"per shading round {v['shade_lanes'] / max(1, v['shade_rounds']):.1f}")
    print(f"per trip: done lanes {v['done_lanes'] / v['trips']:.1f}, waiting at march trips {v['wait_lanes_at_march'] / max(1, v['trips'] - v['shade_rounds']):.1f}, "
          f"light lanes during scene rounds {v['light_lanes_at_scene'] / max(1, v['scene_rounds']):.1f}, marching lanes parked during shading {v['scene_lanes_at_shade'] / max(1, v['shade_rounds']):.1f}")
    tot = max(1, v["clk_total"])
    print(f"wave time (shader clock): scene-march rounds {v['clk_scene'] / tot:.3f}, light-march rounds {v['clk_light'] / tot:.3f}, shading block {v['clk_shade'] / tot:.3f}, "
          f"rest (prologue, state I/O, loop head) {1 - (v['clk_scene'] + v['clk_light'] + v['clk_shade']) / tot:.3f}")
    samples = st.width * st.height * spp
    print(f"per sample: scene rounds {v['scene_rounds'] * 8 / samples * 64:.1f} wave-steps x 64 lanes offered, {v['scene_lanes'] * 8 / samples:.1f} lane-steps used (upper bound: rounds hold up to 8 steps)")
    sr = max(1, v["scene_rounds"])
    print(f"scene rounds by marching lanes: <= 8: {v['scene_le8'] / sr:.3f}, <= 16: {v['scene_le16'] / sr:.3f} (holding {v['scene_lanes_le16'] / max(1, v['scene_lanes']):.3f} of the lane-steps), "
          f"<= 32: {v['scene_le32'] / sr:.3f} ({v['scene_lanes_le32'] / max(1, v['scene_lanes']):.3f} of the lane-steps)")
    print(f"rounds: trips {v['trips']}, scene {v['scene_rounds']}, light {v['light_rounds']}, shade {v['shade_rounds']}")


if __name__ == "__main__":
    main()
