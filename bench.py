#!/usr/bin/env python3
"""bench.py — Msamples/s of the HIP integrator on BASELINE.json's config[1]:
tests/03_volume, --shader volpathtrace --bounces 64, 1280 wide (x533: camera aspect 2.4), 256 spp.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A *step* is one launch of the hot path over the whole frame: `--spp` (256) samples for every
pixel, with the pixel state (radiance sums, hit counts, PCG32 streams) already resident in HBM in
the tile-major layout of include/vpt.h.  Launches start their waves longest first, using the per-wave
durations the previous launch on the same layout recorded; the very first launch of a
process measures them with a short pilot, which therefore falls into the warm-up — the line's `cold` record
times exactly that first call on a fresh scene handle (N = 1).

N GPUs: the frame is FIXED (1280 wide unless --resolution says otherwise: BASELINE config[4] is --resolution 3840)
and cut into 8x8-pixel tiles dealt round-robin to the ranks (tile t -> rank t % N); each step ends with an RCCL
all_gather of the ranks' tile buffers over xGMI and a resolve kernel on every rank.  `scaling` is
therefore "strong", and after the timed region rank 0 renders the same frame alone (the other ranks wait) so that the
line carries `speedup_vs_1gpu` measured in the same job.  `--weak` instead grows the frame with N (width x sqrt(N)).

The JSON line carries, besides the driver's contract:
  roofline      algorithmic bytes per launch (SURVEY §8(d) formula, event counts measured by the CPU
                oracle on a bounded sample of the same workload) / mean kernel time from HIP events
                on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline  the reference's own renderer (oracle/_ref/ref_driver, "reference") or, where that
                binary is absent, our CPU restatement ("port"), timed on this host's cores on a
                bounded sample (default 640x267x96 spp, ~10 s) of the same workload.  Rank 0, N=1 only.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENE = os.path.join(ROOT, "tests", "golden", "scenes", "03_volume", "volume.json")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_sample(c):
    """B = 32*(scene+shape nodes) + 64*instance tests + 4*(instance+prim tests)
    + P*prim tests + 16*f32 texels + 4*u8 texels + 4*cdf probes + V*surface hits + 72, plus 4 B per voxel the
    trilinear SDF lookup fetches (the implicit shaders' only per-lane gather).  Analytic SDF evaluations read
    wave-uniform records (scalar loads) and are charged nothing."""
    n = float(c["samples"])
    prim = c["quad_tests"] + c["tri_tests"]
    b = (32.0 * (c["scene_nodes"] + c["shape_nodes"]) + 64.0 * c["instance_tests"]
         + 4.0 * (c["instance_tests"] + prim) + 64.0 * c["quad_tests"] + 48.0 * c["tri_tests"]
         + 16.0 * c["texel_f32"] + 4.0 * c["texel_u8"] + 4.0 * c["cdf_probes"] + 80.0 * c["surface_hits"]
         + 4.0 * c["voxel_fetches"]) / n + 72.0
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per step")
    ap.add_argument("--resolution", type=int, default=1280)
    ap.add_argument("--bounces", type=int, default=64)
    ap.add_argument("--shader", default="volpathtrace")
    ap.add_argument("--scene", default=SCENE)
    ap.add_argument("--weak", action="store_true", help="grow the frame with N (width x sqrt(N)) instead of the fixed frame")
    ap.add_argument("--strong", action="store_true", help="(default, kept for older command lines) fixed frame for every N")
    ap.add_argument("--cpu-sample", default="640x96", help="cpu baseline sample: <resolution>x<spp>; '0' disables")
    ap.add_argument("--tile", type=int, default=8)
    ap.add_argument("--balance", action="store_true", help="add per-wave duration statistics of the last launch to the line")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold first-call measurement (N = 1)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import vpt_loader
    vpt = vpt_loader.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # VPT_BENCH_REHEARSAL=1: all ranks of an N > 1 job on GPU 0 with gloo (tile buffers staged through the host) - a dry
    # run of the multi-GPU code path on a one-GPU box; its timings mean nothing and the line says so
    rehearsal = world > 1 and os.environ.get("VPT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    resolution = int(round(args.resolution * (world ** 0.5))) if args.weak else args.resolution
    scene = vpt.HostScene(args.scene)
    # params.samples bounds the progressive render; keep it out of reach (and != 1: preview branch)
    params = vpt.PathtraceParams(resolution=resolution, samples=1 << 30, shader=args.shader, bounces=args.bounces)
    state = scene.make_state(params)
    width, height = state.width, state.height
    dev = vpt.DeviceScene(scene, local_rank)
    layout = vpt.VptLayout(width, height, args.tile, args.tile, rank, world)
    slots = vpt.layout_slots(layout)
    device = torch.device("cuda", local_rank)
    d_image = torch.zeros((slots, 4), dtype=torch.float32, device=device)
    d_hits = torch.zeros((slots,), dtype=torch.int32, device=device)
    d_rng = torch.zeros((slots, 2), dtype=torch.int64, device=device)
    vpt.state_upload(layout, state, d_image.data_ptr(), d_hits.data_ptr(), d_rng.data_ptr())
    gathered = torch.empty((world * slots, 4), dtype=torch.float32, device=device) if world > 1 else d_image
    frame = torch.empty((height, width, 4), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    done = [0]
    kernel_ms = []

    def step(record):
        dev.render_device(params, layout, args.spp, d_image.data_ptr(), d_hits.data_ptr(), d_rng.data_ptr(), stream)
        done[0] += args.spp
        if record:
            kernel_ms.append(dev.last_kernel_ms())
        if rehearsal:
            torch.cuda.synchronize()
            parts = [torch.empty((slots, 4), dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, d_image.cpu())
            gathered.copy_(torch.cat(parts, 0))
        elif world > 1:  # tile buffers of all ranks over xGMI, then de-interleave on every rank
            dist.all_gather_into_tensor(gathered, d_image)
        vpt.resolve_device(layout, gathered.data_ptr(), done[0], frame.data_ptr(), stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    balance = None
    if args.balance:   # load-balance figures of the last launch (after the timed region): per-wave durations
        c = dev.last_wave_costs().astype(np.float64) * 1e-5   # ticks of 100 MHz -> ms
        c = c[c > 0]
        balance = {"waves": int(len(c)), "longest_wave_ms": round(float(c.max()), 3), "mean_wave_ms": round(float(c.mean()), 4),
                   "sum_wave_ms": round(float(c.sum()), 1), "p99_wave_ms": round(float(np.quantile(c, 0.99)), 3)}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1: the same frame on rank 0 alone, for the speed-up (the other ranks wait at the barrier) ----------
    single = None
    if world > 1 and not args.weak:
        if rank == 0:
            lay1 = vpt.VptLayout(width, height, args.tile, args.tile, 0, 1)
            n1 = vpt.layout_slots(lay1)
            i1 = torch.zeros((n1, 4), dtype=torch.float32, device=device)
            h1 = torch.zeros((n1,), dtype=torch.int32, device=device)
            r1 = torch.zeros((n1, 2), dtype=torch.int64, device=device)
            vpt.state_upload(lay1, state, i1.data_ptr(), h1.data_ptr(), r1.data_ptr())
            solo_steps = max(1, min(args.steps, 2))
            for k in range(1 + solo_steps):   # one warm-up launch (pilot + order), then the timed ones
                if k == 1:
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                dev.render_device(params, lay1, args.spp, i1.data_ptr(), h1.data_ptr(), r1.data_ptr(), stream)
                vpt.resolve_device(lay1, i1.data_ptr(), (k + 1) * args.spp, frame.data_ptr(), stream)
            torch.cuda.synchronize()
            single = {"ms_per_step": round((time.perf_counter() - ts) / solo_steps * 1e3, 3), "steps": solo_steps}
            del i1, h1, r1
        dist.barrier()
    # ---- N = 1: the first call on a fresh scene handle (no wave costs known: pilot launch + ordered launch) ---------
    cold = None
    if world == 1 and not args.no_cold:
        dev2 = vpt.DeviceScene(scene, local_rank)
        i2, h2, r2 = torch.zeros_like(d_image), torch.zeros_like(d_hits), torch.zeros_like(d_rng)
        vpt.state_upload(layout, state, i2.data_ptr(), h2.data_ptr(), r2.data_ptr())
        torch.cuda.synchronize()
        tc = time.perf_counter()
        dev2.render_device(params, layout, args.spp, i2.data_ptr(), h2.data_ptr(), r2.data_ptr(), stream)
        vpt.resolve_device(layout, i2.data_ptr(), args.spp, frame.data_ptr(), stream)
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - tc) * 1e3
        cold = {"ms": round(cold_ms, 3), "value": round(width * height * args.spp / cold_ms * 1e-3, 3), "unit": "Msamples/s",
                "what": "first call on a fresh scene handle: no wave costs known yet, so a pilot launch (spp/64 samples) measures "
                        "them and the rest runs in its order; `value` above is the steady state (order from the previous launch)"}
        del dev2, i2, h2, r2
    scene_name = os.path.basename(os.path.dirname(os.path.abspath(args.scene)))
    data_desc = ("reference scene tests/03_volume (real assets), deterministic PCG32 seeds" if args.scene == SCENE else
                 f"{scene_name}: substitute assets (tests/golden/make_scenes.py), deterministic PCG32 seeds")
    samples_per_step = width * height * args.spp
    value = samples_per_step * args.steps / elapsed * 1e-6

    roofline, cpu_baseline = None, None
    if rank == 0:
        import oracle_lib  # the checker, used here only for the reported CPU baseline and event counts
        # --- event counts on a bounded sample of the same workload (oracle, all host threads) ------
        cres, cspp = 320, 4
        cparams = vpt.PathtraceParams(resolution=cres, samples=1 << 30, shader=args.shader, bounces=args.bounces)
        cstate = scene.make_state(cparams)
        counters = oracle_lib.oracle_render(scene, cparams, cstate, cspp, nthreads=0, counters=True)
        bps = algorithmic_bytes_per_sample(counters)
        # this rank's share of a launch
        per_launch_samples = samples_per_step / world
        mean_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        achieved = bps * per_launch_samples / (mean_ms * 1e-3) * 1e-9
        # memory-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs of
        # this same command, profiles/tools/profile_config.sh); only quoted for the workload it was measured on
        traffic = None
        tag = {("03_volume", "volpathtrace", 64): "k1", ("05_head1ss_sub", "volpathtrace", 64): "head",
               ("06_gridsdf_synth", "implicit", 4): "k2"}.get((scene_name, args.shader, args.bounces))

        def _version(path):   # profiles/r<round>_<tag>_v<version>_hbm_traffic.json (round 1: r01_v<version>_...), compared numerically
            import re
            m = re.search(r"r(\d+)_(?:[a-z0-9]+_)?v(\d+)_", os.path.basename(path))
            return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_v*_hbm_traffic.json")) +
                        (glob.glob(os.path.join(ROOT, "profiles", "r01_v*_hbm_traffic.json")) if tag == "k1" else []), key=_version) if tag else []
        if tfiles:
            t = json.load(open(tfiles[-1]))
            if t.get("bytes_per_sample"):
                traffic = round(t["bytes_per_sample"] * per_launch_samples)
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel": ("vpt_render_kernel<%s>" if args.shader.startswith("implicit") else "vpt_mesh_kernel<%s>") % args.shader, "kernel_ms": round(mean_ms, 3),
                    "algorithmic_bytes_per_sample": round(bps, 1)}
        if world == 1 and args.cpu_sample != "0":
            sres, sspp = (int(x) for x in args.cpu_sample.split("x"))
            # cores this process may actually use (cgroup / affinity), not the machine's thread count
            ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:  # cgroup v2 CPU quota ("<quota> <period>" or "max <period>")
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if quota != "max":
                    ncores = max(1, min(ncores, int(int(quota) / int(period))))
            except (OSError, ValueError):
                pass
            if oracle_lib.have_reference():
                *_, info = oracle_lib.reference_render(args.scene, args.shader, sres, sspp, args.bounces,
                                                       workdir=os.environ.get("TMPDIR", "/tmp"))
                cpu_baseline = {"value": round(info["msamples_per_s"], 4), "unit": "Msamples/s", "cores": min(ncores, info["threads"]),
                                "threads_started": info["threads"],
                                "kind": "reference",
                                "sample": f"{scene_name} {info['width']}x{info['height']}x{sspp}spp, reference renderer (g++ -O2)"}
            else:
                sp = vpt.PathtraceParams(resolution=sres, samples=1 << 30, shader=args.shader, bounces=args.bounces)
                sstate = scene.make_state(sp)
                t1 = time.perf_counter()
                oracle_lib.oracle_render(scene, sp, sstate, sspp, nthreads=0)
                dt = time.perf_counter() - t1
                cpu_baseline = {"value": round(sstate.width * sstate.height * sspp / dt * 1e-6, 4), "unit": "Msamples/s",
                                "cores": ncores, "kind": "port",
                                "sample": f"{scene_name} {sstate.width}x{sstate.height}x{sspp}spp, oracle/vpt_oracle.cpp"}
        line = {
            "metric": "Msamples/sec", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": data_desc + (" [REHEARSAL: all ranks on one GPU over gloo - timings are not a measurement]" if rehearsal else ""),
            "config": {"workload": f"{scene_name} {args.shader} bounces={args.bounces} {width}x{height}x{args.spp}spp per step" + ("" if world == 1 else (", frame grows with N" if args.weak else ", fixed frame")),
                       "tile": f"{args.tile}x{args.tile}", "parallelism": f"tiles%{world}, costly tiles split into partly filled waves (auto)" if world > 1 else "1gpu",
                       "samples_per_step": samples_per_step},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if balance:
            line["balance"] = balance
        if cold:
            line["cold"] = cold
        if single:
            line["single_gpu"] = single
            line["speedup_vs_1gpu"] = round(single["ms_per_step"] / (elapsed / args.steps * 1e3), 3)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
