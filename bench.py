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

N GPUs: the frame is FIXED (1280 wide unless --resolution says otherwise) and cut into 8x8-pixel tiles dealt
round-robin to the ranks (tile t -> rank t % N); each step ends with an RCCL all_gather of the ranks' tile buffers
over xGMI and a resolve kernel on every rank.  `scaling` is therefore "strong", and after the timed
region rank 0 renders the same frame alone (the other ranks wait) so that the line carries `speedup_vs_1gpu` measured in
the same job.  A pixel's samples are a serial chain of RNG draws, so a launch is never shorter than its costliest
tile: on the 1280x533 headline frame that chain bounds the speed-up to about 2.4x at 8 GPUs, which is why
the N > 1 line also carries `config5`: BASELINE config[4]'s 3840x1600 frame (9x the tiles, the same chain), timed the
same way with its own `single_gpu` and `speedup_vs_1gpu` — the frame north_star's ">= 6x at 8 GPUs" is about.
`--weak` instead grows the frame with N (width x sqrt(N)).  `--dist` runs the N > 1 code path (nccl process group,
all_gather_into_tensor, resolve of the gathered buffer) at world size 1, so that it executes on a one-GPU box.

The JSON line carries, besides the driver's contract:
  roofline       algorithmic bytes per launch (SURVEY §8(d) formula, event counts measured by the CPU
                 oracle on a bounded sample of the same workload) / mean kernel time from HIP events
                 on the launch stream, against the 8 TB/s HBM peak; `traffic` is the memory-side figure of this launch: in the default
                 run it is measured live - bench.py starts itself twice under rocprofv3 (--pmc FETCH_SIZE, then --pmc WRITE_SIZE, one
                 launch each) after the timed region; otherwise (other workloads, no rocprofv3) the committed passes named by
                 `traffic_source` are scaled to this launch.  `traffic_committed` keeps that replay beside the live figure.
  cpu_baseline   the reference's own renderer (oracle/_ref/ref_driver, "reference": all hardware threads as the
                 reference starts them, plus `single_thread`) and our CPU restatement at one thread per core (`port`),
                 timed on this host's cores on bounded samples of the same workload.  Rank 0, N=1 only.
  rccl_path_one_rank  (N = 1) the collective path of --gpus N run with one rank in a child process (`--dist`): proof in the line itself
                 that the RCCL calls of the one-process-per-GPU front end execute on the box the line was measured on.
  other_configs  (N = 1) the other BASELINE configs' workloads, a few steps each after the timed region: config 3's
                 substitute scene (05_head1ss_sub volpathtrace), config 4's (06_gridsdf_synth implicit), config 5's frame
                 (03_volume 3840x1600), each with its own cold first call and roofline record.
"""
import argparse
import glob
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
SCENE = os.path.join(SCENES, "03_volume", "volume.json")
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_ISSUE_PEAK = 1228.8e9   # wave64 VALU instructions / s: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles (MI355X_MICROARCH.md)


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


def host_cores():
    """cores this process may actually use (cgroup quota / affinity), not the machine's thread count"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota ("<quota> <period>" or "max <period>")
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def profile_file(tag, kind):
    """newest profiles/r<round>_<tag>_v<version>_<kind>.json (round 1: r01_v<version>_...), compared numerically"""
    def version(path):
        m = re.search(r"r(\d+)_(?:[a-z0-9]+_)?v(\d+)_", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    files = glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_v*_{kind}.json"))
    if tag == "k1":
        files += glob.glob(os.path.join(ROOT, "profiles", f"r01_v*_{kind}.json"))
    return sorted(files, key=version)[-1] if files else None


class Bench:
    def __init__(self, args):
        import numpy as np
        import torch
        import vpt_loader
        self.np, self.torch, self.args = np, torch, args
        self.vpt = vpt_loader.load()
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            if self.world == 1 and args.gpus > 1:
                sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
            args.gpus = self.world
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
        # VPT_BENCH_REHEARSAL=1: all ranks of an N > 1 job on GPU 0 with gloo (tile buffers staged through the host) - a dry
        # run of the multi-GPU code path on a one-GPU box; its timings mean nothing and the line says so
        self.rehearsal = self.world > 1 and os.environ.get("VPT_BENCH_REHEARSAL") == "1"
        if self.rehearsal:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.dist = None
        self.collective = self.world > 1 or args.dist   # steps end with the gather of the ranks' tile buffers
        if self.collective:
            import torch.distributed as dist
            self.dist = dist
            if self.rehearsal:
                dist.init_process_group("gloo")
            elif self.world == 1:   # --dist: the N > 1 code path at world size 1 (RCCL communicator of one rank)
                import tempfile
                rendezvous = os.path.join(tempfile.gettempdir(), f"vpt_bench_rendezvous_{os.getpid()}")   # a file: no port to collide on
                if os.path.exists(rendezvous):   # a leftover of an earlier process with this pid would be read as that job's store
                    os.remove(rendezvous)
                dist.init_process_group("nccl", init_method=f"file://{rendezvous}", rank=0, world_size=1, device_id=self.device)
            else:
                dist.init_process_group("nccl", device_id=self.device)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.scenes = {}

    # ---- scenes and workloads ------------------------------------------------------------------------------------
    def host_scene(self, path):
        if path not in self.scenes:
            self.scenes[path] = self.vpt.HostScene(path)
        return self.scenes[path]

    def workload(self, scene_path, shader, bounces, resolution, spp, rank=None, world=None):
        """state of `rank`'s tiles of the frame, resident on this process's GPU, under a FRESH scene handle"""
        vpt, torch = self.vpt, self.torch
        rank = self.rank if rank is None else rank
        world = self.world if world is None else world
        w = argparse.Namespace(scene_path=scene_path, shader=shader, bounces=bounces, spp=spp, rank=rank, world=world, done=0, kernel_ms=[])
        w.scene = self.host_scene(scene_path)
        w.scene_name = os.path.basename(os.path.dirname(os.path.abspath(scene_path)))
        # params.samples bounds the progressive render; keep it out of reach (and != 1: preview branch)
        w.params = vpt.PathtraceParams(resolution=resolution, samples=1 << 30, shader=shader, bounces=bounces)
        w.state = w.scene.make_state(w.params)
        w.width, w.height = w.state.width, w.state.height
        w.dev = vpt.DeviceScene(w.scene, self.local_rank)
        w.layout = vpt.VptLayout(w.width, w.height, self.args.tile, self.args.tile, rank, world)
        w.slots = vpt.layout_slots(w.layout)
        w.d_image = torch.zeros((w.slots, 4), dtype=torch.float32, device=self.device)
        w.d_hits = torch.zeros((w.slots,), dtype=torch.int32, device=self.device)
        w.d_rng = torch.zeros((w.slots, 2), dtype=torch.int64, device=self.device)
        vpt.state_upload(w.layout, w.state, w.d_image.data_ptr(), w.d_hits.data_ptr(), w.d_rng.data_ptr())
        gather = self.collective and world == self.world
        w.gathered = torch.empty((world * w.slots, 4), dtype=torch.float32, device=self.device) if gather else w.d_image
        w.frame = torch.empty((w.height, w.width, 4), dtype=torch.float32, device=self.device)
        w.samples_per_step = w.width * w.height * spp
        w.gather = gather
        return w

    def step(self, w, record=True):
        vpt, torch, dist = self.vpt, self.torch, self.dist
        w.dev.render_device(w.params, w.layout, w.spp, w.d_image.data_ptr(), w.d_hits.data_ptr(), w.d_rng.data_ptr(), self.stream)
        w.done += w.spp
        if record:
            w.kernel_ms.append(w.dev.last_kernel_ms())
        if w.gather and self.rehearsal:
            torch.cuda.synchronize()
            parts = [torch.empty((w.slots, 4), dtype=torch.float32) for _ in range(w.world)]
            dist.all_gather(parts, w.d_image.cpu())
            w.gathered.copy_(torch.cat(parts, 0))
        elif w.gather:  # tile buffers of all ranks over xGMI, then de-interleave on every rank
            dist.all_gather_into_tensor(w.gathered, w.d_image)
        vpt.resolve_device(w.layout, w.gathered.data_ptr(), w.done, w.frame.data_ptr(), self.stream)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, w, steps, warmup):
        """W untimed steps, then exactly `steps` steps between two fences; seconds, max over ranks"""
        for _ in range(warmup):
            self.step(w, False)
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(w, True)
        self.fence()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device="cpu" if self.rehearsal else self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def cold_call(self, w):
        """the first call on w's (fresh) scene handle: no wave costs known yet"""
        self.torch.cuda.synchronize()
        tc = time.perf_counter()
        self.step(w, False)
        self.torch.cuda.synchronize()
        ms = (time.perf_counter() - tc) * 1e3
        return {"ms": round(ms, 3), "value": round(w.samples_per_step / ms * 1e-3, 3), "unit": "Msamples/s"}

    def alone_on_rank0(self, scene_path, shader, bounces, resolution, spp, steps):
        """N > 1: the same frame on rank 0 alone, for the speed-up (the other ranks wait at the barrier)"""
        single = None
        if self.rank == 0:
            w1 = self.workload(scene_path, shader, bounces, resolution, spp, rank=0, world=1)
            self.step(w1, False)   # warm-up launch (pilot + order)
            self.torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(steps):
                self.step(w1, False)
            self.torch.cuda.synchronize()
            single = {"ms_per_step": round((time.perf_counter() - ts) / steps * 1e3, 3), "steps": steps}
            del w1
        self.dist.barrier()
        return single

    # ---- derived records (rank 0) ----------------------------------------------------------------------------------
    def roofline(self, w):
        import oracle_lib  # the checker, used here only for the reported CPU baseline and event counts
        vpt = self.vpt
        # event counts on a bounded sample of the same workload (oracle, all host threads)
        cparams = vpt.PathtraceParams(resolution=320, samples=1 << 30, shader=w.shader, bounces=w.bounces)
        cstate = w.scene.make_state(cparams)
        counters = oracle_lib.oracle_render(w.scene, cparams, cstate, 4, nthreads=0, counters=True)
        bps = algorithmic_bytes_per_sample(counters)
        per_launch_samples = w.samples_per_step / w.world   # this rank's share of a launch
        mean_ms = sum(w.kernel_ms) / max(1, len(w.kernel_ms))
        achieved = bps * per_launch_samples / (mean_ms * 1e-3) * 1e-9
        implicit = w.shader.startswith("implicit")
        rec = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
               "traffic": None, "traffic_source": None,
               "kernel": ("vpt_render_kernel<%s>" if implicit else "vpt_mesh_kernel<%s>") % w.shader, "kernel_ms": round(mean_ms, 3),
               "algorithmic_bytes_per_sample": round(bps, 1)}
        # memory-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs of
        # this same command, profiles/tools/profile_config.sh); only quoted for the workload it was measured on
        tag = {("03_volume", "volpathtrace", 64): "k1", ("05_head1ss_sub", "volpathtrace", 64): "head",
               ("06_gridsdf_full", "implicit", 4): "k2"}.get((w.scene_name, w.shader, w.bounces))
        tfile = profile_file(tag, "hbm_traffic") if tag else None
        if tfile:
            t = json.load(open(tfile))
            if t.get("bytes_per_sample"):
                rec["traffic"] = round(t["bytes_per_sample"] * per_launch_samples)
                rec["traffic_source"] = (os.path.relpath(tfile, ROOT) + " (committed rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this kernel version, bytes per sample x this launch's "
                                         "samples; the write half is almost entirely register-spill scratch cycling through L2, not data: "
                                         f"{t.get('WRITE_SIZE_KB', 0) * 1024.0 / max(1, t.get('samples_per_launch', 1)):.0f} of {t['bytes_per_sample']:.0f} B per sample)")
        if implicit:
            # K2 is VALU-bound by a wide margin: its record is priced against the chip's VALU issue peak - wave-level VALU
            # instructions per sample from the committed SQ pass of this kernel x live samples/s / 1 228.8 G wave-instructions/s - and the
            # HBM figure on algorithmic bytes becomes the secondary key
            pfile = profile_file(tag, "pmc_summary") if tag else None
            derived = json.load(open(pfile)).get("_derived", {}) if pfile else {}
            per_sample = derived.get("valu_wave_instructions_per_sample")
            if per_sample:
                issued = per_sample * per_launch_samples / (mean_ms * 1e-3)
                rec = {"bound": "valu", "achieved": round(issued * 1e-9, 2), "peak": round(VALU_ISSUE_PEAK * 1e-9, 1), "unit": "G wave64 VALU instructions/s",
                       "frac": round(issued / VALU_ISSUE_PEAK, 4),
                       "lane_utilisation": round(derived["valu_lane_utilisation"], 4) if derived.get("valu_lane_utilisation") else None,
                       "valu_source": os.path.relpath(pfile, ROOT) + f" ({per_sample:.0f} wave-level VALU instructions per sample, lane utilisation: committed rocprofv3 SQ pass) "
                                      "x live samples/s / (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction)",
                       "traffic": rec["traffic"], "traffic_source": rec["traffic_source"], "kernel": rec["kernel"], "kernel_ms": rec["kernel_ms"],
                       "algorithmic_bytes_per_sample": rec["algorithmic_bytes_per_sample"],
                       "hbm": {"achieved": rec["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rec["frac"]}}
        return rec

    def live_traffic(self, kernel="vpt_mesh_kernel", workload_args=(), passes=(("FETCH_SIZE",), ("WRITE_SIZE",))):
        """Memory-side bytes of one launch of the default workload, measured NOW: two child runs of this script under
        `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counters only: MI355X_MICROARCH.md, HBM section;
        the program after `--` is python3 itself), the last dispatch of the kernel summed over its rows (XCDs).  Units are KB;
        Infinity-Cache hits are included, so the figure is an upper bound on DRAM bytes.  Any failure returns {"error": ...} and the
        line falls back to the committed passes of profiles/."""
        import csv
        import glob
        import shutil
        import subprocess
        import tempfile
        exe = shutil.which("rocprofv3")
        if not exe:
            return {"error": "rocprofv3 not found"}
        if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
            return {"error": "this run is itself under a profiler: no nested rocprofv3"}
        out = {}
        env = dict(os.environ, TMPDIR="/tmp")
        for counters in passes:
            d = tempfile.mkdtemp(prefix="vpt_pmc_")
            cmd = [exe, "--kernel-trace", "--output-format", "csv", "--pmc", *counters, "-d", d, "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--no-cold", "--no-others"] + list(workload_args)
            try:
                subprocess.run(cmd, capture_output=True, text=True, timeout=180, env=env, cwd=ROOT)
                rows = []
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    rows += [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] in counters]
                if not rows:
                    return {"error": f"no {counters} rows for {kernel}"}
                last = max(int(r["Dispatch_Id"]) for r in rows)
                for counter in counters:
                    out[counter + ("_KB" if counter.endswith("_SIZE") else "")] = sum(float(r["Counter_Value"]) for r in rows
                                                                                      if int(r["Dispatch_Id"]) == last and r["Counter_Name"] == counter)
            except Exception as e:   # noqa: BLE001
                return {"error": repr(e)[:300]}
            finally:
                shutil.rmtree(d, ignore_errors=True)
        if "FETCH_SIZE_KB" in out and "WRITE_SIZE_KB" in out:
            out["bytes_per_launch"] = (out["FETCH_SIZE_KB"] + out["WRITE_SIZE_KB"]) * 1024.0
        return out

    def cpu_baseline(self, w):
        import oracle_lib
        vpt, args = self.vpt, self.args
        sres, sspp = (int(x) for x in args.cpu_sample.split("x"))
        ncores = host_cores()
        workdir = os.environ.get("TMPDIR", "/tmp")
        rec = None

        def port(threads, res, spp):   # our CPU restatement, one thread per core: does not oversubscribe
            sp = vpt.PathtraceParams(resolution=res, samples=1 << 30, shader=w.shader, bounces=w.bounces)
            st = w.scene.make_state(sp)
            t1 = time.perf_counter()
            oracle_lib.oracle_render(w.scene, sp, st, spp, nthreads=threads)
            dt = time.perf_counter() - t1
            return {"value": round(st.width * st.height * spp / dt * 1e-6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
                    "sample": f"{w.scene_name} {st.width}x{st.height}x{spp}spp, oracle/vpt_oracle.cpp, {threads} thread(s)"}
        if oracle_lib.have_reference():
            *_, info = oracle_lib.reference_render(w.scene_path, w.shader, sres, sspp, w.bounces, workdir=workdir)
            rec = {"value": round(info["msamples_per_s"], 4), "unit": "Msamples/s", "cores": min(ncores, info["threads"]),
                   "threads_started": info["threads"], "kind": "reference",
                   "sample": f"{w.scene_name} {info['width']}x{info['height']}x{sspp}spp, reference renderer (g++ -O2), "
                             f"{info['threads']} threads (its own hardware_concurrency()) on {ncores} cores"}
            *_, one = oracle_lib.reference_render(w.scene_path, w.shader, max(64, sres // 2), max(1, sspp // 4), w.bounces, workdir=workdir, noparallel=True)
            rec["single_thread"] = {"value": round(one["msamples_per_s"], 4), "unit": "Msamples/s", "cores": 1, "kind": "reference",
                                    "sample": f"{w.scene_name} {one['width']}x{one['height']}x{max(1, sspp // 4)}spp, reference renderer --noparallel"}
            rec["port"] = port(ncores, sres, max(1, sspp // 2))
        else:
            rec = port(ncores, sres, sspp)
            rec["single_thread"] = port(1, max(64, sres // 2), max(1, sspp // 4))
        rec["cpu"] = cpu_model()
        return rec


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
    ap.add_argument("--no-others", action="store_true", help="skip the other BASELINE configs (N = 1) / the config-5 record (N > 1)")
    ap.add_argument("--dist", action="store_true", help="N = 1: run the N > 1 code path (nccl group, all_gather_into_tensor) at world size 1")
    ap.add_argument("--dump-frame", default=None, help="write the resolved frame of the last step (float32 h x w x 4) to this .npy")
    args = ap.parse_args()

    B = Bench(args)
    np, world, rank = B.np, B.world, B.rank
    default_workload = args.scene == SCENE and args.shader == "volpathtrace" and args.bounces == 64 and args.resolution == 1280
    resolution = int(round(args.resolution * (world ** 0.5))) if args.weak else args.resolution

    w = B.workload(args.scene, args.shader, args.bounces, resolution, args.spp)
    elapsed = B.timed(w, args.steps, args.warmup)
    if args.dump_frame and rank == 0:
        np.save(args.dump_frame, w.frame.cpu().numpy())
    balance = None
    if args.balance:   # load-balance figures of the last launch (after the timed region): per-wave durations
        c = w.dev.last_wave_costs().astype(np.float64) * 1e-5   # ticks of 100 MHz -> ms
        c = c[c > 0]
        balance = {"waves": int(len(c)), "longest_wave_ms": round(float(c.max()), 3), "mean_wave_ms": round(float(c.mean()), 4),
                   "sum_wave_ms": round(float(c.sum()), 1), "p99_wave_ms": round(float(np.quantile(c, 0.99)), 3)}

    cold = None
    if world == 1 and not args.no_cold:   # a fresh handle, after the timed region (warm clocks)
        wc = B.workload(args.scene, args.shader, args.bounces, resolution, args.spp)
        cold = B.cold_call(wc)
        cold["what"] = ("first call on a fresh scene handle: no wave costs known yet, so a pilot launch (spp/64 samples) measures "
                        "them and the rest runs in its order; `value` above is the steady state (order from the previous launch)")
        del wc
    single = config5 = None
    if world > 1 and not args.weak:
        single = B.alone_on_rank0(args.scene, args.shader, args.bounces, resolution, args.spp, max(1, min(args.steps, 2)))
        if default_workload and not args.no_others:
            # BASELINE config[4]'s frame, the one ">= 6x at 8 GPUs" can hold on: 03_volume 3840x1600, 256 spp per step
            w5 = B.workload(SCENE, "volpathtrace", 64, 3840, 256)
            e5 = B.timed(w5, 2, 2)   # the first call measures tile costs (pilot), the second may split tiles: both warm-up
            s5 = B.alone_on_rank0(SCENE, "volpathtrace", 64, 3840, 256, 1)
            if rank == 0:
                ms5 = e5 / 2 * 1e3
                config5 = {"workload": f"03_volume volpathtrace bounces=64 {w5.width}x{w5.height}x256spp per step, fixed frame, tiles%{world} "
                                       "(BASELINE config[4]'s frame at 256 of its 4096 spp per step: cost per sample does not depend on spp)",
                           "steps": 2, "warmup": 2, "ms_per_step": round(ms5, 3), "value": round(w5.samples_per_step / ms5 * 1e-3, 3),
                           "unit": "Msamples/s", "kernel_ms_rank0": round(sum(w5.kernel_ms) / len(w5.kernel_ms), 3),
                           "gather_bytes_per_rank": int(w5.slots * 16), "single_gpu": s5,
                           "speedup_vs_1gpu": round(s5["ms_per_step"] / ms5, 3)}
            del w5

    scene_name = w.scene_name
    data_desc = ("reference scene tests/03_volume (real assets), deterministic PCG32 seeds" if args.scene == SCENE else
                 f"{scene_name}: substitute assets (tests/golden/make_scenes.py), deterministic PCG32 seeds")
    value = w.samples_per_step * args.steps / elapsed * 1e-6

    if rank == 0:
        roofline = B.roofline(w)
        def with_live_traffic(rec, wl, kernel, workload_args):
            # the counter passes of THIS run replace the replay from profiles/ (which stays as the fallback and as the per-round record)
            live = B.live_traffic(kernel, workload_args)
            if "bytes_per_launch" in live:
                rec["traffic_committed"] = {"traffic": rec["traffic"], "source": rec["traffic_source"]}
                rec["traffic"] = round(live["bytes_per_launch"])
                rec["traffic_source"] = ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around child runs of this workload "
                                         f"with --steps 1 --warmup 1, last dispatch of the kernel: FETCH_SIZE {live['FETCH_SIZE_KB']:.0f} KB + WRITE_SIZE {live['WRITE_SIZE_KB']:.0f} KB "
                                         f"= {live['bytes_per_launch'] / wl.samples_per_step:.0f} B per sample (the write half is almost entirely register-spill scratch cycling "
                                         "through L2, not data; Infinity-Cache hits count as traffic)")
            else:
                rec["traffic_live_error"] = live.get("error")
            if rec.get("bound") == "valu":   # K2: its VALU figures from a counter pass of this run as well
                sq = B.live_traffic(kernel, workload_args, passes=(("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"),))
                if sq.get("SQ_INSTS_VALU") and sq.get("SQ_ACTIVE_INST_VALU"):
                    per_sample = sq["SQ_INSTS_VALU"] / wl.samples_per_step
                    issued = per_sample * wl.samples_per_step / (rec["kernel_ms"] * 1e-3)
                    rec["valu_committed"] = {"achieved": rec["achieved"], "frac": rec["frac"], "lane_utilisation": rec["lane_utilisation"], "source": rec["valu_source"]}
                    rec["achieved"], rec["frac"] = round(issued * 1e-9, 2), round(issued / VALU_ISSUE_PEAK, 4)
                    rec["lane_utilisation"] = round(sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"]), 4)
                    rec["valu_source"] = (f"live: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU around a child run of this workload ({per_sample:.0f} wave-level "
                                          "VALU instructions per sample) x this run's samples/s / (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction)")
                else:
                    rec["valu_live_error"] = sq.get("error", "counters missing")
            return rec
        live_ok = world == 1 and default_workload and args.spp == 256 and not args.no_others and not args.dist
        if live_ok:
            with_live_traffic(roofline, w, "vpt_mesh_kernel", ())
        cpu_baseline = B.cpu_baseline(w) if (world == 1 and args.cpu_sample != "0") else None
        others = None
        if world == 1 and default_workload and not args.no_others:
            others = []
            # Each at its BASELINE sample count per step (config 5's frame: 256 of its 4096): a lane runs its pixel's samples one after the other and a wave
            # ends with its slowest lane, so the share of a wave's time spent waiting for that lane falls with the length of the chain (config 3: 790 /
            # 842 / 870 Msamples/s at 64 / 256 / 1024 samples per launch, profiles/r04_full_spp_configs.txt) - rounds 1-4 quoted these records at 64 / 128 / 64
            for name, path, shader, bounces, res, spp in (
                    ("config3 (tests/05_head1ss: assets missing, substitute scene)", os.path.join(SCENES, "05_head1ss_sub", "head1ss_sub.json"), "volpathtrace", 64, 1280, 1024),
                    ("config4 (tests/06_gridsdf: assets missing, substitute scene with 96^3 + 64^3 grids)", os.path.join(SCENES, "06_gridsdf_full", "gridsdf_full.json"), "implicit", 4, 1280, 512),
                    ("config5's frame on one GPU", SCENE, "volpathtrace", 64, 3840, 256)):
                wo = B.workload(path, shader, bounces, res, spp)
                first = B.cold_call(wo)
                eo = B.timed(wo, 3, 1)
                ms = eo / 3 * 1e3
                others.append({"config": name, "workload": f"{wo.scene_name} {shader} bounces={bounces} {wo.width}x{wo.height}x{spp}spp per step",
                               "steps": 3, "warmup": 1, "ms_per_step": round(ms, 3), "value": round(wo.samples_per_step / ms * 1e-3, 3),
                               "unit": "Msamples/s", "cold": first, "roofline": B.roofline(wo)})
                if live_ok:
                    with_live_traffic(others[-1]["roofline"], wo, "vpt_render_kernel" if shader.startswith("implicit") else "vpt_mesh_kernel",
                                      ["--scene", path, "--shader", shader, "--bounces", str(bounces), "--resolution", str(res), "--spp", str(spp)])
                del wo
        rccl_path = None
        if world == 1 and default_workload and not args.no_others and not args.dist:
            # the N > 1 code path with one rank, in a process of its own (a failure there must not cost the headline line):
            # nccl (= RCCL) process group, all_gather_into_tensor of the tile buffer and the resolve of the gathered buffer per step
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--dist", "--steps", "3", "--warmup", "1", "--no-cold", "--no-others", "--cpu-sample", "0"]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
                sub = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                rccl_path = {"ok": True, "value": sub["value"], "unit": "Msamples/s", "ms_per_step": sub["ms_per_step"], "steps": 3,
                             "what": "bench.py --dist: the same workload with every step ending in all_gather_into_tensor over an nccl (RCCL) "
                                     "process group of ONE rank + resolve of the gathered buffer - the collective path of --gpus N, executed on this box"}
            except Exception as e:   # noqa: BLE001
                rccl_path = {"ok": False, "error": repr(e)[:300]}
        if world == 1:
            workload_note = ""
        elif args.weak:
            workload_note = ", frame grows with N"
        else:
            workload_note = (", fixed frame (a pixel's samples are a serial RNG chain: the costliest 8x8 tile bounds the launch; `serial_chain` "
                             "holds the bound measured in this run" + ("; `config5` is the 3840x1600 frame" if config5 else "") + ")")
        line = {
            "metric": "Msamples/sec", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": data_desc + (" [REHEARSAL: all ranks on one GPU over gloo - timings are not a measurement]" if B.rehearsal else ""),
            "config": {"workload": f"{scene_name} {args.shader} bounces={args.bounces} {w.width}x{w.height}x{args.spp}spp per step" + workload_note,
                       "tile": f"{args.tile}x{args.tile}",
                       "parallelism": (f"tiles%{world}, costly tiles split into partly filled waves (auto)" if world > 1 else
                                       "1gpu + nccl group of one rank (all_gather_into_tensor per step)" if args.dist else "1gpu"),
                       "samples_per_step": w.samples_per_step},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if balance:
            line["balance"] = balance
        if cold:
            line["cold"] = cold
        if single:
            # what the serial chain allows on this frame, from this run's own launch: rank 0's costliest wave (100 MHz ticks the kernel
            # records per wave) against the time the whole frame took on rank 0 alone
            c = w.dev.last_wave_costs().astype(np.float64) * 1e-5
            c = c[c > 0]
            if len(c):
                line["serial_chain"] = {"longest_wave_ms_rank0": round(float(c.max()), 3), "mean_wave_ms_rank0": round(float(c.mean()), 4),
                                        "waves_rank0": int(len(c)), "single_gpu_ms": single["ms_per_step"],
                                        "speedup_bound": round(single["ms_per_step"] / float(c.max()), 3),
                                        "what": "a launch is never shorter than its costliest wave: speed-up over one GPU <= single_gpu_ms / longest_wave_ms (measured here, not a constant)"}
            line["single_gpu"] = single
            line["speedup_vs_1gpu"] = round(single["ms_per_step"] / (elapsed / args.steps * 1e3), 3)
        if config5:
            line["config5"] = config5
        if others:
            line["other_configs"] = others
        if rccl_path:
            line["rccl_path_one_rank"] = rccl_path
        print(json.dumps(line), flush=True)
    if B.dist is not None:
        B.dist.barrier()
        B.dist.destroy_process_group()


if __name__ == "__main__":
    main()
