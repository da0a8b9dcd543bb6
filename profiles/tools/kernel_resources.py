#!/usr/bin/env python3
"""Register / scratch / LDS budget of every vpt_* kernel as the compiler reports it:

    python profiles/tools/kernel_resources.py [extra hipcc flags ...]

Compiles csrc/vpt_capi.hip for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed) and
prints one line per kernel of ours (rocPRIM's sort kernels are skipped)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "volumetric-path-tracer_amd")


def main():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-DVPT_WAVES_PER_SIMD=3", "-I../include", "-Icsrc", "-Ihost", "-Rpass-analysis=kernel-resource-usage",
           "csrc/vpt_capi.hip", "-o", "/tmp/vpt_resources_probe.so"] + sys.argv[1:]
    err = subprocess.run(cmd, cwd=PKG, stderr=subprocess.PIPE, text=True).stderr
    cur, rows = None, []
    for line in err.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: +(\w[\w \[\]/]*): +(\S+)", line) or re.search(r"remark: +(\w[\w \[\]/]*): +(\S+)", line)
        if not m:
            m = re.search(r":\d+:\d+: +([A-Za-z][\w \[\]/]*): +(\S+) \[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key in ("Function Name", "Name"):
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
        if "rocprim" in name:
            continue
        short = re.sub(r"\(.*", "", name)
        print(f"{short:55s} VGPR {r.get('VGPRs', '?'):>4} AGPR {r.get('AGPRs', '?'):>3} SGPR {r.get('TotalSGPRs', r.get('SGPRs', '?')):>4} "
              f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>5} B/lane  occupancy {r.get('Occupancy [waves/SIMD]', '?')}  LDS {r.get('LDS Size [bytes/block]', '?')}")


if __name__ == "__main__":
    main()
