#!/usr/bin/env python3
"""gpurun_out/<tag>/ (written by profile_config.sh) -> profiles/<name>_{bench.json, kernel_stats.csv, hbm_traffic.json,
pmc_summary.json}:   python profiles/tools/summarize_profile.py <tag> <name> <kernel name substring>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def counters(directory, kernel):
    """per counter: value of the LAST dispatch of `kernel` (the timed launch), summed over its rows (XCDs / dimensions).
    Reads rocprofv3's CSV output or, where it wrote its default rocpd database, the counters_collection view of that."""
    rows = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    for f in glob.glob(os.path.join(directory, "**", "*_results.db"), recursive=True):
        import sqlite3
        cur = sqlite3.connect(f).execute("select dispatch_id, counter_name, value from counters_collection where kernel_name like ?", (f"%{kernel}%",))
        rows += [{"Dispatch_Id": d, "Counter_Name": n, "Counter_Value": v} for d, n, v in cur]
    if not rows:
        return {}
    last = max(int(r["Dispatch_Id"]) for r in rows)
    tot = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
    return dict(tot)


def main():
    tag, name, kernel = sys.argv[1:4]
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", name)
    line = None
    if os.path.exists(os.path.join(src, "bench.json")):
        text = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")]
        if text:
            line = json.loads(text[-1])
            json.dump(line, open(dst + "_bench.json", "w"), indent=1)
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        keep = [r for r in csv.reader(open(f))]
        with open(dst + "_kernel_stats.csv", "w", newline="") as out:
            csv.writer(out).writerows(keep)
    for f in glob.glob(os.path.join(src, "stats", "**", "*_results.db"), recursive=True):   # rocpd database: the same summary
        import sqlite3
        con = sqlite3.connect(f)
        rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        with open(dst + "_kernel_stats.csv", "w", newline="") as out:
            w = csv.writer(out)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for name, calls, tot, avg, mn, mx in rows:
                short = name if len(name) < 160 else name[:157] + "..."
                w.writerow([short, calls, int(tot), round(avg, 1), round(100.0 * tot / total, 4), int(mn), int(mx)])
    fetch, write = counters(os.path.join(src, "fetch"), kernel), counters(os.path.join(src, "write"), kernel)
    if fetch or write:
        samples = line["config"]["samples_per_step"] if line else None
        fk, wk = fetch.get("FETCH_SIZE"), write.get("WRITE_SIZE")
        t = {"workload": line["config"]["workload"] if line else tag, "kernel": kernel, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
             "samples_per_launch": samples,
             "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE around `python3 bench.py <args> "
                       "--steps 1 --warmup 1 --cpu-sample 0`; the last dispatch of the kernel is quoted.  Units are KB "
                       "(MI355X_MICROARCH.md, HBM section); FETCH_SIZE under-counts wide coalesced streams by 2x on gfx950 - this "
                       "kernel's reads are 16 B/lane gathers (uncalibrated, not doubled); Infinity-Cache hits are included, so the sum "
                       "is an upper bound on DRAM bytes."}
        if fk is not None and wk is not None:
            t["bytes_per_launch"] = (fk + wk) * 1024.0
            if samples:
                t["bytes_per_sample"] = t["bytes_per_launch"] / samples
        json.dump(t, open(dst + "_hbm_traffic.json", "w"), indent=1)
    sq = counters(os.path.join(src, "sq"), kernel)
    if sq:
        d = {}
        if sq.get("SQ_ACTIVE_INST_VALU"):
            d["valu_lane_utilisation"] = sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"])
        if sq.get("SQ_WAVE_CYCLES"):
            d["wave_wait_fraction"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
        if line:
            d["valu_wave_instructions_per_sample"] = sq["SQ_INSTS_VALU"] / line["config"]["samples_per_step"]
            d["vmem_read_wave_instructions_per_sample"] = sq.get("SQ_INSTS_VMEM_RD", 0) / line["config"]["samples_per_step"]
        sq["_derived"] = d
        sq["_note"] = f"{kernel}, last dispatch, summed over XCDs; one rocprofv3 --pmc pass; workload: {line['config']['workload'] if line else tag}"
        json.dump(sq, open(dst + "_pmc_summary.json", "w"), indent=1)
    print("wrote", dst + "_*")


if __name__ == "__main__":
    main()
