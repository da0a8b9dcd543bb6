#!/usr/bin/env python3
"""gpurun_out/<tag>/{tcp,tcc,sq2} (profiles/tools/latency_pmc.sh) -> profiles/<name>_latency_pmc.json:
   python profiles/tools/latency_pmc.py <tag> <name> <kernel substring>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import ROOT, counters  # noqa: E402

tag, name, kernel = sys.argv[1:4]
src = os.path.join(ROOT, "gpurun_out", tag)
c = {}
for d in ("tcp", "tcc", "sq2"):
    c.update(counters(os.path.join(src, d), kernel))
der = {}
if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
    der["vector_L1_hit_rate"] = 1 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
if c.get("TCP_TCC_READ_REQ_sum"):
    der["mean_L1_miss_latency_cycles"] = c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / c["TCP_TCC_READ_REQ_sum"]
if c.get("TCC_REQ_sum"):
    der["L2_hit_rate"] = 1 - c.get("TCC_MISS_sum", 0) / c["TCC_REQ_sum"]
if c.get("SQ_WAVE_CYCLES"):
    der["wave_cycles_waiting_any"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
    der["wave_cycles_waiting_for_an_instruction_slot"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    der["wave_cycles_waiting_on_lds"] = c.get("SQ_WAIT_INST_LDS", 0) / c["SQ_WAVE_CYCLES"]
c["_derived"] = der
c["_note"] = f"{kernel}, last dispatch, summed over XCDs; three rocprofv3 --pmc passes (tcp / tcc / sq2 of profiles/tools/latency_pmc.sh)"
json.dump(c, open(os.path.join(ROOT, "profiles", name + "_latency_pmc.json"), "w"), indent=1)
print(json.dumps(der, indent=1))
