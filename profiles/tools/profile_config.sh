#!/bin/bash
# Profile one bench.py configuration on the GPU box.  Usage (from the repo root, inside gpurun):
#   bash profiles/tools/profile_config.sh <tag> [passes] -- <bench.py arguments>
# passes: quoted list out of  plain stats fetch write sq  (default: all).  Everything lands in gpurun_out/<tag>/ ;
# profiles/tools/summarize_profile.py turns that directory into the files committed under profiles/.
#
# rocprofv3 rules on this pool (MI355X_MICROARCH.md, "rocprofv3 PMC slots"): counters in their own runs next to
# --kernel-trace only; per pass at most 8 SQ counters and 4 TCC slots — FETCH_SIZE takes 3, WRITE_SIZE 2, so each
# gets a pass of its own.  A request beyond that aborts rocprofv3 with "Request exceeds the capabilities of the
# hardware to collect" (round 1, gpurun_out/pmc_l.log: a counter-set mistake, not a kernel fault).
# The program after `--` is python3 itself (no env / bash -c hop: the profiler initialises the GPU first).
set -e -o pipefail
tag=$1; shift
passes="plain stats fetch write sq"
if [ "$1" != "--" ]; then passes=$1; shift; fi
shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
small="--steps 1 --warmup 1 --cpu-sample 0 --no-cold --no-others"
for p in $passes; do
  case $p in
    plain) python3 bench.py "$@" > $out/bench.json 2> $out/bench.err ;;
    stats) rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py "$@" --cpu-sample 0 --no-cold --no-others > $out/stats.log 2>&1 ;;
    fetch) rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $out/fetch -o run -- python3 bench.py "$@" $small > $out/fetch.log 2>&1 ;;
    write) rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $out/write -o run -- python3 bench.py "$@" $small > $out/write.log 2>&1 ;;
    sq)    rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD \
             -d $out/sq -o run -- python3 bench.py "$@" $small > $out/sq.log 2>&1 ;;
  esac
  echo "[$tag] pass $p done"
done
