# Where K1's waiting cycles go: three rocprofv3 --pmc passes (vector L1, L2, sequencer) around one timed launch.
#   bash profiles/tools/latency_pmc.sh <tag> -- <bench.py arguments>        (inside gpurun; summaries: profiles/tools/latency_pmc.py)
set -e -o pipefail
tag=$1; shift; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
small="--steps 1 --warmup 1 --cpu-sample 0 --no-cold --no-others"
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum -d $out/tcp -o run -- python3 bench.py "$@" $small > $out/tcp.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_REQ_sum TCC_MISS_sum -d $out/tcc -o run -- python3 bench.py "$@" $small > $out/tcc.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA -d $out/sq2 -o run -- python3 bench.py "$@" $small > $out/sq2.log 2>&1
echo "[$tag] done"
