#!/bin/bash
# Profiles of one benchmark frame (BASELINE config 3 by default) on the GPU box: rocprofv3 kernel statistics and PMC passes (counters in their own
# runs, --kernel-trace only).  usage: tools/profile_r02.sh TAG [bench.py args]   -> gpurun_out/prof_TAG/
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu --no-others $*"
python3 bench.py --steps 3 --warmup 1 $ARGS > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 bench.py --steps 3 --warmup 1 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
pass() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- python3 bench.py --steps 1 --warmup 0 $ARGS > $OUT/$name.json 2> $OUT/$name.err; echo "$name done: $(tail -c 200 $OUT/$name.err | tr '\n' ' ')"; }
pass sq_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM
pass sq_b SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM
pass sq_c SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32
pass ta TA_BUSY TA_TOTAL_WAVEFRONTS
pass tcp TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ
pass fetch FETCH_SIZE
pass write WRITE_SIZE
find $OUT -name "*.csv" | head -30
