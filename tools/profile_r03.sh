#!/bin/bash
# Profiles of one benchmark frame on the GPU box: rocprofv3 kernel statistics and PMC passes (counters in their own runs, --kernel-trace only),
# folded on the spot into the summaries that are committed under profiles/ (tools/fold_profile.py); the raw per-dispatch CSVs are dropped
# (gpurun copies back at most 64 MiB).
# usage: tools/profile_r03.sh TAG [SCENE W H SPP PHOTONS]   -> gpurun_out/prof_TAG/summary_*     (default: BASELINE config 3)
set -u
TAG=$1; shift
SCENE=${1:-caustics}; W=${2:-1920}; H=${3:-1080}; SPP=${4:-256}; PH=${5:-200000}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu --no-others --no-executed --scene $SCENE --width $W --height $H --spp $SPP --photons $PH"
python3 bench.py --steps 3 --warmup 1 $ARGS > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 bench.py --steps 3 --warmup 1 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
pass() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- python3 bench.py --steps 1 --warmup 0 $ARGS > $OUT/$name.json 2> $OUT/$name.err; echo "$name done: $(tail -c 120 $OUT/$name.err | tr '\n' ' ')"; }
pass sq_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM
pass sq_b SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM
pass sq_c SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32
pass ta TA_BUSY TA_TOTAL_WAVEFRONTS
pass tcp TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 tools/fold_profile.py $OUT $OUT/summary $SCENE $W $H $SPP $PH > $OUT/summary.txt 2>&1
python3 tools/check_profile_agreement.py $OUT/summary >> $OUT/summary.txt 2>&1
for d in stats sq_a sq_b sq_c ta tcp fetch write; do rm -rf $OUT/$d; done
cat $OUT/summary.txt; du -sh $OUT
