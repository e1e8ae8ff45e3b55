#!/bin/bash
# One profiling session on the GPU box: bench line + rocprofv3 kernel trace + PMC passes (each counter set in its own run,
# --kernel-trace/--stats never combined with --pmc).  Usage: bash tools/profile_session.sh gpurun_out/prof_r02
# Afterwards (build container): python tools/profile_summary.py gpurun_out/prof_r02 r02
set -o pipefail
S=${1:-gpurun_out/prof}
R=$PWD
mkdir -p $S && S=$(cd $S && pwd)
export TMPDIR=/tmp
python3 bench.py > $S/bench_default.json 2> $S/bench_default.err || exit 1
cd /tmp
Q="--no-cpu-baseline --profile-steps 0 --no-small-batch"
rocprofv3 --kernel-trace --stats --output-format csv -d $S/stats -- python3 $R/bench.py --steps 10 --warmup 3 $Q --single-stream > $S/stats.log 2>&1 || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $S/fetch -- python3 $R/bench.py --steps 2 --warmup 1 $Q > $S/fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $S/write -- python3 $R/bench.py --steps 2 --warmup 1 $Q > $S/write.log 2>&1 || exit 4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $S/mfma -- python3 $R/bench.py --steps 2 --warmup 1 $Q --single-stream > $S/mfma.log 2>&1 || exit 5
# keep what is merged back small: the per-dispatch traces are not needed, the stats and counter CSVs are
find $S -name "*_kernel_trace.csv" -delete
du -sh $S
