# Kernel-trace statistics of the step with the data-parallel wiring live over RCCL at world size 1.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_dp
rm -rf $OUT; mkdir -p $OUT
export PE_DP_REHEARSE=1 GPU_MAX_HW_QUEUES=3 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-native-ref --family-steps 0 --host-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || echo rc=$?
find $OUT -name "*kernel_trace.csv" -size +30M -delete
find $OUT -name "*kernel_stats.csv" | head -3
