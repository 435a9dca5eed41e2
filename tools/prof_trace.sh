# Kernel-trace statistics of the default step (side-stream overlaps on) and of the serialised step.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02c
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-native-ref --family-steps 0 --host-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/overlap -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/overlap.log 2>&1 || echo rc=$?
export PE_OVERLAP_CONV_WGRAD=0 PE_OVERLAP_LSTM_WGRAD=0 PE_OVERLAP_TF_WGRAD=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/serial.log 2>&1 || echo rc=$?
find $OUT -name "*kernel_trace.csv" -size +30M -delete
find $OUT -name "*.csv" | head
