# Kernel-trace statistics + three separate PMC passes of one bench step (run on the GPU box via gpurun).
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-native-ref --family-steps 0 --host-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || echo trace-rc=$?
ARGS1="--steps 1 --warmup 1 --no-cpu-baseline --no-native-ref --family-steps 0 --host-steps 0"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS1 > $OUT/fetch.log 2>&1 || echo fetch-rc=$?
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS1 > $OUT/write.log 2>&1 || echo write-rc=$?
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS1 > $OUT/sq.log 2>&1 || echo sq-rc=$?
du -sh $OUT/*; find $OUT -name "*.csv" | head -20
# keep the merge small: drop anything bigger than 20 MB
find $OUT -size +20M -delete
