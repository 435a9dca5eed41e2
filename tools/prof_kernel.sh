# PMC view of one kernel: tools/prof_kernel.sh <kernel-name-substring> <tag> <prof_one.py args...>   (run on the GPU box)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
PAT=$1; TAG=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
run() { sub=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$sub -- python3 $GRAFT_REPO_ROOT/tools/prof_one.py $ARGS > $OUT/$sub.log 2>&1 || echo $sub-rc=$?; }
ARGS="$*"
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA
run p3 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM
run p4 FETCH_SIZE
run p5 TCC_HIT_sum TCC_MISS_sum
run p6 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY
PAT="$PAT" OUT="$OUT" python3 - <<'PY'
import csv, glob, os, collections
out, pat = os.environ["OUT"], os.environ["PAT"]
for sub in ("p1", "p2", "p3", "p4", "p5", "p6"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if pat in row.get("Kernel_Name", ""):
                a = agg[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
        for k, (n, v) in sorted(agg.items()):
            print(f"{sub:4s} {k:34s} launches {n:4d}  per launch {v / max(n, 1):18.1f}")
PY
find $OUT -size +5M -delete
