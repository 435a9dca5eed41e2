# PMC view of the mel kernel alone (tools/bench_mel.py): VALU / LDS instruction counts, LDS bank conflicts, busy cycles.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_mel
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/insts -- python3 $GRAFT_REPO_ROOT/tools/bench_mel.py > $OUT/insts.log 2>&1 || echo insts-rc=$?
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/lds -- python3 $GRAFT_REPO_ROOT/tools/bench_mel.py > $OUT/lds.log 2>&1 || echo lds-rc=$?
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/busy -- python3 $GRAFT_REPO_ROOT/tools/bench_mel.py > $OUT/busy.log 2>&1 || echo busy-rc=$?
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_mel"
for sub in ("insts", "lds", "busy"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            if "mel_fwd" in row.get("Kernel_Name", ""):
                a = agg[row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
        for k, (n, v) in sorted(agg.items()):
            print(f"{sub:6s} {k:28s} launches {n:5d}  per launch {v / max(n, 1):16.1f}")
PY
find $OUT -size +5M -delete
