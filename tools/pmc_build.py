"""Merge the outputs of tools/prof_step.sh (rocprofv3 --kernel-trace --stats, and three separate --pmc passes:
FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE) into one JSON:
per kernel the launch count, average duration, HBM-side bytes per launch and the MFMA-busy share.

  python3 tools/pmc_build.py gpurun_out/prof_r02 profiles/pmc_r02_summary.json

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 sections): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes, so read bytes =
2 x FETCH_SIZE x 1024 (an upper bound for kernels whose reads are narrower); WRITE_SIZE is exact.
SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles with the matrix pipe busy, summed over the SIMDs (32 per
v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is summed over the 8 XCDs.  mfma_busy = MFMA busy cycles /
(1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of the dispatch during which a SIMD's matrix pipe was executing."""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

root, out = Path(sys.argv[1]), Path(sys.argv[2])


def short(full):
    m = re.search(r"(?:\)::|\s|^)(\w+)(<[^(]*>)?\(", full)
    name = (m.group(1) + (m.group(2) or "")) if m else full[:80]
    return re.sub(r"\(anonymous namespace\)::|pe::", "", name)[:100]


def counters(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for path in (root / sub).rglob("*counter_collection.csv"):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


stats = {}
for path in (root / "trace").rglob("*kernel_stats.csv"):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                       "share_of_gpu_time": float(r["Percentage"]) / 100.0}
fetch, write, sq = counters("fetch"), counters("write"), counters("sq")
mean = lambda v: sum(v) / len(v) if v else None
kernels = {}
for name in sorted(set(stats) | set(sq), key=lambda n: -(stats.get(n, {}).get("share_of_gpu_time", 0.0))):
    if not any(tag in name for tag in ("kernel", "Kernel")) or name.startswith("__amd"):
        continue
    f, w, s = mean(fetch[name]["FETCH_SIZE"]), mean(write[name]["WRITE_SIZE"]), sq[name]
    mfma, busy, gui = mean(s["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(s["SQ_BUSY_CYCLES"]), mean(s["GRBM_GUI_ACTIVE"])
    e = dict(stats.get(name, {}))
    e["pmc_launches"] = len(s["GRBM_GUI_ACTIVE"])
    if f is not None and w is not None:
        e.update(fetch_size_kib=f, write_size_kib=w, hbm_bytes_per_launch=(2.0 * f + w) * 1024.0)
    if mfma is not None and gui:
        e.update(sq_valu_mfma_busy_cycles=mfma, sq_busy_cycles=busy, sq_wave_cycles=mean(s["SQ_WAVE_CYCLES"]),
                 grbm_gui_active=gui, mfma_busy=mfma / (1024.0 * gui / 8.0),
                 mfma_busy_over_sq_busy=(mfma / busy) if busy else None)
    kernels[name] = e
out.write_text(json.dumps({"note": __doc__, "kernels": kernels}, indent=1))
for n, e in list(kernels.items())[:16]:
    print(f"{n[:70]:70s} calls {e.get('calls')} avg {e.get('avg_us', 0):9.1f} us  mfma_busy {e.get('mfma_busy')}"
          f"  hbm MB/launch {e.get('hbm_bytes_per_launch', 0) / 1e6:9.1f}")
