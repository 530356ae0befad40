"""Averages of the rocprofv3 --pmc passes collected by tools/env_step_pmc.sh over the k_env_step launches (per launch, and per wave where
that is the natural unit).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_env_step"
acc = collections.OrderedDict()
meta = {}
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if kernel not in r.get("Kernel_Name", ""):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        meta = {k: r.get(k) for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
if not acc:
    raise SystemExit("no rows for " + kernel)
# the probe's last launches are the steady state; skip the first 20 (warm-up, reset_all) of every counter
mean = {k: sum(v[20:]) / max(len(v[20:]), 1) for k, v in acc.items()}
waves = mean.get("SQ_WAVES", 0.0)
print(f"# {kernel}: {meta}")
print(f"# launches averaged per counter: {[len(v[20:]) for v in acc.values()][0]}; waves per launch {waves:.0f}")
for k, v in mean.items():
    per = f"{v / waves:12.1f} / wave" if waves else ""
    print(f"{k:30s} {v:16.0f} {per}")
g = lambda k: mean.get(k, 0.0)
if waves:
    print("# derived")
    tot = g("SQ_INSTS_VALU") + g("SQ_INSTS_SALU") + g("SQ_INSTS_LDS") + g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR") + g("SQ_INSTS_SMEM") + g("SQ_INSTS_BRANCH")
    print(f"instructions per wave (VALU+SALU+LDS+VMEM+SMEM+BRANCH) {tot / waves:10.0f}; VALU share {g('SQ_INSTS_VALU') / max(tot, 1):.2f}")
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    print(f"of SQ_WAVE_CYCLES: WAIT_ANY {g('SQ_WAIT_ANY') / wc:.2f}  WAIT_INST_ANY {g('SQ_WAIT_INST_ANY') / wc:.2f}  ACTIVE_INST_ANY {g('SQ_ACTIVE_INST_ANY') / wc:.2f}  "
          f"(ACTIVE_INST_VALU {g('SQ_ACTIVE_INST_VALU') / wc:.2f}, LDS {g('SQ_ACTIVE_INST_LDS') / wc:.2f}, VMEM {g('SQ_ACTIVE_INST_VMEM') / wc:.2f}, SCA {g('SQ_ACTIVE_INST_SCA') / wc:.2f})")
if g("SQC_ICACHE_REQ"):
    print(f"instruction cache: {g('SQC_ICACHE_REQ'):.0f} requests, {g('SQC_ICACHE_MISSES'):.0f} misses (+{g('SQC_ICACHE_MISSES_DUPLICATE'):.0f} duplicate) -> hit rate "
          f"{g('SQC_ICACHE_HITS') / g('SQC_ICACHE_REQ'):.4f}")
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    print(f"VALU lane utilisation: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = {g('SQ_THREAD_CYCLES_VALU') / (64.0 * g('SQ_ACTIVE_INST_VALU')):.2f}")
