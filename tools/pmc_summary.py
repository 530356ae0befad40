"""Summarise the two rocprofv3 PMC passes of k_env_step into profiles/<round>_k_env_step_pmc.json, keyed by a hash of the kernel sources so
that bench.py reports `roofline.traffic` only for the code that was actually profiled.

  cd /tmp && export TMPDIR=/tmp      # on the GPU box, separate passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/kernel_probe.py 4096
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 tools/kernel_probe.py 4096
  python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/round3_k_env_step_pmc.json
"""
import csv
import glob
import hashlib
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["pbhc_amd/csrc/pbhc_kernels.hip", "pbhc_amd/csrc/pbhc_env_step.h", "pbhc_amd/csrc/pbhc_env_step_spec.hip", "pbhc_amd/csrc/pbhc_math.h", "include/pbhc_hip.h"]


def _code_only(text):
    """C / C++ source without comments and whitespace: what the compiler sees, so that editing documentation does not orphan a profile"""
    import re

    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


def source_hash():
    h = hashlib.sha256()
    for f in SOURCES:
        h.update(_code_only(open(os.path.join(ROOT, f), "r").read()).encode())
    return h.hexdigest()[:16]


def counter_values(d, name, kernel="k_env_step"):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name and kernel in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
    return vals


if __name__ == "__main__":
    fetch_dir, write_dir, out = sys.argv[1:4]
    what = sys.argv[4] if len(sys.argv) > 4 else "tools/kernel_probe.py 4096 (v1 walk, 4096 envs)"
    fv, wv = counter_values(fetch_dir, "FETCH_SIZE"), counter_values(write_dir, "WRITE_SIZE")
    if not fv or not wv:
        raise SystemExit(f"no k_env_step rows found (FETCH {len(fv)}, WRITE {len(wv)})")
    rec = {"kernel": "k_env_step", "source_sha16": source_hash(), "sources": SOURCES, "launches": [len(fv), len(wv)],
           "FETCH_SIZE_KiB_median": statistics.median(fv), "WRITE_SIZE_KiB_median": statistics.median(wv),
           "note": "rocprofv3 --pmc, separate passes, " + what + "; on gfx950 FETCH_SIZE counts 64 B per 128-B request "
                   "(MI355X_MICROARCH.md, HBM): bytes read = 2 x FETCH_SIZE; both counters are in KiB"}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))
