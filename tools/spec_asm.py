"""Diagnostic (CPU only, no GPU needed): build the config-specialised k_env_step for one of the fixture configs exactly as an env would
(env_config.build -> pbhc_env_config_finalize -> generated header -> hipcc) and write its gfx950 assembly + a static instruction census.

    python tools/spec_asm.py [v1_g1_23dof_walk.yaml] [--general] [--out /tmp/spec] [--defs "-DX -DY"]
Not part of the product or the tests."""
import argparse
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbhc_amd import _lib, specialise as S          # noqa: E402
from pbhc_amd.envs import env_config                # noqa: E402
from pbhc_amd.envs.motion_tracking import _TopView  # noqa: E402
from pbhc_amd.skeleton import Skeleton              # noqa: E402
from pbhc_amd.utils.config import load_config       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("cfg", nargs="?", default="v1_g1_23dof_walk.yaml")
ap.add_argument("--general", action="store_true")
ap.add_argument("--out", default="/tmp/spec")
ap.add_argument("--defs", default="")
ap.add_argument("--noise-off", action="store_true")
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = load_config(os.path.join(ROOT, "tests", "golden", "configs", a.cfg), {"num_envs": 4096, "simulator._target_": "pbhc_amd.simulator.replay_stub.ReplaySimStub"}, now="t")
if a.noise_off:
    for k in list(cfg.obs.noise_scales.keys()):
        cfg.obs.noise_scales[k] = 0.0
skel = Skeleton.from_motion_config(cfg.robot.motion)


class _ML:
    has_contact_mask = False


nl = len(cfg.domain_rand.get("randomize_link_body_names", []))
c, L = env_config.build(_TopView(cfg.env.config), skel, _ML(), 4096, "cpu", nl, seed=1, mode=1 if a.general else 0)
cf = S.finalised(c)
text = S.emit_header(cf)
os.makedirs(a.out, exist_ok=True)
hdr = os.path.join(a.out, "cfg.h")
open(hdr, "w").write(text)
asm = os.path.join(a.out, "k.s")
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + S.KERNEL_FLAGS + a.defs.split() + [f'-DPBHC_STATIC_CFG="{hdr}"', f"-DPBHC_SPEC_MODE={int(cf.tracking_mode)}",
      "--cuda-device-only", "-S", "-o", asm, os.path.join(S.CSRC, "pbhc_env_step_spec.hip")]
subprocess.run(cmd, check=True)
src = open(asm).read()
body = src[src.index("_Z10k_env_step"):]
census = collections.Counter()
n = 0
for line in body.splitlines():
    line = line.strip()
    m = re.match(r"^([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|flat_[a-z0-9_]+|scratch_[a-z0-9_]+)\b", line)
    if not m:
        if line.startswith(".end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            pass
        continue
    op = m.group(1)
    n += 1
    cls = ("VMEM_LD" if re.match(r"(global|buffer|flat)_load", op) else "VMEM_ST" if re.match(r"(global|buffer|flat)_store", op) else "SCRATCH" if op.startswith("scratch") else
           "LDS" if op.startswith("ds_") else "SMEM" if op.startswith("s_load") or op.startswith("s_buffer") else "BRANCH" if op.startswith("s_cbranch") or op.startswith("s_branch") else
           "WAIT" if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop") else "SALU" if op.startswith("s_") else "VALU")
    census[cls] += 1
    census["op:" + op] += 1
print(f"static instructions: {n}")
for k in ("VALU", "SALU", "LDS", "VMEM_LD", "VMEM_ST", "SMEM", "BRANCH", "WAIT", "SCRATCH"):
    print(f"  {k:8s} {census[k]}")
for key in ("vgpr_count", "sgpr_count", "lds_size", "scratch", "spill"):
    for m in re.finditer(r"^\s*[;.].*" + key + r".*$", src, flags=re.M | re.I):
        print("  ", m.group(0).strip())
top = sorted(((v, k) for k, v in census.items() if k.startswith("op:")), reverse=True)[:40]
print("  top ops: " + ", ".join(f"{k[3:]} {v}" for v, k in top))
print("asm:", asm)

# ---- per-source-line census (needs --defs "-gline-tables-only"): instructions attributed to the last .loc before them
if "-gline-tables-only" in a.defs or "-g" in a.defs.split():
    files = {}
    for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', src, flags=re.M):
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    per = collections.Counter()
    cur = None
    for line in body.splitlines():
        t = line.strip()
        m = re.match(r"^\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if re.match(r"^(v_|s_|ds_|global_|buffer_|flat_)", t) and cur:
            per[cur] += 1
    byfile = collections.Counter()
    for (f, l), v in per.items():
        byfile[f] += v
    print("by file:", dict(byfile))
    BUCKET = 25
    agg = collections.Counter()
    for (f, l), v in per.items():
        agg[(f, l // BUCKET * BUCKET)] += v
    for (f, l), v in sorted(agg.items()):
        if v >= 15:
            print(f"  {f}:{l:5d}-{l + BUCKET - 1:5d}  {v}")
