"""Diagnostic: average k_env_step time and (with PBHC_LIB=pbhc_amd/libpbhc_hip_stamps.so) the per-phase
shader-clock shares of workgroup 0.  Not part of the product or the tests."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import build, make_replay_on_device
from pbhc_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
WORKLOAD = sys.argv[2] if len(sys.argv) > 2 else "v1_walk"          # kernel_probe.py N [workload [clips]]: bench.py's WORKLOADS
CLIPS = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg, env, _ = build(N, "cuda:0", 0, workload=WORKLOAD, num_clips=CLIPS)
env.reset_all()
env.simulator.set_replay(*make_replay_on_device(env, 130, 1))
lib = _lib.lib()
print(f"step kernel: {'specialised to the config' if env.is_specialised else 'generic'}")
dbg = lib
if env.is_specialised and "-DPBHC_STAMPS" in os.environ.get("PBHC_SPEC_DEFINES", ""):
    # the specialised object carries its own stamp buffers (PBHC_SPEC_DEFINES=-DPBHC_STAMPS python tools/kernel_probe.py 4096)
    from pbhc_amd import specialise as _spec

    _c = _lib.PbhcEnvConfig()
    _lib.check(lib.pbhc_env_get_config(env._env, C.byref(_c)))
    dbg = C.CDLL(_spec.ensure(_c, "cached"))
    dbg.pbhc_debug_read_stamps, dbg.pbhc_debug_read_wg_times = dbg.pbhc_spec_read_stamps, dbg.pbhc_spec_read_wg_times
# experiments on the write traffic (PMC WRITE_SIZE): have the step store the optional outputs it leaves to lazy re-derivation by default
if os.environ.get("PBHC_PROBE_EAGER_OUTPUTS", "0") == "1":
    env.set_eager_outputs(True)
act = torch.zeros(N, env.num_dof, device="cuda:0")
STEPS = int(os.environ.get("PBHC_PROBE_STEPS", "100"))
for _ in range(20):
    env.step({"actions": act})
env.set_profiling(True)
for _ in range(STEPS):
    env.step({"actions": act})
if os.environ.get("PBHC_PROBE_RESET_WG0", "0") == "1":       # the stamped workgroup's env 0 times out in the last launch: phase stamps of the reset path
    env.episode_length_buf[0] = 10 ** 6
    env.step({"actions": act})
buf = (C.c_float * 512)(); cnt = C.c_int(0)
_lib.check(lib.pbhc_env_profile_read(env._env, buf, min(STEPS, 100), C.byref(cnt)))
ov = C.c_float(0.0)
if hasattr(lib, "pbhc_env_profile_overhead"):
    _lib.check(lib.pbhc_env_profile_overhead(env._env, _lib.current_stream(), C.byref(ov)))     # what the event pair reads beyond the kernel (20 us spin calibration)
ms = sorted(buf[i] for i in range(cnt.value))
print(f"k_env_step N={N}: median {ms[len(ms)//2]*1e3:.1f} us  min {ms[0]*1e3:.1f} us  mean {sum(ms)/len(ms)*1e3:.1f} us   (raw event pair; its calibrated overhead reads {ov.value * 1e3:.1f} us)")
if hasattr(dbg, "pbhc_debug_read_stamps"):
    st = (C.c_ulonglong * 64)()
    dbg.pbhc_debug_read_stamps(st, 64)
    # role A (thread 0 of workgroup 0): stamps 0..12 at its phase boundaries; role B (thread 128): stamps 32+1..32+5
    namesA = ["loads issued (frame, constants)", "wait for the frame + rigid-body chain + root scalars", "wait bar1", "E body diffs + termination", "wait bar2", "F reward", "G reset", "H features + J write-back",
              "rows of role 0 (none: hist_b) + wait bar3", "(late rows of a reset env)", "-", "partials"]
    tot = st[12] - st[0]
    print("  role A (dynamics chain):")
    for i, n in enumerate(namesA):
        dd = st[i + 1] - st[i]
        print(f"    {n:56s} {dd:8d} cyc  {100.0 * dd / tot:5.1f}%")
    print(f"    total {tot} cycles")
    if st[20]:
        print(f"    inside the chain: wait for the frame + segment of every body (one sincos + quaternion product per lane) {st[20] - st[1]}, pointer-jumping rounds (walk: the chain walk) {st[21] - st[20]}, "
              f"extended bodies + stores {st[22] - st[21]}, root-state scalars + the Philox noise base {st[2] - st[22]}")
        print(f"    inside E: body loop {st[23] - st[3]}, reductions {st[24] - st[23]}, lane-0 means + termination {st[4] - st[24]}")
        print(f"    inside F: exps {st[25] - st[5]}, term values {st[26] - st[25]}, episode sums {st[27] - st[26]}, reward row + err {st[6] - st[27]}")
    namesB = ["prologue: every load back, contacts in LDS", "D reference frame (lerp / slerp of the two rows)", "history request, pre-physics step + torques, reference time (then bar1)",
              "joint-space sums, foot norms, no-reset features (from bar1)", "history -> LDS, reset draws, every observation row (from bar2)"]
    refs = [st[0], st[32 + 1], st[32 + 2], st[3], st[5]]
    print("  role B (reference / observations):")
    for i, n in enumerate(namesB):
        dd = st[32 + i + 1] - refs[i]
        print(f"    {n:80s} ends {st[32 + i + 1] - st[0]:8d} cyc after start  (+{dd})")

    if st[32 + 6]:
        print(f"  role B prologue detail (cycles after start): wave entered its role {st[32 + 9] - st[0]}, tensor addresses in SGPRs {st[32 + 10] - st[0]}, independent loads issued {st[32 + 6] - st[0]}, env scalars back + rows issued {st[32 + 7] - st[0]}, "
              f"torque-noise Philox done {st[32 + 8] - st[0]}, contacts in LDS (all prologue loads back) {st[32 + 1] - st[0]}")

if hasattr(dbg, "pbhc_debug_read_wg_times"):
    nwg = (N + 3) // 4
    wt = (C.c_ulonglong * (2 * nwg))()
    dbg.pbhc_debug_read_wg_times(wt, nwg)
    import statistics
    t0 = min(wt[2 * i] for i in range(nwg))
    starts = sorted((wt[2 * i] - t0) / 100.0 for i in range(nwg))                     # 100 MHz clock -> us
    ends = sorted((wt[2 * i + 1] - t0) / 100.0 for i in range(nwg))
    durs = sorted((wt[2 * i + 1] - wt[2 * i]) / 100.0 for i in range(nwg))
    q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))]
    print(f"  workgroups {nwg}: entry  min {starts[0]:.2f} median {q(starts, .5):.2f} p99 {q(starts, .99):.2f} max {starts[-1]:.2f} us after the first")
    print(f"                  exit   min {ends[0]:.2f} median {q(ends, .5):.2f} p99 {q(ends, .99):.2f} max {ends[-1]:.2f} us")
    print(f"                  in-kernel time per workgroup: min {durs[0]:.2f} median {q(durs, .5):.2f} p99 {q(durs, .99):.2f} max {durs[-1]:.2f} us")
    if hasattr(dbg, "pbhc_debug_read_stamps"):
        d0 = (wt[1] - wt[0]) / 100.0
        print(f"                  workgroup 0 (the one with the cycle stamps): {d0:.2f} us in the kernel for {st[12] - st[0]} shader cycles between its first and last stamp -> shader clock >= {(st[12] - st[0]) / d0:.0f} MHz")
    rb = env.reset_buf.cpu().view(-1)
    has = [bool(rb[4 * i:4 * i + 4].any()) for i in range(nwg)]
    d_all = [(wt[2 * i + 1] - wt[2 * i]) / 100.0 for i in range(nwg)]
    a = [d for d, h in zip(d_all, has) if h]; b = [d for d, h in zip(d_all, has) if not h]
    if a and b:
        print(f"                  workgroups with a resetting env: {len(a)} (mean {sum(a) / len(a):.2f} us, max {max(a):.2f}); without: {len(b)} (mean {sum(b) / len(b):.2f} us, max {max(b):.2f})")
