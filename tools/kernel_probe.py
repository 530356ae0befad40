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
cfg, env, _ = build(N, "cuda:0", 0)
env.reset_all()
env.simulator.set_replay(*make_replay_on_device(env, 130, 1))
lib = _lib.lib()
act = torch.zeros(N, env.num_dof, device="cuda:0")
for _ in range(20):
    env.step({"actions": act})
_lib.check(lib.pbhc_env_profile(env._env, 1))
for _ in range(100):
    env.step({"actions": act})
buf = (C.c_float * 512)(); cnt = C.c_int(0)
_lib.check(lib.pbhc_env_profile_read(env._env, buf, 100, C.byref(cnt)))
ms = sorted(buf[i] for i in range(cnt.value))
print(f"k_env_step N={N}: median {ms[len(ms)//2]*1e3:.1f} us  min {ms[0]*1e3:.1f} us  mean {sum(ms)/len(ms)*1e3:.1f} us")
if hasattr(lib, "pbhc_debug_read_stamps"):
    st = (C.c_ulonglong * 32)()
    lib.pbhc_debug_read_stamps(st, 32)
    names = ["A load+torque", "B fk", "C scalars", "D lookup", "E diffs", "F reward", "G reset", "H features", "I obs", "J writeback", "partials"]
    tot = st[11] - st[0]
    for i, n in enumerate(names):
        d = st[i + 1] - st[i]
        print(f"  {n:16s} {d:8d} cyc  {100.0 * d / tot:5.1f}%")
    print(f"  total {tot} cycles")
    if st[12]:
        prev = st[8]
        for g in range(3):
            print(f"  obs group {g}: pass1 {st[12 + 2 * g] - prev} cyc, noisy pass {st[13 + 2 * g] - st[12 + 2 * g]} cyc")
            prev = st[13 + 2 * g]
        if st[20]:
            print(f"  group 1 first batch: map reads {st[20] - st[13]} cyc, feature+segment reads {st[21] - st[20]} cyc, scale/clip/stores {st[22] - st[21]} cyc")

if hasattr(lib, "pbhc_debug_read_wg_times"):
    nwg = (N + 3) // 4
    wt = (C.c_ulonglong * (2 * nwg))()
    lib.pbhc_debug_read_wg_times(wt, nwg)
    import statistics
    t0 = min(wt[2 * i] for i in range(nwg))
    starts = sorted((wt[2 * i] - t0) / 100.0 for i in range(nwg))                     # 100 MHz clock -> us
    ends = sorted((wt[2 * i + 1] - t0) / 100.0 for i in range(nwg))
    durs = sorted((wt[2 * i + 1] - wt[2 * i]) / 100.0 for i in range(nwg))
    q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))]
    print(f"  workgroups {nwg}: entry  min {starts[0]:.2f} median {q(starts, .5):.2f} p99 {q(starts, .99):.2f} max {starts[-1]:.2f} us after the first")
    print(f"                  exit   min {ends[0]:.2f} median {q(ends, .5):.2f} p99 {q(ends, .99):.2f} max {ends[-1]:.2f} us")
    print(f"                  in-kernel time per workgroup: min {durs[0]:.2f} median {q(durs, .5):.2f} p99 {q(durs, .99):.2f} max {durs[-1]:.2f} us")
    rb = env.reset_buf.cpu().view(-1)
    has = [bool(rb[4 * i:4 * i + 4].any()) for i in range(nwg)]
    d_all = [(wt[2 * i + 1] - wt[2 * i]) / 100.0 for i in range(nwg)]
    a = [d for d, h in zip(d_all, has) if h]; b = [d for d, h in zip(d_all, has) if not h]
    if a and b:
        print(f"                  workgroups with a resetting env: {len(a)} (mean {sum(a) / len(a):.2f} us, max {max(a):.2f}); without: {len(b)} (mean {sum(b) / len(b):.2f} us, max {max(b):.2f})")
