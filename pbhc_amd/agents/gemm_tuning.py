"""GEMM solution selection for the MLP forward/backward (plain library GEMMs: hipBLASLt / rocBLAS through PyTorch-ROCm).

PyTorch's TunableOp picks, per GEMM shape, the fastest hipBLASLt / rocBLAS solution.  `pbhc_amd/tuning/tunableop_gfx950.csv` holds the
selections measured on MI355X for the shapes of the shipped configurations (4096 envs: 4096-row rollout GEMMs, 24576-row minibatch GEMMs
and their dgrad / wgrad forms); a shape that is not in the file is tuned once, online, the first time it runs (a few hundred ms).
On MI355X this takes the fp32 update GEMMs from ~45 % to ~85 % of the 157 TFLOP/s MFMA peak (critic 768x24576x630: 0.179 ms).
The shipped file is copied to a per-process scratch file so that online tuning never writes into the source tree.
`PBHC_GEMM_TUNING=0` switches the whole thing off (PyTorch's default heuristics); a user-set PYTORCH_TUNABLEOP_ENABLED is respected.
"""
from __future__ import annotations

import os
import shutil
import tempfile

_done = False


def enable():
    global _done
    if _done or os.environ.get("PBHC_GEMM_TUNING", "1") == "0" or "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return
    _done = True
    import torch.cuda.tunable as tn

    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning", "tunableop_gfx950.csv")
    dst = os.environ.get("PBHC_GEMM_TUNING_FILE") or os.path.join(tempfile.gettempdir(), f"pbhc_tunableop_{os.getpid()}.csv")   # maintainers: collect new selections
    if os.path.exists(src) and not (os.environ.get("PBHC_GEMM_TUNING_FILE") and os.path.exists(dst)):
        shutil.copyfile(src, dst)
    tn.enable(True)
    tn.set_filename(dst, insert_device_ordinal=False)
    tn.tuning_enable(True)
    tn.set_max_tuning_duration(30)
    tn.set_max_tuning_iterations(100)
