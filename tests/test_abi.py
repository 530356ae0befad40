"""The C-ABI library loads and exports every symbol include/pbhc_hip.h declares (no compute, no GPU)."""
import ctypes as C
import os
import re

from pbhc_amd import _lib


def test_header_symbols_are_exported():
    text = _lib._strip_comments(open(_lib.HEADER).read())
    declared = set(re.findall(r"\b(pbhc_\w+)\s*\(", text))
    assert declared, "no declarations parsed"
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in pbhc_hip.h but not exported"
    assert declared == set(_lib.EXPORTS)


def test_struct_sizes_agree():
    lib = _lib.lib()
    assert lib.pbhc_sizeof_env_config() == C.sizeof(_lib.PbhcEnvConfig)
    assert lib.pbhc_sizeof_step_io() == C.sizeof(_lib.PbhcStepIO)
    assert lib.pbhc_abi_version() == _lib.K["PBHC_ABI_VERSION"]


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = _lib.lib()
    assert lib.pbhc_env_step(None, None, None) == _lib.K["PBHC_EINVAL"]
    sk = _lib.PbhcSkeleton()
    sk.num_bodies = 100
    assert lib.pbhc_sim_fk(C.byref(sk), None, None, None, 1, 0, None, None) == _lib.K["PBHC_EINVAL"]
    assert b"bad argument" in lib.pbhc_last_error()


def test_product_does_not_import_the_oracle():
    root = os.path.join(_lib.ROOT, "pbhc_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"
