"""Resident workgroups per CU of a config-specialised k_env_step build (hipOccupancyMaxActiveBlocksPerMultiprocessor): python3 tools/probes/occupancy_probe.py <spec .so>"""
import sys
import ctypes as C, torch
torch.zeros(1, device="cuda")
hip = C.CDLL("libamdhip64.so")
l = C.CDLL(sys.argv[1])
l.pbhc_spec_kernel.restype = C.c_void_p
k = C.c_void_p(l.pbhc_spec_kernel())
for lds in (27072, 26624, 26112, 32000, 20000):
    n = C.c_int(0)
    rc = hip.hipOccupancyMaxActiveBlocksPerMultiprocessor(C.byref(n), k, 256, C.c_size_t(lds))
    print("lds", lds, "rc", rc, "blocks/CU", n.value)
