"""(GPU box) pbf_comm_create_rccl with ONE rank, in a fresh process, with and without torch's process group."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "torch":  # like bench.py: torch (and its bundled HIP runtime + librccl) is in the process FIRST
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
import bench
pkg = bench.load_package()
L = pkg.lib()
ident = (C.c_uint8 * 128)()
print("unique_id rc", L.pbf_comm_unique_id(ident), flush=True)
comm = C.c_void_p()
rc = L.pbf_comm_create_rccl(ident, 1, 0, 0, C.byref(comm))
print("create rc", rc, (L.pbf_comm_last_error(None) or b"").decode(), flush=True)
if rc == 0:
    L.pbf_comm_destroy(comm)
    print("ok")
