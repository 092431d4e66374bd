import ctypes, os, sys
here = os.path.dirname(os.path.abspath(__file__))
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
if mode == "torch":
    import torch
    print("torch", torch.__version__, "cuda avail", torch.cuda.is_available())
lib = ctypes.CDLL(os.path.join(here, "libprobe.so"))
buf = ctypes.create_string_buffer(1024)
print("info rc", lib.probe_info(buf, 1024), buf.value.decode())
print("selftest (0=ok):", lib.probe_selftest())
if mode == "torch":
    a = torch.arange(64, dtype=torch.float32, device="cuda").contiguous()
    b = (torch.arange(64, dtype=torch.float32, device="cuda") * 0.5 + 1).contiguous()
    c = torch.zeros(1024, dtype=torch.float32, device="cuda")
    rc = lib.probe_mfma(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(c.data_ptr()))
    ref = a.view(32, 2) @ b.view(2, 32)
    print("torch-mem mfma rc", rc, "maxdiff", (c.view(32, 32) - ref).abs().max().item())
print("rccl (0=ok):", lib.probe_rccl())
os.system("grep -E 'rccl|amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
