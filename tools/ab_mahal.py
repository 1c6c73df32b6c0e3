"""A/B timing of cgps_mahal_logdet between two builds of the library on the same box:
   python tools/ab_mahal.py LIB_A LIB_B [rows d dtype]   (direct ctypes: only the symbols both builds have)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclic-gps_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import _util  # noqa: E402

libs = sys.argv[1:3]
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
d = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dtype = torch.float32 if (len(sys.argv) <= 5 or sys.argv[5] == "f32") else torch.float64
rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
Rs, Os, b, _, _ = _util.conditioned_system(rows, d, dtype=dtype, device="cuda")
out = torch.zeros(2, dtype=torch.float64, device="cuda")
info = torch.zeros(1, dtype=torch.int32, device="cuda")
vp = ctypes.c_void_p
handles = []
for path in libs:
    L = ctypes.CDLL(os.path.abspath(path))
    L.cgps_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
    L.cgps_mahal_logdet.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp, vp, vp]
    n = ctypes.c_size_t(0)
    assert L.cgps_workspace_bytes(rows, d, 0 if dtype == torch.float32 else 1, 0, ctypes.byref(n)) == 0
    ws = torch.empty(n.value, dtype=torch.uint8, device="cuda")
    handles.append((path, L, ws, n.value))
sp = vp(torch.cuda.current_stream().cuda_stream)


def run(h, reps):
    _, L, ws, nb = h
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        rc = L.cgps_mahal_logdet(vp(Rs.data_ptr()), vp(Os.data_ptr()), vp(b.data_ptr()), rows, d,
                                 0 if dtype == torch.float32 else 1, vp(ws.data_ptr()), nb, vp(out.data_ptr()),
                                 vp(info.data_ptr()), sp)
        assert rc == 0
    c.record()
    torch.cuda.synchronize()
    return a.elapsed_time(c) / reps * 1e3


for h in handles:
    run(h, 5)
for trial in range(4):
    print("  ".join("%s %.1f us" % (os.path.basename(h[0]), run(h, 30)) for h in handles), " [%s]" % out.tolist(), flush=True)
