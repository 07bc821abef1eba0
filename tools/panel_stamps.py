"""Phase stamps of workgroup 0 of the batched QR panel kernel (ND4HIP_QRB_STAMPS=1): one line per call on stderr."""
import ctypes, os, sys
os.environ["ND4HIP_QRB_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nd4js_amd import _lib, dev  # noqa: E402
h = _lib.handle(0)
shapes = [(256, 2048), (2048, 512)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for nb, rows in shapes:
    A = dev.fill_uniform(21, (nb, rows, 16))
    V = torch.empty_like(A)
    T = torch.empty((nb, 16, 16), dtype=torch.float64, device="cuda")
    W = A.clone()
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(6):
        W.copy_(A)
        _lib.check(h.lib.nd4hip_dgeqr2_panel_batched_dev(h.ptr, nb, rows, 16, ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(V.data_ptr()),
                                                         ctypes.c_void_p(T.data_ptr())))
    torch.cuda.synchronize()
