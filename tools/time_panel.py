"""The QR panel entry point on its own: the ops.qr_panel block of bench_ops.py (HBM fraction per batch shape)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench_ops  # noqa: E402
from nd4js_amd import _lib, dev  # noqa: E402
h = _lib.handle(0)
print(json.dumps(bench_ops.qr_panel(h, dev), indent=1))
