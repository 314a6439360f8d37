"""In-kernel timeline of the pipelined dK/dV kernel (build with -DTAD_PIPE_STAMPS): cycles per tile, DMA wait, barrier wait."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K, _lib
lib = _lib.load()
B, N, H = 32, 1568, 12
qkv = torch.randn(B * N, 3 * H * 64, device="cuda").bfloat16()
ao, lse, lo = K.attn_fwd(qkv, B, N, H, 0.125, want_lo=True)
d_ao = torch.randn_like(ao)
K.attn_tuning(dkv_pipe=1)
for _ in range(5):
    K.attn_bwd(qkv, ao, d_ao, lse, B, N, H, 0.125, out_lo=lo)
nwg = 7 * H * B
buf = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
assert lib.tad_attn_debug_stamps(buf.data_ptr()) == 0, lib.tad_last_error_string()
K.attn_bwd(qkv, ao, d_ao, lse, B, N, H, 0.125, out_lo=lo)
torch.cuda.synchronize()
lib.tad_attn_debug_stamps(None)
s = buf.cpu().numpy().reshape(nwg, 64).astype(np.float64)
ok = s[:, 3] > s[:, 0]
s = s[ok]
print("workgroups", len(s))
med = lambda x: float(np.median(x))
print("kernel total cycles (median)", med(s[:, 3] - s[:, 0]), " prologue", med(s[:, 1] - s[:, 0]), " loop", med(s[:, 2] - s[:, 1]), " epilogue", med(s[:, 3] - s[:, 2]))
for t in range(10):
    c = s[:, 4 + 3 * t] - (s[:, 1] if t == 0 else s[:, 6 + 3 * (t - 1)])
    print("tile", t, "compute %.0f" % med(c), "dma wait %.0f" % med(s[:, 5 + 3 * t] - s[:, 4 + 3 * t]), "barrier %.0f" % med(s[:, 6 + 3 * t] - s[:, 5 + 3 * t]))
span = (s[:, 3].max() - s[:, 0].min())
print("launch span cycles", span, "sum of WG cycles / 256 CUs", (s[:, 3] - s[:, 0]).sum() / 256)
