import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simple_tad_amd import kernels as K
torch.manual_seed(0)
import itertools
for PIPE, (B, N, H) in itertools.product((1, 2), [(1, 96, 1), (1, 100, 1), (2, 197, 3), (2, 320, 2), (1, 40, 2), (4, 1568, 12), (32, 1568, 12)]):
    qkv = (torch.randn(B * N, 3 * H * 64, device="cuda")).bfloat16()
    ao, lse, lo = K.attn_fwd(qkv, B, N, H, 0.125, want_lo=True)
    d_ao = torch.randn_like(ao)
    K.attn_tuning(dkv_pipe=0)
    ref = K.attn_bwd(qkv, ao, d_ao, lse, B, N, H, 0.125, out_lo=lo).float().view(B, N, 3, H, 64)
    K.attn_tuning(dkv_pipe=PIPE)
    got = K.attn_bwd(qkv, ao, d_ao, lse, B, N, H, 0.125, out_lo=lo).float().view(B, N, 3, H, 64)
    K.attn_tuning(dkv_pipe=0)
    torch.cuda.synchronize()
    for i, nm in ((1, "dK"), (2, "dV")):
        e = (got[:, :, i] - ref[:, :, i])
        perkey = e.abs().amax(dim=(0, 2, 3)) / ref[:, :, i].abs().max()
        print(PIPE, (B, N, H), nm, "rel-l2 %.3e" % (e.norm() / ref[:, :, i].norm()).item(), "finite", bool(torch.isfinite(got[:, :, i]).all()),
              "per-key-block(32) max:", [round(float(perkey[k:k + 32].max()), 3) for k in range(0, N, 32)])
        perd = e.abs().amax(dim=(0, 1, 2)) / ref[:, :, i].abs().max()
        print("     per-d(8):", [round(float(perd[k:k + 8].max()), 3) for k in range(0, 64, 8)])
