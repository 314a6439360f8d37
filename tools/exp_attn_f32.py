import torch, time, sys
sys.path.insert(0, "/root/repo")
from simple_tad_amd import kernels as K
for d, H in ((64, 12), (80, 16)):
    B, N = 32, 1568
    qkv = torch.randn(B * N, 3 * H * d, device="cuda")
    dout = torch.randn(B * N, H * d, device="cuda")
    out, lse = K.attn_fwd_f32(qkv, B, N, H, d ** -0.5, want_lse=True, d=d)
    def t(fn, n=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    f = t(lambda: K.attn_fwd_f32(qkv, B, N, H, d ** -0.5, want_lse=True, d=d))
    b = t(lambda: K.attn_bwd_f32(qkv, out, dout, lse, B, N, H, d ** -0.5, d=d))
    fl = 4.0 * B * H * N * N * d
    print(f"d={d} H={H}: fwd {f:.2f} ms = {fl / f / 1e9:.1f} TF/s; bwd {b:.2f} ms = {2 * fl / b / 1e9:.1f} TF/s (algorithmic)")
