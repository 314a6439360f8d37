"""Run-to-run determinism of the HIP path: the same forward + backward twice, gradients compared bit for bit (per precision mode),
then the three-step MAE trajectory twice."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import simple_tad_amd as T
from simple_tad_amd import engine as E, engine_pretrain as EP, ops
import golden_recipe as R
import test_engine_trajectory as TT

def grads(mode, scale):
    m = TT.build_pretrain("cuda", torch.float32)
    x, mask = R.g13_batches()[0][:2]
    T.set_precision(mode)
    try:
        mask = torch.as_tensor(mask).cuda().flatten(1).bool()
        nm = int(mask[0].sum())
        labels = EP.reconstruction_target(x.cuda(), mask, 16, 2, True, nm)
        loss = ops.MseLossFn.apply(m(x.cuda(), mask, num_masked=nm), labels)
        (loss * scale).backward()
        torch.cuda.synchronize()
    finally:
        T.set_precision("fast")
    return {k: p.grad.detach().clone() for k, p in m.named_parameters()}, loss.item()

for mode, scale in (("fast", 1.0), ("half", 65536.0)):
    a, la = grads(mode, scale); b, lb = grads(mode, scale)
    bad = {k: float((a[k] - b[k]).abs().max() / a[k].abs().max().clamp_min(1e-30)) for k in a if not torch.equal(a[k], b[k])}
    print(mode, "loss equal", la == lb, "tensors that differ between two identical runs:", len(bad), sorted(bad.items(), key=lambda kv: -kv[1])[:6])
