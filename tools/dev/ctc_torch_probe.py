"""Dev: torch.nn.functional.ctc_loss on a per-utterance SMALL alphabet (blank + the utterance's labels) against the dense alphabet:
loss and the gradient at the gathered columns, CPU and GPU kernels."""
import torch, torch.nn.functional as func
torch.manual_seed(0)
def run(dev, B, T, L, V, uniq=True, il=None, note=""):
    g = torch.Generator().manual_seed(1)
    z = torch.randn(B, T, V, generator=g) * 0.5
    if uniq:
        tg = torch.stack([torch.randperm(V - 1, generator=g)[:L] + 1 for _ in range(B)])
    else:
        tg = torch.randint(1, min(V, 6), (B, L), generator=g)
    tl = torch.full((B,), L, dtype=torch.long); tl[0] = max(1, L - 2)
    il = torch.full((B,), T, dtype=torch.long) if il is None else il
    il[-1] = max(2 * L + 1, T - 7)
    z, tg = z.to(dev), tg.to(dev)
    zd = z.double().requires_grad_(True)
    logp = func.log_softmax(zd, -1)
    ref = func.ctc_loss(logp.transpose(0, 1), tg, il.tolist(), tl.tolist(), blank=0, reduction="mean", zero_infinity=True)
    (dz,) = torch.autograd.grad(ref, zd)
    # small alphabet
    pos = torch.arange(L, device=dev)
    same = tg.unsqueeze(2) == tg.unsqueeze(1)
    first = same.to(torch.int8).argmax(dim=2)
    classes = (first + 1).contiguous()
    cols = torch.cat([torch.zeros(B, 1, dtype=torch.long, device=dev), tg], 1)
    lp = torch.gather(func.log_softmax(z.float(), -1), 2, cols.unsqueeze(1).expand(B, T, L + 1)).contiguous()
    leaf = lp.detach().requires_grad_(True)
    nll = func.ctc_loss(leaf.transpose(0, 1), classes, il.tolist(), tl.tolist(), blank=0, reduction="none", zero_infinity=True)
    loss = (nll / tl.to(dev).clamp_min(1)).mean()
    (gs,) = torch.autograd.grad(loss, leaf)
    # contiguous variant
    leaf2 = lp.detach().transpose(0, 1).contiguous().requires_grad_(True)
    nll2 = func.ctc_loss(leaf2, classes, il.tolist(), tl.tolist(), blank=0, reduction="none", zero_infinity=True)
    (gs2,) = torch.autograd.grad((nll2 / tl.to(dev).clamp_min(1)).mean(), leaf2)
    refg = torch.gather(dz, 2, cols.unsqueeze(1).expand(B, T, L + 1))
    valid = (pos.view(1, -1) < tl.to(dev).view(-1, 1)) & (first == pos.view(1, -1))
    m = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=dev), valid], 1).unsqueeze(1).expand(B, T, L + 1)
    tm = (torch.arange(T, device=dev).view(1, -1, 1) < il.to(dev).view(-1, 1, 1)).expand(B, T, L + 1)
    m = m & tm
    rel = lambda a, b: ((a.double() - b.double())[m].norm() / b.double()[m].norm()).item()
    print("%-5s B %3d T %4d L %3d V %5d uniq %d %s: loss %.5f ref %.5f | g(view) rel %.3e | g(contig) rel %.3e" % (dev, B, T, L, V, uniq, note, float(loss), float(ref), rel(gs, refg), rel(gs2.transpose(0, 1), refg)))
import sys
for dev in sys.argv[1:]:
    run(dev, 3, 20, 6, 37, uniq=False)
    run(dev, 3, 100, 6, 37)
    run(dev, 8, 300, 20, 500)
    run(dev, 32, 1000, 50, 4337)
    run(dev, 32, 1000, 50, 4337, uniq=False)
    run(dev, 4, 1000, 50, 4337)
    run(dev, 32, 200, 50, 4337)
    run(dev, 32, 1000, 10, 4337)
