import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, pcgan_amd
from pcgan_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, O, I) in [(64, 32, 38), (64, 32, 32), (4096, 32, 38), (4096, 128, 64), (300, 30, 32), (4096, 1, 128)]:
    dy = torch.randn(B, O, device=dev) * 0.01; x = torch.rand(B, I, device=dev)
    ref = dy.double().T @ x.double(); refb = dy.double().sum(0)
    dW = torch.zeros(O, I, device=dev); db = torch.zeros(O, device=dev)
    ops.linear_wgrad_grouped([(dy, x, O, I, dW, db, O, I, False, False)], B, dev)
    dW1 = torch.zeros(O, I, device=dev); db1 = torch.zeros(O, device=dev)
    ops.linear_wgrad(dy, x, B, O, I, dW1, db1)
    cpu = (dy.cpu().T @ x.cpu())
    sc = ref.abs().max().item()
    print(B, O, I, "mfma err/scale", ((dW.double() - ref).abs().max() / sc).item(), "bias", ((db.double() - refb).abs().max() / refb.abs().max()).item(),
          "valu", ((dW1.double() - ref).abs().max() / sc).item(), "cpu32", ((cpu.double() - ref.cpu()).abs().max() / sc).item())

# the generator's 29 layers in one call (strided operands, shared inputs), batch 64 and 4096
for B in (64, 4096):
    T, K = 70, 38
    inp = torch.rand(B, K, device=dev); cond = inp[:, 17:]
    H = torch.rand(6, B, 32, device=dev); A1 = torch.rand(5, B, 32, device=dev)
    mk = lambda *s: torch.randn(*s, device=dev) * 0.01
    DZIN, DZ1, DZ2, DG, DB, DC, DL = mk(B, 32), mk(5, B, 32), mk(5, B, 32), mk(5, B, 32), mk(5, B, 32), mk(B, 10), mk(B, T)
    seg = [0, 9, 39, 45, 47, 52, 57, 70]
    items, refs = [], []
    def add(dy, x, O, I, ldy=None, ldx=None):
        dW = torch.zeros(O, I, device=dev); db = torch.zeros(O, device=dev)
        items.append((dy, x, O, I, dW, db, ldy or O, ldx or I, True, True))
        refs.append((dy[:, :O].double().T @ x[:, :I].double(), dy[:, :O].double().sum(0)))
    add(DZIN, inp, 32, 38)
    for k in range(5):
        add(DZ1[k], H[k], 32, 32); add(DZ2[k], A1[k], 32, 32); add(DG[k], cond, 32, 21, ldx=K); add(DB[k], cond, 32, 21, ldx=K)
    add(DC, H[5], 10, 32)
    for s_ in range(7):
        add(DL[:, seg[s_]:], H[5], seg[s_ + 1] - seg[s_], 32, ldy=T)
    ops.linear_wgrad_grouped(items, B, dev)
    worst = 0.0
    for i, (it, (rw, rb)) in enumerate(zip(items, refs)):
        ew = ((it[4].double() - rw).abs().max() / rw.abs().max()).item(); eb = ((it[5].double() - rb).abs().max() / rb.abs().max()).item()
        worst = max(worst, ew, eb)
        if ew > 2e-6 or eb > 2e-6:
            print("B", B, "layer", i, "O,I", it[2], it[3], "err W", ew, "err b", eb)
    print("B", B, "29 layers: worst error / scale", worst)
