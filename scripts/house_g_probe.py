"""Generator and critic passes of the tabular step in isolation (single-stream HIP-graph replay, batch 4096): us per replay."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, pcgan_amd
from pcgan_amd import house as H, ops
dev = torch.device("cuda:0")
G, D, C = H.build(dev, seed=0)
B = int(os.environ.get("B", "4096"))
rng = ops.DeviceRNG(seed=1)
x = torch.rand((B, 17), device=dev); t = torch.randint(0, 4, (B,), device=dev); m = torch.ones((B, 17), device=dev)
noise = rng.gumbel((B, G.total_cat), dev)
oh = ops.onehot(t, 4)
G._ensure_flat(); D._ensure_flat()
cot = torch.full((B, 1), 1.0 / B, device=dev)


def g_fwd():
    return G._run_forward(x, oh, m, noise, 0.5, False)


def g_fwd_bwd():
    cont, logits, samples, saved = g_fwd()
    G._run_backward(saved, cont, None, samples)


def d_fwd():
    return D._run_forward(x, oh, keep=True)


def d_fwd_bwd():
    out, sv = d_fwd()
    D._run_backward(sv, cot, True, True)


def c_fwd_bwd():
    logits, acts = C._run_forward(x, keep=True)
    g, dlog = ops.cross_entropy_fwd_bwd(logits.contiguous(), t, need_loss=True, need_grad=True, grad_scale=2.0)
    C._run_backward(acts, dlog)


for name, fn in (("G fwd", g_fwd), ("G fwd+bwd", g_fwd_bwd), ("D fwd", d_fwd), ("D fwd+bwd", d_fwd_bwd), ("C fwd+CE+bwd", c_fwd_bwd)):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3):
            G.zero_grad(); D.zero_grad(); fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        fn()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): g.replay()
    torch.cuda.synchronize()
    print(f"B={B} {name}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us/replay")
