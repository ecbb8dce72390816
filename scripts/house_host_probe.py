import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, pcgan_amd
from pcgan_amd import house as H, ops
dev = torch.device("cuda:0")
G, D, C = H.build(dev, seed=0)
opt_g, opt_d = H.make_optimizers(G, D)
norm = H.cat_norm_maps(G, H.CONFIG, dev)
B = 4096
ov = {"1": True, "0": False, "c": "critic", "i": "inline"}[os.environ.get("OV", "1")]
gs = H.GraphedTrainStep(G, D, C, opt_g, opt_d, norm, B, overlap=ov)
for _ in range(20): gs.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): gs.replay()
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"OV={ov} PKT={os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE')}: host enqueue {host/200*1e6:.1f} us/replay, wall {tot/200*1e6:.1f} us/replay")
