"""What a cross-stream edge costs inside a replayed HIP graph on this part: N tiny dependent kernels captured (a) on one stream,
(b) alternating between two streams with a wait at every hand-over (N cross-queue edges), (c) two independent chains of N/2 on two
streams (fork + join only).  Prints host enqueue time and wall time per replay."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, pcgan_amd
from pcgan_amd import ops
dev = torch.device("cuda:0")
N = 64
bufs = [torch.zeros(256, device=dev) for _ in range(2)]


def chain_one():
    for i in range(N):
        ops.fill(bufs[0], float(i))


def chain_alternating(s2):
    main = torch.cuda.current_stream()
    for i in range(N):
        if i % 2:
            s2.wait_stream(main)
            with torch.cuda.stream(s2):
                ops.fill(bufs[0], float(i))
            main.wait_stream(s2)
        else:
            ops.fill(bufs[0], float(i))


def chain_two(s2):
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        for i in range(N // 2):
            ops.fill(bufs[1], float(i))
    for i in range(N // 2):
        ops.fill(bufs[0], float(i))
    main.wait_stream(s2)


s2 = torch.cuda.Stream()
for name, fn in (("one stream", chain_one), ("alternating streams (an edge per kernel)", lambda: chain_alternating(s2)),
                 ("two independent chains (fork + join)", lambda: chain_two(s2))):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"{N} tiny kernels, {name}: host {host / 200 * 1e6:.1f} us/replay, wall {wall / 200 * 1e6:.1f} us/replay = {wall / 200 / N * 1e6:.2f} us per kernel")
