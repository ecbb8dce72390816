#!/usr/bin/env python3
"""Per-layer timing of the conv family at the DCGAN bs-512 shapes (diagnostic; not the bench contract).

  python scripts/conv_microbench.py [--batch 512] [--iters 20] [--only fwd,dgrad,wgrad] [--layers D2,D3,...]
Prints one line per (layer, op): average ms over `iters` launches (HIP events on the launch stream) and TFLOP/s.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pcgan_amd  # noqa: E402
from pcgan_amd import ops  # noqa: E402

# name: (Cin, Cout, H_in(conv input), k, s, p)  -- geometry of the (adjoint) conv
LAYERS = {
    "D1": (1, 64, 64, 4, 2, 1), "D2": (64, 128, 32, 4, 2, 1), "D3": (128, 256, 16, 4, 2, 1), "D4": (256, 512, 8, 4, 2, 1),
    "D5": (512, 1, 4, 4, 1, 0),
    "G1": (512, 100, 4, 4, 1, 0), "G2": (256, 512, 8, 4, 2, 1), "G3": (128, 256, 16, 4, 2, 1), "G4": (64, 128, 32, 4, 2, 1),
    "G5": (1, 64, 64, 4, 2, 1),
    "R64": (64, 64, 28, 3, 1, 1),
    # tile-overhead probes: the same N = 64 output with longer K (9 x Cin), and N = 128 with the short K
    "R128x64": (128, 64, 28, 3, 1, 1), "R256x64": (256, 64, 28, 3, 1, 1), "R64x128": (64, 128, 28, 3, 1, 1),
    # counteRGAN thin layers: conv_in 3->64, conv_out 64->1, discriminator entry 2->64 (s2)
    "CI": (3, 64, 28, 3, 1, 1), "CO": (64, 1, 28, 3, 1, 1), "CD": (2, 64, 28, 3, 2, 1),
    # WGAN-GP (mnist_wgan_conditional.py:51-108) at width 1024: critic convs, critic Linear 8192->1024, generator ConvT adjoints
    "WC2": (256, 512, 13, 3, 2, 0), "WC3": (512, 1024, 6, 3, 2, 0), "WL1": (8192, 1024, 1, 1, 1, 0),
    "WG1": (1024, 1024, 4, 4, 1, 0), "WG2": (512, 1024, 7, 3, 2, 1), "WG3": (256, 512, 14, 4, 2, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--layers", default="D2,D3,D4,G2,G3,G4")
    ap.add_argument("--xf", action="store_true", help="fwd / wgrad with an input transform on x (the fold_bn_apply form)")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands (DVFS probe: not a performance number)")
    ap.add_argument("--ab", default=None, help="A/B a tuning switch in THIS process, interleaved rounds: e.g. korder=0,1 or wgrad_order=0,1 "
                                               "(ops.tune / pcg_tune_set); prints the median and min ms per variant")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--timeline", type=int, default=0, help="with --clock: print the per-tile timeline (loop end / epilogue end, us since the "
                                                            "stream began) of this many workgroups of the persistent kernels")
    ap.add_argument("--clock", action="store_true", help="report the in-kernel clock (needs the stamp build: make -C csrc stamp; "
                                                         "PCG_LIB=.../csrc/build_stamp/libpcgan_hip.so)")
    args = ap.parse_args()
    lib = pcgan_amd.load()
    dev = torch.device("cuda:0")
    stamps = None
    if args.clock:
        stamps = torch.zeros(4 * 65536, dtype=torch.int64, device=dev)
        if lib.pcg_debug_stamp_buffer(stamps.data_ptr(), stamps.numel() * 8) != 1:
            sys.exit("--clock: this libpcgan_hip.so has no stamp code; make -C promptable-counterfactual-gan_amd/csrc stamp and set PCG_LIB")

    def clock_mhz():
        half = stamps.numel() // 2
        if args.timeline:
            tl = stamps[half:].cpu().view(-1)
            nb = 0
            t00 = None
            rows = []
            for b in range(min(1024, (half // 2) // 36 if False else 1024)):
                rec = tl[b * 34:(b + 1) * 34]
                if int(rec[0]) == 0:
                    continue
                rows.append((b, int(rec[0]), int(rec[1]), [int(v) / 100.0 for v in rec[2:] if int(v) > 0]))
            if rows:
                t00 = min(r[1] for r in rows)
                for b, r0, hw, ev in rows[:args.timeline]:
                    cu = (hw >> 8) & 15; se = (hw >> 13) & 7; tg = (hw >> 16) & 15; simd = (hw >> 4) & 3
                    print(f"    wg {b:4d} start +{(r0 - t00) / 100.0:6.1f} us hw_id 0x{hw:08x} (se {se} cu {cu} tg {tg}) | " +
                          " ".join(f"{v:.1f}" for v in ev))
        ph = stamps[half:half + half // 2].cpu().view(-1, 4).double()
        ph = ph[(ph[:, 0] > 0) & (ph[:, 3] > 0)]
        clock_mhz.phases = None
        if ph.shape[0] > 0:
            t0 = ph[:, 0].min()
            med = lambda v: float(v.median()) / 100.0
            clock_mhz.phases = (med(ph[:, 0] - t0), med(ph[:, 1] - ph[:, 0]), med(ph[:, 2] - ph[:, 1]), med(ph[:, 3] - ph[:, 2]),
                                float(ph[:, 3].max() - t0) / 100.0, float((ph[:, 3] - t0).median()) / 100.0)
        bw = stamps[half + half // 2:].cpu().view(-1, 4).double()
        bw = bw[(bw[:, 1] > 0) & (bw[:, 3] > 0)]
        clock_mhz.barwait = None
        if bw.shape[0] > 0:
            q3 = half + half // 2
            secraw = stamps[q3 + half // 4:].cpu().view(-1, 2)[:bw.shape[0]]
            wait_only = (secraw[:, 0] >> 32).double()
            sec = torch.stack([(secraw[:, 0] & 0xFFFFFFFF).double(), secraw[:, 1].double()], 1)
            loopc = stamps[q3:q3 + half // 4].cpu().view(-1, 4).double()
            loopc = loopc[(loopc[:, 1] > 0) & (loopc[:, 3] > 0)][:sec.shape[0], 3]
            n = min(sec.shape[0], loopc.shape[0])
            clock_mhz.barwait = (float((bw[:, 0] / bw[:, 1]).median()), float((bw[:, 2] / bw[:, 3]).median()),
                                 float((sec[:n, 0] / loopc[:n]).median()) if n else float("nan"), float((sec[:n, 1] / loopc[:n]).median()) if n else float("nan"),
                                 float((wait_only[:n] / loopc[:n]).median()) if n else float("nan"))
        st = stamps[:half].view(-1, 2).cpu().double()
        st = st[st[:, 1] > 0]
        stamps.zero_()
        if st.numel() == 0:
            return float("nan")
        loop_us = st[:, 1] / 100.0                     # 100 MHz ticks -> us: consumer wave 0's main loop, per block
        clock_mhz.loop = (float(loop_us.median()), float(loop_us.min()), float(loop_us.max()), int(st.shape[0]))
        return float((st[:, 0] / st[:, 1] * 100.0).median())
    B = args.batch
    tot_f = tot_t = 0.0
    for name in args.layers.split(","):
        Cin, Cout, H, k, s, p = LAYERS[name]
        g = ops.conv_geom(B, H, H, Cin, Cout, k, k, s, p)
        x = torch.randn(B, H, H, Cin, device=dev)
        w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
        dy = torch.randn(B, g.OH, g.OW, Cout, device=dev)
        if args.zeros:
            x.zero_(); w.zero_(); dy.zero_()
        y = torch.empty(B, g.OH, g.OW, Cout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        flops = 2.0 * B * g.OH * g.OW * Cout * k * k * Cin
        fns = {"fwd": lambda: ops.conv2d_fwd(g, x, w, None, out=y), "dgrad": lambda: ops.conv2d_dgrad(g, dy, w, None, out=dx),
               "wgrad": lambda: ops.conv2d_wgrad(g, x, dy, dw, False)}
        if args.xf:      # the same launches with an input transform (folded BatchNorm + LeakyReLU) on the activation operand x
            xf = ops.InputXform(torch.cat([torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1]).contiguous(), ops.ACT_LRELU, 0.2)
            fns = {"fwd": lambda: ops.conv2d_fwd(g, x, w, None, out=y, xf=xf), "wgrad": lambda: ops.conv2d_wgrad(g, x, dy, dw, False, xf_x=xf)}
        for op in args.only.split(","):
            fn = fns[op]
            if args.ab:
                key, vals = args.ab.split("=")
                vals = [int(v) for v in vals.split(",")]
                res = {v: [] for v in vals}
                clk = {}
                for v in vals:
                    ops.tune(key, v)
                    for _ in range(2):
                        fn()
                for _ in range(args.rounds):
                    for v in vals:
                        ops.tune(key, v)
                        fn()
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(args.iters):
                            fn()
                        e1.record()
                        torch.cuda.synchronize()
                        res[v].append(e0.elapsed_time(e1) / args.iters)
                        if stamps is not None:
                            clk.setdefault(v, []).append(clock_mhz())
                ops.tune(key, -1)
                line = f"{name:4s} {op:6s} B={B} {Cin:4d}->{Cout:4d} {H:3d}x{H:<3d} k{k}s{s}p{p} "
                for v in vals:
                    r = sorted(res[v])
                    med = r[len(r) // 2]
                    line += f" | {key}={v}: med {med:7.4f} ms ({flops / med / 1e9:6.1f} TF) min {r[0]:7.4f}"
                    if stamps is not None:
                        c = sorted(clk[v])
                        line += f" clk {c[len(c) // 2]:5.0f} MHz"
                print(line, flush=True)
                continue
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            tot_f += flops; tot_t += ms * 1e-3
            extra = ""
            if stamps is not None:
                c = clock_mhz()
                lo = clock_mhz.loop
                extra = f"  in-kernel clock {c:5.0f} MHz; main loop per block med/min/max {lo[0]:.1f}/{lo[1]:.1f}/{lo[2]:.1f} us over {lo[3]} blocks"
                if clock_mhz.barwait:
                    extra += (f"\n        share of the main loop spent in the hand-over barrier: consumer wave 0 {100 * clock_mhz.barwait[0]:.1f} %, "
                              f"producer wave 4 {100 * clock_mhz.barwait[1]:.1f} % (waiting for the gathers {100 * clock_mhz.barwait[4]:.1f} %, its ds_writes {100 * clock_mhz.barwait[2]:.1f} %, "
                              f"issuing the next gathers {100 * clock_mhz.barwait[3]:.1f} %)")
                if clock_mhz.phases:
                    q = clock_mhz.phases
                    extra += (f"\n        phases (us, medians over blocks of the LAST launch): entry skew {q[0]:.1f}, entry->loop {q[1]:.1f}, loop {q[2]:.1f}, "
                              f"epilogue {q[3]:.1f}; last block done at {q[4]:.1f} (median block at {q[5]:.1f}) after the first entry")
            print(f"{name:4s} {op:6s} M/N/K-ish B={B} {Cin:4d}->{Cout:4d} {H:3d}x{H:<3d} k{k}s{s}p{p}  {ms:8.4f} ms  {flops / ms / 1e9:7.2f} TFLOP/s{extra}", flush=True)
    if tot_t > 0:
        print(f"TOTAL {tot_f / tot_t / 1e12:.2f} TFLOP/s over {tot_t * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
