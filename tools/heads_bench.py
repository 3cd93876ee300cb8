"""Times the three-head launches (forward / backward-data / weight gradient) of the narrow decoder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvfpcc_amd import ops


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda")
    shapes = [(16, 8), (8, 16), (8, 32)]
    xs = [torch.randn(B, c, s, s, s, device=dev) for c, s in shapes]
    ws = [torch.randn(1, c, 3, 3, 3, device=dev) for c, s in shapes]
    packs = [ops.pack_conv_weight(w) for w in ws]
    bs = [torch.randn(1, device=dev) for _ in shapes]
    dls = [torch.randn(B, 1, s, s, s, device=dev) for c, s in shapes]
    print("heads3_fwd      %.1f us" % timeit(lambda: ops.heads3_fwd(xs, [p[0] for p in packs], bs)))
    print("heads3_bwd_data %.1f us" % timeit(lambda: ops.heads3_bwd_data(dls, [p[1] for p in packs], [c for c, s in shapes], xs)))
    wb = ops.WgradBatch(dev)
    outs = [torch.empty(1, c, 3, 3, 3, device=dev) for c, s in shapes]

    def wg():
        wb.add_heads3(dls, xs, outs)
        wb.finish()
    print("heads3_wgrad+reduce %.1f us" % timeit(wg))


if __name__ == "__main__":
    main()
