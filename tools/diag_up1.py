"""Diagnostic (not a test): the wide decoder's up1 / conv0 weight gradients at batch 16, q = 1 -- is the 8e-4 distance from
float64 the weight-gradient kernel's own (its inputs taken as given) or does it arrive with its inputs?"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests import test_gpu_measured_path as T
from nvfpcc_amd import ops
F = torch.nn.functional
gpu = torch.device("cuda")
T.H["w2"] = 0.0                      # no weight-rate addend: the gradient slices are pure data terms
cap = []
_add = ops.WgradBatch.add


def add(self, p, q, k, stride, pad, out_mode, out):
    cap.append((p.clone(), q.clone(), k, stride, pad, out))
    return _add(self, p, q, k, stride, pad, out_mode, out)


ops.WgradBatch.add = add
for q_mode in (1, 2):
    cap.clear()
    net, eng, P, gt, dist, emb = T.make(gpu, 8, (16, 32, 16, 16), 40)
    ids = np.random.default_rng(3).permutation(40)[:16].astype(np.int64)
    n_pts = float(eng.counts[ids].sum())
    a = eng.train_step(ids, q_mode, update=False)
    g64, = [T._oracle_step({k: v.double() for k, v in P.items()}, emb.double(), gt.double(), dist.double(), ids, q_mode,
                           n_pts, eng.noise_step, layer_ids=T._layer_ids(net))[3]]
    print(f"== q={q_mode}: captured {len(cap)} per-layer weight-gradient calls")
    for p, q, k, stride, pad, out in cap:
        if stride != 2:
            continue
        # dW[a][b][k] = sum p[n,a,i] q[n,b,2i-pad+k]  (transposed conv: p = layer input, q = output gradient)
        w = torch.zeros(p.shape[1], q.shape[1], k, k, k, dtype=torch.float64, device=gpu, requires_grad=True)
        op = q.shape[-1] - ((p.shape[-1] - 1) * 2 - 2 * pad + k)
        y = F.conv_transpose3d(p.double(), w, None, 2, pad, op)
        (y * q.double()).sum().backward()
        ref = w.grad
        mag = None
        sc = ref.abs().max().item()
        err = (out.double() - ref).abs().max().item() / sc
        name = [n for n, L in eng.layers.items() if L.gk.data_ptr() == out.data_ptr()][0]
        e2e = (out.double().cpu() - g64["reconstructor." + name + ".kernel"]).abs().max().item() / g64["reconstructor." + name + ".kernel"].abs().max().item()
        # conditioning of the worst entry: sum |terms| / |sum|
        wa = torch.zeros_like(w, requires_grad=True)
        ya = F.conv_transpose3d(p.double().abs(), wa, None, 2, pad, op)
        (ya * q.double().abs()).sum().backward()
        kk = (out.double() - ref).abs().argmax().item()
        print(f"  {name:6s} p{tuple(p.shape)} q{tuple(q.shape)}: kernel's own error (HIP inputs -> fp64) {err:.2e}; end to end vs fp64 oracle {e2e:.2e}; "
              f"worst entry: sum|terms|/max|dW| = {wa.grad.reshape(-1)[kk].item() / sc:.3g}, p zero fraction {float((p == 0).float().mean()):.2f}")
