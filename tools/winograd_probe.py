#!/usr/bin/env python3
"""How much fp32 accuracy would Winograd F(2^3, 4^3) cost the 4^3 layers?  ("what is left" (1).)

Cook-Toom F(2, 4) on the points {0, 1, -1, 2, inf}: Y = A^T [(G g) * (B^T d)] gives 2 outputs of a 4-tap correlation from
5 inputs with 5 multiplications instead of 8; nested over three axes, 125 instead of 512 per (ci, co) pair (4.1 x).
This script builds the three matrices (B^T solved from the bilinear identities), checks them in fp64, then runs conv2's
shape (8 -> 8 channels, 35^3 -> 32^3) in fp32 both ways and compares each with the fp64 result.  CPU only.

    python tools/winograd_probe.py
"""
import numpy as np
import torch

pts = [0.0, 1.0, -1.0, 2.0]
m, r = 2, 4
n = m + r - 1
AT = np.zeros((m, n))
G = np.zeros((n, r))
for k, p in enumerate(pts):
    AT[:, k] = [p ** i for i in range(m)]
    G[k] = [p ** j for j in range(r)]
AT[:, n - 1] = [0.0] * (m - 1) + [1.0]
G[n - 1] = [0.0] * (r - 1) + [1.0]
# B^T from  sum_p AT[i,p] G[p,j] BT[p,k] = [k == i + j]
rows, rhs = [], []
for i in range(m):
    for j in range(r):
        for k in range(n):
            row = np.zeros((n, n))
            row[:, k] = AT[i] * G[:, j]
            rows.append(row.reshape(-1))
            rhs.append(1.0 if k == i + j else 0.0)
BT = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0].reshape(n, n)
resid = np.abs(np.array(rows) @ BT.reshape(-1) - np.array(rhs)).max()
print("F(2,4): bilinear identities hold to %.1e; |B^T| max %.2f, |G| max %.2f" % (resid, np.abs(BT).max(), np.abs(G).max()))


def winograd3d(x, w, dtype):
    """x [Cin, D, H, W] (D = H = W = 2 T + 3), w [Cout, Cin, 4, 4, 4] -> y [Cout, 2T, 2T, 2T], everything in `dtype`."""
    AT_, G_, BT_ = (torch.tensor(a, dtype=dtype) for a in (AT, G, BT))
    cout, cin = w.shape[:2]
    T = (x.shape[1] - 3) // 2
    U = torch.einsum("ai,bj,ck,ouijk->ouabc", G_, G_, G_, w.to(dtype))                       # [co, ci, 5, 5, 5]
    tiles = x.to(dtype).unfold(1, 5, 2).unfold(2, 5, 2).unfold(3, 5, 2)                       # [ci, T, T, T, 5, 5, 5]
    V = torch.einsum("ai,bj,ck,uxyzijk->uxyzabc", BT_, BT_, BT_, tiles)
    M = torch.einsum("ouabc,uxyzabc->oxyzabc", U, V)
    Y = torch.einsum("ia,jb,kc,oxyzabc->oxyzijk", AT_, AT_, AT_, M)                           # [co, T, T, T, 2, 2, 2]
    return Y.permute(0, 1, 4, 2, 5, 3, 6).reshape(cout, 2 * T, 2 * T, 2 * T)


torch.manual_seed(0)
x = torch.relu(torch.randn(8, 35, 35, 35) * 0.7)                       # a ReLU output
w = torch.round(torch.randn(8, 8, 4, 4, 4) * 0.08 * 16) / 16           # kernels on the 1/16 grid
ref = torch.nn.functional.conv3d(x.double()[None], w.double())[0]
direct = torch.nn.functional.conv3d(x[None], w)[0]
wino = winograd3d(x, w, torch.float32)
wino64 = winograd3d(x, w, torch.float64)
scale = ref.abs().max().item()
print("fp64 Winograd vs fp64 direct: %.1e (max abs / max |y|)" % ((wino64 - ref).abs().max().item() / scale))
for name, y in (("direct fp32 (aten)", direct), ("Winograd fp32", wino)):
    e = (y.double() - ref).abs()
    print("%-20s max abs err / max|y| = %.2e, rms = %.2e" % (name, e.max().item() / scale, e.pow(2).mean().sqrt().item() / scale))
