#!/usr/bin/env python3
"""Numerical study behind DESIGN.md 3.6: would a Winograd F(2x2, 5x5) evaluation of the 5x5 convs (2.78x fewer
multiplies: 36 per 2x2 outputs instead of 100) be "no less accurate than exact fp32"?  Emulated on the CPU with every
intermediate rounded to fp32 exactly as a kernel would (input transform, weight transform, 36-point channel reduction
with fp32 accumulation, inverse transform), on the operands the network actually sees (post-ReLU activations,
He-normal weights; CODON_x4.py:50-53, :81) and on uniform weights with heavier cancellation.

Cook-Toom points {0, 1, -1, 2, -2, inf}: F(2,5) needs 2 + 5 - 1 = 6 evaluation points per dimension.
Prints per-conv rel-RMSE against an fp64 direct convolution for: direct fp32 (sequential K order, like the MFMA kernel),
Winograd fp32 with fp32-computed transformed weights, Winograd fp32 with fp64-computed-then-rounded transformed weights."""
import numpy as np
import torch


def cook_toom(m, r, pts):
    """AT (m x a), G (a x r), BT (a x a), a = m + r - 1, from the finite points `pts` plus infinity (float64)."""
    a = m + r - 1
    assert len(pts) == a - 1
    P = np.array(pts, dtype=np.float64)

    def vander(n):
        V = np.zeros((a, n))
        for i, p in enumerate(P):
            V[i] = p ** np.arange(n)
        V[a - 1, n - 1] = 1.0
        return V
    AT = vander(m).T
    Gm = vander(r)
    for i in range(a - 1):                 # scale rows of G by 1 / prod_{j != i} (p_i - p_j)
        Gm[i] /= np.prod([P[i] - P[j] for j in range(a - 1) if j != i])
    BT = np.zeros((a, a))                  # row i: coefficients of M(x) / (x - p_i), last row: M(x) = prod (x - p_i)
    M = np.poly(P)                         # highest power first
    for i in range(a - 1):
        qi, _ = np.polydiv(M, np.array([1.0, -P[i]]))
        BT[i, :a - 1] = qi[::-1]
    BT[a - 1, :] = M[::-1]
    return AT, Gm, BT


def check_1d(AT, G, BT, m, r):
    g = np.random.default_rng(0).standard_normal(r)
    d = np.random.default_rng(1).standard_normal(m + r - 1)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([np.dot(d[i:i + r], g) for i in range(m)])
    return np.abs(y - ref).max()


def conv_direct(x, w, dtype):
    """x (C,H,W) zero padded outside, w (O,C,5,5); sequential accumulation over (c, dy, dx) in `dtype`."""
    C, H, W = x.shape
    O = w.shape[0]
    xp = torch.zeros((C, H + 4, W + 4), dtype=dtype)
    xp[:, 2:-2, 2:-2] = x.to(dtype)
    out = torch.zeros((O, H, W), dtype=dtype)
    wd = w.to(dtype)
    for c in range(C):
        for dy in range(5):
            for dx in range(5):
                out += wd[:, c, dy, dx, None, None] * xp[c, dy:dy + H, dx:dx + W][None]
    return out


def conv_winograd(x, w, AT, G, BT, wt_dtype):
    """F(2x2,5x5), everything rounded to fp32; the transformed weights U are computed in `wt_dtype` then rounded."""
    f32 = torch.float32
    C, H, W = x.shape
    O = w.shape[0]
    ATt, BTt = torch.tensor(AT, dtype=f32), torch.tensor(BT, dtype=f32)
    Gt = torch.tensor(G, dtype=wt_dtype)
    U = torch.einsum("ai,ocij,bj->ocab", Gt, w.to(wt_dtype), Gt).to(f32)          # (O,C,6,6)
    xp = torch.zeros((C, H + 4, W + 4), dtype=f32)
    xp[:, 2:-2, 2:-2] = x.to(f32)
    th, tw = H // 2, W // 2
    tiles = xp.unfold(1, 6, 2).unfold(2, 6, 2)                                      # (C, th, tw, 6, 6)
    V = torch.einsum("ai,cyxij,bj->cyxab", BTt, tiles, BTt)                         # fp32 input transform
    Mt = torch.zeros((O, th, tw, 6, 6), dtype=f32)
    for c in range(C):                                                              # fp32 accumulation over channels
        Mt += U[:, c, None, None] * V[c][None]
    Y = torch.einsum("ia,oyxab,jb->oyxij", ATt, Mt, ATt)                            # (O, th, tw, 2, 2)
    return Y.permute(0, 1, 3, 2, 4).reshape(O, H, W)


def rel(a, ref):
    return float((a.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


def main():
    AT, G, BT = cook_toom(2, 5, [0, 1, -1, 2, -2])
    print("1-D identity check (fp64):", check_1d(AT, G, BT, 2, 5))
    print("max |entry| of A^T, G, B^T:", np.abs(AT).max(), np.abs(G).max(), np.abs(BT).max())
    C = O = 128
    H = W = 16
    g = torch.Generator().manual_seed(0)
    rows = []
    for name, wgen in (("He-normal weights, post-ReLU activations (conv3/6/10 operands)",
                        lambda: torch.randn((O, C, 5, 5), generator=g, dtype=torch.float64) * np.sqrt(2.0 / (25 * O))),
                       ("uniform weights (KAT-0 flavour)",
                        lambda: (torch.rand((O, C, 5, 5), generator=g, dtype=torch.float64) - 0.5) * 2 * np.sqrt(3) * np.sqrt(2.0 / (25 * O)))):
        w = wgen().float().double()
        x = torch.relu(torch.randn((C, H, W), generator=g, dtype=torch.float64)).float().double()
        ref = conv_direct(x, w, torch.float64)
        d32 = conv_direct(x, w, torch.float32)
        w32 = conv_winograd(x, w, AT, G, BT, torch.float32)
        w64 = conv_winograd(x, w, AT, G, BT, torch.float64)
        rows.append((name, rel(d32, ref), rel(w32, ref), rel(w64, ref)))
    print("| operands | direct fp32 (sequential K) | Winograd F(2x2,5x5) fp32 | same, weights transformed in fp64 |")
    print("|---|---|---|---|")
    for n, a, b, c in rows:
        print(f"| {n} | {a:.2e} | {b:.2e} ({b / a:.1f}x) | {c:.2e} ({c / a:.1f}x) |")


if __name__ == "__main__":
    main()
