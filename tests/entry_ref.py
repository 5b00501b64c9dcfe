"""Test-side emulation (plain torch, CPU) of the *entry-space* algorithm the HIP kernels implement.

The reference evaluates the shared MLP on every (window, slot) position, although padded slots
repeat the window's first hit (query_depth_point_cuda_kernel.cu:55-59).  The kernels evaluate each
distinct (window, point) pair -- an "entry" -- once and carry a weight w = multiplicity, which enters
only the BatchNorm statistics and the BN-backward mean terms.  This file states that algorithm in
torch so that (1) its equivalence with the dense oracle (oracle/det_ref.py) is proven on CPU,
forward and backward, and (2) the GPU tests have stage-by-stage expected tensors.

Layout (shared with frustum_convnet_amd/csrc): per sample b a region of `cap = L*K` entry rows;
window l owns rows [woff[b,l], woff[b,l+1]); nent[b] = woff[b,L].
"""
import numpy as np
import torch

EPS = 1e-5


def compact(idx, cnt, pc, ref, K):
    """idx (B,L,K) int64, cnt (B,L) int32, pc (B,3,N), ref (B,3,L) -> dict of entry tensors."""
    B, L, _ = idx.shape
    cap = L * K
    ne = torch.clamp(cnt.long(), min=1)
    woff = torch.zeros(B, L + 1, dtype=torch.int32)
    woff[:, 1:] = torch.cumsum(ne, 1).int()
    ent = torch.zeros(B, cap, 4)
    ewin = torch.full((B, cap), -1, dtype=torch.int32)
    epnt = torch.full((B, cap), -1, dtype=torch.int64)
    for b in range(B):
        for l in range(L):
            n = int(ne[b, l])
            o = int(woff[b, l])
            p = idx[b, l, :n]
            ent[b, o:o + n, :3] = (pc[b][:, p] - ref[b, :, l:l + 1]).t()
            ent[b, o:o + n, 3] = 1.0
            ent[b, o, 3] = float(K - n + 1)
            ewin[b, o:o + n] = l
            epnt[b, o:o + n] = p
    return dict(cnt=cnt.clone(), ent=ent, ewin=ewin, epnt=epnt, woff=woff, nent=woff[:, L].clone(), cap=cap, B=B, L=L, K=K)


def _flat(c, t):
    """Gather the live rows of a (B,cap,...) tensor into one (E,...) tensor (sample-major order)."""
    return torch.cat([t[b, :int(c["nent"][b])] for b in range(c["B"])], 0)


def _bn_scale_shift(sum_, sumsq, M, gamma, beta):
    mean = sum_ / M
    var = sumsq / M - mean * mean
    rstd = 1.0 / torch.sqrt(var + EPS)
    s = gamma.double() * rstd
    t = beta.double() - mean * s
    return mean, var, rstd, s.float(), t.float()


def forward(c, W1, g1, b1, W2, g2, b2, W3, g3, b3, valid_mask=True):
    """Training-mode forward in entry space.  W* are (Cout,Cin) fp32.  Returns dict with every stage."""
    B, L, K = c["B"], c["L"], c["K"]
    M = float(B * L * K)
    e4 = _flat(c, c["ent"])
    u, w = e4[:, :3], e4[:, 3]
    wd = w.double()
    # layer 1 statistics from the weighted input moments (conv1 is linear in u)
    mu_u = (wd[:, None] * u.double()).sum(0) / M
    m2 = (wd[:, None, None] * u.double()[:, :, None] * u.double()[:, None, :]).sum(0) / M
    cov = m2 - mu_u[:, None] * mu_u[None, :]
    W1d = W1.double()
    mean1 = W1d @ mu_u
    var1 = ((W1d @ cov) * W1d).sum(1)
    rstd1 = 1.0 / torch.sqrt(var1 + EPS)
    s1 = (g1.double() * rstd1).float()
    t1 = (b1.double() - mean1 * g1.double() * rstd1).float()
    y1 = u @ W1.t()
    a1 = torch.relu(y1 * s1 + t1)
    y2 = a1 @ W2.t()
    sum2 = (wd[:, None] * y2.double()).sum(0)
    sq2 = (wd[:, None] * y2.double() ** 2).sum(0)
    mean2, var2, rstd2, s2, t2 = _bn_scale_shift(sum2, sq2, M, g2, b2)
    a2 = torch.relu(y2 * s2 + t2)
    y3 = a2 @ W3.t()
    sum3 = (wd[:, None] * y3.double()).sum(0)
    sq3 = (wd[:, None] * y3.double() ** 2).sum(0)
    mean3, var3, rstd3, s3, t3 = _bn_scale_shift(sum3, sq3, M, g3, b3)
    a3 = torch.relu(y3 * s3 + t3)
    # pool: max over each window's entries; masked (cnt == 0) windows pool to 0
    C3 = W3.shape[0]
    feat = torch.zeros(B, C3, L)
    amax = torch.full((B, L, C3), -1, dtype=torch.int32)   # entry row (within sample) of the max, -1 = no grad
    base = 0
    for b in range(B):
        n = int(c["nent"][b])
        a3b = a3[base:base + n]
        for l in range(L):
            o0, o1 = int(c["woff"][b, l]), int(c["woff"][b, l + 1])
            if c["cnt"][b, l] > 0:
                v, i = a3b[o0:o1].max(0)
                feat[b, :, l] = v
                amax[b, l] = torch.where(v > 0, (i + o0).int(), torch.full_like(i, -1).int())
        base += n
    return dict(u=u, w=w, y1=y1, a1=a1, y2=y2, a2=a2, y3=y3, a3=a3, feat=feat, amax=amax,
                mean=(mean1, mean2, mean3), var=(var1, var2, var3), rstd=(rstd1, rstd2, rstd3),
                s=(s1, s2, s3), t=(t1, t2, t3), M=M, mom=(mu_u, cov))


def backward(c, f, dfeat, W1, g1, W2, g2, W3, g3):
    """Entry-space backward.  dfeat (B,C3,L).  Returns grads of W1..W3, gamma/beta 1..3."""
    B, L = c["B"], c["L"]
    M = f["M"]
    w = f["w"]
    E = w.shape[0]
    C3 = W3.shape[0]
    # sparse gradient w.r.t. a3: only the argmax entry of each (window, channel)
    G3 = torch.zeros(E, C3)
    base = 0
    for b in range(B):
        n = int(c["nent"][b])
        am = f["amax"][b].long()                      # (L,C3)
        ok = am >= 0
        ll, cc = ok.nonzero(as_tuple=True)
        G3[base + am[ll, cc], cc] = dfeat[b, cc, ll]
        base += n

    def bn_back(G, y, mean, rstd, gamma, t_s):
        s, t = t_s
        z = y * s + t
        dz = G * (z > 0).float()
        xh = ((y.double() - mean) * rstd)
        dbeta = dz.double().sum(0)
        dgamma = (dz.double() * xh).sum(0)
        k = (gamma.double() * rstd)
        dy = k * (dz.double() - w.double()[:, None] * (dbeta / M) - w.double()[:, None] * xh * (dgamma / M))
        return dy.float(), dgamma.float(), dbeta.float()

    dy3, dg3, db3 = bn_back(G3, f["y3"], f["mean"][2], f["rstd"][2], g3, (f["s"][2], f["t"][2]))
    dW3 = dy3.t() @ f["a2"]
    G2 = dy3 @ W3
    dy2, dg2, db2 = bn_back(G2, f["y2"], f["mean"][1], f["rstd"][1], g2, (f["s"][1], f["t"][1]))
    dW2 = dy2.t() @ f["a1"]
    G1 = dy2 @ W2
    dy1, dg1, db1 = bn_back(G1, f["y1"], f["mean"][0], f["rstd"][0], g1, (f["s"][0], f["t"][0]))
    dW1 = dy1.t() @ f["u"]
    return dict(dW1=dW1, dg1=dg1, db1=db1, dW2=dW2, dg2=dg2, db2=db2, dW3=dW3, dg3=dg3, db3=db3,
                dy3=dy3, dy2=dy2, G2=G2, G1=G1)
