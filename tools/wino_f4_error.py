"""Error budget of Winograd F(4x4, 3x3) against F(2x2, 3x3) in fp32 (numpy, CPU): relative max-norm error of one 3x3 valid conv layer
against a float64 direct convolution, for U-Net-like operands (post-ReLU inputs, He-scaled weights). Informs DESIGN section 8.
usage: python tools/wino_f4_error.py"""
import numpy as np
f32 = np.float32
def transforms(m):
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    else:
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
    return BT, G, AT
def wino(x, w, m):
    """x (C, H, W) fp32, w (N, C, 3, 3) fp32 -> (N, H-2, W-2), every step rounded to fp32 (U from float64 like the packers)."""
    BT, G, AT = transforms(m)
    t = m + 2
    C, H, W = x.shape; N = w.shape[0]
    OH, OW = H - 2, W - 2
    ty, tx = -(-OH // m), -(-OW // m)
    xp = np.zeros((C, ty * m + 2, tx * m + 2), f32); xp[:, :H, :W] = x
    U = np.einsum('ai,ncij,bj->abnc', G, w.astype(np.float64), G).astype(f32)           # (t, t, N, C)
    BTf = BT.astype(f32); ATf = AT.astype(f32)
    out = np.zeros((N, ty * m, tx * m), f32)
    for iy in range(ty):
        for ix in range(tx):
            d = xp[:, iy * m:iy * m + t, ix * m:ix * m + t]                                # (C, t, t)
            V = np.einsum('ai,cij,bj->abc', BTf, d, BTf, optimize=False).astype(f32)       # fp32 sums
            M = np.einsum('abnc,abc->abn', U, V).astype(f32)                               # fp32 accumulate over C
            Y = np.einsum('ia,abn,jb->nij', ATf, M, ATf).astype(f32)
            out[:, iy * m:(iy + 1) * m, ix * m:(ix + 1) * m] = Y
    return out[:, :OH, :OW]
def direct64(x, w):
    C, H, W = x.shape; N = w.shape[0]
    out = np.zeros((N, H - 2, W - 2))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum('nc,chw->nhw', w[:, :, ky, kx].astype(np.float64), x[:, ky:ky + H - 2, kx:kx + W - 2].astype(np.float64))
    return out
rs = np.random.RandomState(0)
print("layer (C -> N, map)      F(2x2) rel err   F(4x4) rel err   ratio")
for (C, N, H, W) in ((128, 128, 30, 42), (256, 256, 27, 37), (512, 512, 10, 15), (512, 256, 16, 26)):
    x = np.maximum(rs.standard_normal((C, H, W)), 0).astype(f32)
    w = (rs.standard_normal((N, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(f32)
    ref = direct64(x, w)
    e2 = np.abs(wino(x, w, 2) - ref).max() / np.abs(ref).max()
    e4 = np.abs(wino(x, w, 4) - ref).max() / np.abs(ref).max()
    print(f"{C:4d} -> {N:4d}, {H}x{W}        {e2:.2e}         {e4:.2e}        {e4 / e2:.1f}x")
