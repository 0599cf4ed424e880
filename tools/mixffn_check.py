"""Operator-level check of mixffn16.hip (evfly_op_mixffn_block_bf16) against a torch restatement of the five launches it replaces,
with the same bf16 rounding points; optional probes (identity-like weights) to localise an error.  python tools/mixffn_check.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from evfly_amd import _lib

def bf(t): return t.to(torch.bfloat16).float()

def ref(x, w1, b1, dw, db, w2, b2, g, bt, h, w):
    n, N, C = x.shape
    h1 = bf(x @ bf(w1).t() + b1)                                     # (n, N, E)
    E = h1.shape[-1]
    t = h1.transpose(1, 2).reshape(n, E, h, w)
    t = F.conv2d(t, bf(dw), db, padding=1, groups=E // 8)
    h2 = bf(F.gelu(t).flatten(2).transpose(1, 2))
    x2 = bf(x + (h2 @ bf(w2).t() + b2))
    return bf(F.layer_norm(x2, (C,), g, bt, 1e-5)), h1, h2, x2

S2 = os.environ.get('MF_STAGE') == '2'


def run(n=3, h=8 if S2 else 15, w=12 if S2 else 23, C=256 if S2 else 128, E=None, seed=0, probe=None, check=True):
    E = E or 8 * C
    rs = np.random.RandomState(seed)
    x = bf(torch.from_numpy(rs.standard_normal((n, h * w, C)).astype(np.float32)))
    w1 = torch.from_numpy((rs.standard_normal((E, C)) / np.sqrt(C)).astype(np.float32))
    b1 = torch.from_numpy((0.1 * rs.standard_normal(E)).astype(np.float32))
    dw = torch.from_numpy((rs.standard_normal((E, 8, 3, 3)) / np.sqrt(72)).astype(np.float32))
    db = torch.from_numpy((0.1 * rs.standard_normal(E)).astype(np.float32))
    w2 = torch.from_numpy((rs.standard_normal((C, E)) / np.sqrt(E)).astype(np.float32))
    b2 = torch.from_numpy((0.1 * rs.standard_normal(C)).astype(np.float32))
    g = torch.from_numpy((1 + 0.1 * rs.standard_normal(C)).astype(np.float32))
    bt = torch.from_numpy((0.1 * rs.standard_normal(C)).astype(np.float32))
    if probe == "nodw":       # grouped conv = identity on the centre tap
        dw.zero_(); db.zero_()
        for c in range(E): dw[c, c % 8, 1, 1] = 1.0
    if probe == "w2zero": w2.zero_()
    if probe == "w1zero": w1.zero_(); dw.zero_(); db.zero_(); [dw.__setitem__((c, c % 8, 1, 1), 1.0) for c in range(E)]
    if probe == "w1zero_b0": w1.zero_(); b1.zero_()
    if check: want, h1, h2, x2 = ref(x, w1, b1, dw, db, w2, b2, g, bt, h, w)
    L = _lib.lib()
    dev = [t.cuda().contiguous() for t in (w1, b1, dw, db, w2, b2, g, bt)]
    xb = x.to(torch.bfloat16).view(torch.int16).cuda().contiguous()
    y = torch.zeros(n, h * w, C, dtype=torch.int16, device="cuda")
    _lib.check(L.evfly_op_mixffn_block_bf16(_lib.ptr(xb), n, h, w, C, E, *[_lib.ptr(t) for t in dev], _lib.ptr(y), _lib.cur_stream()))
    torch.cuda.synchronize()
    if not check: return 0.0
    got = y.view(torch.bfloat16).float().cpu()
    dbg = os.environ.get("MF_DBG")           # "1:sl" / "2:sl": a developer build that dumps a slab of h1 / h2 (EVFLY_LIB variant)
    if dbg:
        kind, sl = (int(v) for v in dbg.split(":"))
        want = (h1 if kind == 1 else h2)[:, :, sl * 32:(sl + 1) * 32]
        got = got[:, :, :32]
    err = (got - want).abs()
    if dbg:
        for t in (10, 287, 288, 300, 330):
            print("  token", t, "got", [round(v, 3) for v in got[0, t, :8].tolist()], "want", [round(v, 3) for v in want[0, t, :8].tolist()])
    if dbg: print("  per-channel max err", [round(v, 2) for v in err.amax((0, 1)).tolist()])
    print("probe", probe, "n", n, "max err", err.max().item(), "rel", (err.max() / want.abs().max()).item(), "finite", torch.isfinite(got).all().item())
    bad = (err > 0.1).nonzero()
    if len(bad):
        print("bad elems", len(bad), "of", err.numel(), "frames", sorted(set(bad[:, 0].tolist()))[:10], "tokens", sorted(set(bad[:, 1].tolist()))[:40],
              "chans", sorted(set(bad[:, 2].tolist()))[:40])
    return err.max().item() / max(want.abs().max().item(), 1e-30)



def probe_w1_rows(n=1, h=15, w=23, C=128, E=1024):
    """developer probe (MF_DBG=1:sl build): W1[e][k] = delta(k, e % C), b1 = 0 -> h1[t][e] must be x[t][e % C]; prints, for the dumped
    slab, which x column every hidden channel actually received."""
    kind, sl = (int(v) for v in os.environ["MF_DBG"].split(":"))
    rs = np.random.RandomState(3)
    x = bf(torch.from_numpy(rs.standard_normal((n, h * w, C)).astype(np.float32)))
    mul, add = (int(v) for v in os.environ.get('MF_PERM', '1:0').split(':'))
    perm = (mul * torch.arange(E) + add) % C
    w1 = torch.zeros(E, C); w1[torch.arange(E), perm] = 1.0
    z = lambda *s: torch.zeros(*s)
    dw = z(E, 8, 3, 3)
    dev = [t.cuda().contiguous() for t in (w1, z(E), dw, z(E), z(C, E), z(C), torch.ones(C), z(C))]
    xb = x.to(torch.bfloat16).view(torch.int16).cuda().contiguous()
    y = torch.zeros(n, h * w, C, dtype=torch.int16, device="cuda")
    L = _lib.lib()
    _lib.check(L.evfly_op_mixffn_block_bf16(_lib.ptr(xb), n, h, w, C, E, *[_lib.ptr(t) for t in dev], _lib.ptr(y), _lib.cur_stream()))
    got = y.view(torch.bfloat16).float().cpu()[0, :, :32]           # (tokens, 32)
    for e in range(32):
        col = got[:, e]
        match = [(k, (x[0, :, k] - col).abs().max().item()) for k in range(C)]
        k, err = min(match, key=lambda p: p[1])
        print("hidden", sl * 32 + e, "<- x column", k, "err", round(err, 4), "" if k == perm[sl * 32 + e].item() else "  <-- WRONG, want %d" % perm[sl * 32 + e].item())


if os.environ.get("MF_GELU"):          # the cubic / cubic erf rational of mixffn16.hip: minimax fit on |v| <= 3.5 and its error (CPU only)
    from scipy.special import erf
    from scipy.optimize import least_squares
    m, n_, c = 3, 3, 3.5
    v = np.linspace(1e-3, c, 6000); t = v * v; yv = erf(v) / v
    model = lambda p, t: sum(p[i] * t ** i for i in range(m + 1)) / (1 + sum(p[m + 1 + i] * t ** (i + 1) for i in range(n_)))
    res = lambda p: (model(p, t) - yv) * v
    A = np.concatenate([np.stack([t ** i for i in range(m + 1)], 1), -np.stack([yv * t ** (i + 1) for i in range(n_)], 1)], 1)
    p = np.linalg.lstsq(A, yv, rcond=None)[0]
    best = None
    for it in range(40):                 # growing exponent: least squares -> minimax
        p = least_squares(lambda q: np.sign(res(q)) * np.abs(res(q)) ** (1 + it * 0.3), p, xtol=1e-15, ftol=1e-15, max_nfev=4000).x
        e = np.abs(res(p)).max()
        if best is None or e < best[0]: best = (e, p.copy())
    e, p = best
    print("max |erf error| on [0, %.1f]: %.3e" % (c, e))
    print("P (v^0, v^2, v^4, v^6):", [float(np.float32(x)) for x in p[:m + 1]])
    print("Q:", [1.0] + [float(np.float32(x)) for x in p[m + 1:]])
    a = np.linspace(-12, 12, 2400001).astype(np.float32)
    v32 = np.clip(a * np.float32(0.70710678118), -c, c).astype(np.float32); t32 = v32 * v32
    P = [np.float32(x) for x in p[:m + 1]]; Q = [np.float32(1)] + [np.float32(x) for x in p[m + 1:]]
    num = ((P[3] * t32 + P[2]) * t32 + P[1]) * t32 + P[0]
    den = ((Q[3] * t32 + Q[2]) * t32 + Q[1]) * t32 + Q[0]
    g = a * ((v32 * num * (np.float32(1) / den)) * np.float32(0.5) + np.float32(0.5))
    gt = 0.5 * a.astype(np.float64) * (1 + erf(a.astype(np.float64) / np.sqrt(2)))
    print("max |GELU error| in fp32 arithmetic on [-12, 12]: %.3e" % np.abs(g - gt).max())
    sys.exit(0)
if os.environ.get("MF_TS"):            # phase timeline of block 0 (an -DEVFLY_MF_TS build selected with EVFLY_LIB)
    import ctypes
    E = int(os.environ.get("MF_E", "0")) or (2048 if S2 else 1024)
    run(int(os.environ.get("MF_TS_FRAMES", "2560")), E=E, seed=2, check=False)
    buf = np.zeros(12 * 8, dtype=np.uint64)
    L = _lib.lib()
    L.evfly_debug_mixffn_ts.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.evfly_debug_mixffn_ts(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(12, 8).astype(np.float64) / (E // 32)         # s_memtime ticks (100 MHz) per slab
    names = ["mlp1", "wait+bar1", "gconv+gelu", "wait+bar2", "mlp2", "prologue", "epilogue", "-"]
    print("ticks of the 100 MHz counter per slab (x ~24 = shader cycles at 2.4 GHz); prologue / epilogue per frame")
    for w in range(12):
        row = t[w].copy(); row[5] *= E // 32; row[6] *= E // 32
        print("wave %2d " % w + "  ".join("%s %.1f" % (names[i], row[i]) for i in range(7)))
    sys.exit(0)
if os.environ.get("MF_TIME"):          # MF_TIME=<frames>: EVFLY_MIXFFN_TIME launches of the full-size problem (prints ms per launch)
    os.environ.setdefault("EVFLY_MIXFFN_TIME", "20")
    run(int(os.environ["MF_TIME"]), E=int(os.environ.get("MF_E", "0")) or None, seed=2, check=False)
    sys.exit(0)
if os.environ.get("MF_PROBE_W1"):
    probe_w1_rows()


def probe_h1_values():
    kind, sl = (int(v) for v in os.environ["MF_DBG"].split(":"))
    n, h, w, C, E = 1, 15, 23, 128, 1024
    for zero_b1 in (True, False):
        rs = np.random.RandomState(0)
        x = bf(torch.from_numpy(rs.standard_normal((n, h * w, C)).astype(np.float32)))
        w1 = torch.from_numpy((rs.standard_normal((E, C)) / np.sqrt(C)).astype(np.float32))
        b1 = torch.from_numpy((0.1 * rs.standard_normal(E)).astype(np.float32))
        if zero_b1: b1.zero_()
        z = lambda *s: torch.zeros(*s)
        dev = [t.cuda().contiguous() for t in (w1, b1, z(E, 8, 3, 3), z(E), z(C, E), z(C), torch.ones(C), z(C))]
        xb = x.to(torch.bfloat16).view(torch.int16).cuda().contiguous()
        y = torch.zeros(n, h * w, C, dtype=torch.int16, device="cuda")
        L = _lib.lib()
        _lib.check(L.evfly_op_mixffn_block_bf16(_lib.ptr(xb), n, h, w, C, E, *[_lib.ptr(t) for t in dev], _lib.ptr(y), _lib.cur_stream()))
        got = y.view(torch.bfloat16).float().cpu()[0, :, :32]
        want = bf(x[0] @ bf(w1).t() + b1)[:, sl * 32:(sl + 1) * 32]
        nob = bf(x[0] @ bf(w1).t())[:, sl * 32:(sl + 1) * 32]
        err = (got - want).abs()
        print("zero_b1", zero_b1, "max err per channel", [round(v, 3) for v in err.max(0).values.tolist()])
        print(" token 5: got", [round(v, 3) for v in got[5, :12].tolist()])
        print("         want", [round(v, 3) for v in want[5, :12].tolist()])
        print("   b1", [round(v, 3) for v in b1[sl * 32:sl * 32 + 12].tolist()])
        print("   got - (x W1)", [round(v, 3) for v in (got - nob)[5, :12].tolist()])


if os.environ.get("MF_PROBE_H1"):
    probe_h1_values()


if __name__ == "__main__" and not os.environ.get("MF_PROBE_W1") and not os.environ.get("MF_PROBE_H1"):
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for pr in os.environ.get("MF_PROBES", "w2zero,w1zero,w1zero_b0,nodw,None").split(","): run(n, probe=None if pr == "None" else pr)
