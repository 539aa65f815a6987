"""Host model of the index arithmetic of ``ops/csrc/small_linear.hip`` (backward, 32x32 tiles).

Every global load and store of every lane is replayed in numpy with a bounds check, every output element must be
written exactly once, and the result is compared with the plain matrix products.  A new tiling is run through this model
before its kernel is launched on a GPU: an out-of-range address here is an assertion, there it is a device fault.
Run as a script for the full shape list; ``tests/test_kernel_index_models_cpu.py`` runs a few small shapes.
"""
import numpy as np
WAVES = 16

def d_row(i, l):
    return (i & 3) + 8 * (i >> 2) + 4 * (l >> 5)

class Buf:
    def __init__(self, arr, name):
        self.a = arr.reshape(-1); self.name = name
    def chk(self, idx):
        idx = np.asarray(idx)
        assert idx.size == 0 or ((idx >= 0).all() and (idx < self.a.size).all()), (self.name, idx.min(), idx.max(), self.a.size)
        return idx
    def ld(self, idx): return self.a[self.chk(idx)]
    def st(self, idx, v):
        idx = self.chk(idx)
        assert np.isnan(self.a[idx]).all(), "element written twice"
        self.a[idx] = v

def mfma(a, b, acc):
    l = np.arange(64)
    for k in range(2):
        A = np.zeros(32); B = np.zeros(32)
        sel = (l >> 5) == k
        A[l[sel] & 31] = a[sel]; B[l[sel] & 31] = b[sel]
        for i in range(16):
            acc[i] += A[d_row(i, l)] * B[l & 31]
    return acc

def run(R, inn, out, want_dx=True, want_db=True, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((R, inn)); W = rng.standard_normal((out, inn)); dY = rng.standard_normal((R, out))
    bX, bW, bdY = Buf(X.copy(), "X"), Buf(W.copy(), "W"), Buf(dY.copy(), "dY")
    dX = Buf(np.full((R, inn), np.nan), "dX"); dW = Buf(np.full((out, inn), np.nan), "dW"); db = Buf(np.full(out, np.nan), "db")
    nti, nto = (inn + 31) // 32, (out + 31) // 32
    rpw = ((R + WAVES - 1) // WAVES + 1) // 2 * 2
    kq = ((out + 3) // 4 + 7) // 8 * 8
    assert WAVES * rpw >= R and 4 * kq >= out and rpw % 2 == 0
    n_dx = ((R + 127) // 128) * nti if want_dx else 0
    l = np.arange(64); n = l & 31; kh = l >> 5
    T = np.arange(1024)
    vec = out % 4 == 0
    for b in range(n_dx):
        red = np.zeros((WAVES, 1024))
        i0 = (b % nti) * 32
        for wave in range(WAVES):
            sub, quarter = wave >> 2, wave & 3
            r0 = (b // nti) * 128 + sub * 32
            row, col = r0 + n, i0 + n
            row_ok, col_ok = row < R, col < inn
            arow = np.where(row_ok, row, 0) * out
            bcol = np.where(col_ok, col, 0)
            k_begin = quarter * kq; k_end = min(out, k_begin + kq)
            acc = np.zeros((16, 64))
            if r0 < R:
                for k0 in range(k_begin, max(k_end, k_begin), 64):
                    for it in range(8):
                        kb = k0 + it * 8 + kh * 4
                        vmask = row_ok & vec & (kb + 3 < k_end)
                        if vmask.any():
                            addr = (arow + kb)[vmask]
                            assert (addr % 4 == 0).all()
                            bdY.chk(addr + 3)
                        for j in range(4):
                            ok = row_ok & (kb + j < k_end)
                            a = np.zeros(64); a[ok] = bdY.ld((arow + kb + j)[ok])
                            okb = col_ok & (kb + j < k_end)
                            bb = np.zeros(64); bb[okb] = bW.ld((bcol + (kb + j) * inn)[okb])
                            if k0 + it * 8 < k_end:
                                acc = mfma(a, bb, acc)
                            else:
                                assert not a.any() and not bb.any()
            for i in range(16):
                red[wave][i * 64 + l] = acc[i]
        g, tt = T >> 8, T & 255
        rb = (b // nti) * 128 + g * 32
        for j in range(4):
            e = tt + 256 * j; i = e >> 6; ll = e & 63
            v = red[g * 4, e] + red[g * 4 + 1, e] + red[g * 4 + 2, e] + red[g * 4 + 3, e]
            r = rb + d_row(i, ll); c = i0 + (ll & 31)
            ok = (r < R) & (c < inn)
            dX.st((r * inn + c)[ok], v[ok])
    for tile in range(nto * nti):
        to, ti = tile // nti, tile % nti
        o0, i0 = to * 32, ti * 32
        o_ok, i_ok = o0 + n < out, i0 + n < inn
        ap = np.where(o_ok, o0 + n, 0); bp = np.where(i_ok, i0 + n, 0)
        red = np.zeros((WAVES, 1024)); redb = np.zeros((WAVES, 32))
        for wave in range(WAVES):
            w_begin = wave * rpw; w_end = min(R, w_begin + rpw)
            acc = np.zeros((16, 64)); bsum = np.zeros(64)
            for r in range(w_begin, max(w_end, w_begin), 64):
                for j in range(32):
                    rr = r + 2 * j + kh
                    ok = rr < w_end
                    a = np.zeros(64); m = ok & o_ok; a[m] = bdY.ld((ap + rr * out)[m])
                    bb = np.zeros(64); m = ok & i_ok; bb[m] = bX.ld((bp + rr * inn)[m])
                    if r + 2 * j < w_end:
                        acc = mfma(a, bb, acc); bsum += a
                    else:
                        assert not a.any() and not bb.any()
            bsum = bsum + bsum[l ^ 32]
            for i in range(16):
                red[wave][i * 64 + l] = acc[i]
            redb[wave][n[kh == 0]] = bsum[kh == 0]
        v = red.sum(0)
        i = T >> 6; ll = T & 63
        o = o0 + d_row(i, ll); c = i0 + (ll & 31)
        ok = (o < out) & (c < inn)
        dW.st((o * inn + c)[ok], v[ok])
        if want_db and ti == 0:
            t32 = np.arange(32); ok = o0 + t32 < out
            db.st((o0 + t32)[ok], redb.sum(0)[ok])
    np.testing.assert_allclose(dW.a.reshape(out, inn), dY.T @ X, rtol=1e-9, atol=1e-9)
    if want_dx:
        np.testing.assert_allclose(dX.a.reshape(R, inn), dY @ W, rtol=1e-9, atol=1e-9)
    if want_db:
        np.testing.assert_allclose(db.a, dY.sum(0), rtol=1e-9, atol=1e-9)
    print("ok", R, inn, out, "rpw", rpw, "kq", kq, "blocks", n_dx + nto * nti)


if __name__ == "__main__":
    run(160, 64, 96)
    run(37, 50, 33)
    run(32, 64, 1)
    run(300, 10, 64, want_dx=False)
    run(130, 36, 40, want_db=False)
    run(1600, 32, 32)
    run(70, 40, 300)
    run(1, 1, 1)
    run(257, 33, 65)
    run(2100, 8, 16)

