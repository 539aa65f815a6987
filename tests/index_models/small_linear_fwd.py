"""Host model of the index arithmetic of ``ops/csrc/small_linear.hip`` (forward, 16x16 tiles).

Every global load and store of every lane is replayed in numpy with a bounds check, every output element must be
written exactly once, and the result is compared with the plain matrix products.  A new tiling is run through this model
before its kernel is launched on a GPU: an out-of-range address here is an assertion, there it is a device fault.
Run as a script for the full shape list; ``tests/test_kernel_index_models_cpu.py`` runs a few small shapes.
"""
import numpy as np
WAVES = 16
class Buf:
    def __init__(self, arr, name): self.a = arr.reshape(-1); self.name = name
    def chk(self, idx):
        idx = np.asarray(idx)
        assert idx.size == 0 or ((idx >= 0).all() and (idx < self.a.size).all()), (self.name, idx.min(), idx.max(), self.a.size)
        return idx
    def ld(self, idx): return self.a[self.chk(idx)]
    def st(self, idx, v):
        idx = self.chk(idx); assert np.isnan(self.a[idx]).all(), "written twice"; self.a[idx] = v
def mfma4(a, b, acc):
    l = np.arange(64)
    for k in range(4):
        A = np.zeros(16); B = np.zeros(16); sel = (l >> 4) == k
        A[l[sel] & 15] = a[sel]; B[l[sel] & 15] = b[sel]
        for i in range(4): acc[i] += A[4 * (l >> 4) + i] * B[l & 15]
    return acc
def run(R, inn, out, bias=True, relu=False, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((R, inn)); W = rng.standard_normal((out, inn)); bvec = rng.standard_normal(out)
    bX, bW, bB = Buf(X.copy(), "X"), Buf(W.copy(), "W"), Buf(bvec.copy(), "b")
    Y = Buf(np.full((R, out), np.nan), "Y")
    nto = (out + 15) // 16; ks = 4 if inn >= 256 else 1
    kq = ((inn + ks - 1) // ks + 15) // 16 * 16
    assert ks * kq >= inn
    rpb = 64 if ks == 4 else 256
    blocks = ((R + rpb - 1) // rpb) * nto
    l = np.arange(64); n = l & 15; kg = l >> 4; T = np.arange(1024); vec = inn % 4 == 0
    for blk in range(blocks):
        rb = (blk // nto) * rpb; o0 = (blk % nto) * 16
        red = np.zeros((WAVES, 256))
        for wave in range(WAVES):
            sub = wave >> 2 if ks == 4 else wave; quarter = wave & 3 if ks == 4 else 0
            r0 = rb + sub * 16
            row, ocol = r0 + n, o0 + n
            row_ok, col_ok = row < R, ocol < out
            arow = np.where(row_ok, row, 0) * inn; brow = np.where(col_ok, ocol, 0) * inn
            k_begin = quarter * kq; k_end = min(inn, k_begin + kq)
            acc = np.zeros((4, 64))
            if r0 < R:
                for k0 in range(k_begin, max(k_end, k_begin), 128):
                    for it in range(8):
                        kb = k0 + it * 16 + kg * 4
                        vm = vec & (kb + 3 < k_end)
                        if vm.any():
                            assert ((arow + kb)[vm & row_ok] % 4 == 0).all() and ((brow + kb)[vm & col_ok] % 4 == 0).all()
                            bX.chk((arow + kb + 3)[vm & row_ok]); bW.chk((brow + kb + 3)[vm & col_ok])
                        for j in range(4):
                            a = np.zeros(64); m = row_ok & (kb + j < k_end); a[m] = bX.ld((arow + kb + j)[m])
                            b = np.zeros(64); m = col_ok & (kb + j < k_end); b[m] = bW.ld((brow + kb + j)[m])
                            if k0 + it * 16 < k_end: acc = mfma4(a, b, acc)
                            else: assert not a.any() and not b.any()
            if ks == 4:
                for i in range(4): red[wave][i * 64 + l] = acc[i]
            else:
                for i in range(4):
                    r = r0 + 4 * kg + i
                    ok = col_ok & (r < R)
                    v = acc[i] + (bB.ld(np.where(col_ok, ocol, 0)) if bias else 0)
                    if relu: v = np.maximum(v, 0)
                    Y.st((r * out + ocol)[ok], v[ok])
        if ks == 4:
            g, e = T >> 8, T & 255; i = e >> 6; ll = e & 63
            v = red[g * 4, e] + red[g * 4 + 1, e] + red[g * 4 + 2, e] + red[g * 4 + 3, e]
            r = rb + g * 16 + 4 * (ll >> 4) + i; c = o0 + (ll & 15)
            ok = (r < R) & (c < out)
            if bias: v = v + bB.ld(np.where(c < out, c, 0))
            if relu: v = np.maximum(v, 0)
            Y.st((r * out + c)[ok], v[ok])
    ref = X @ W.T + (bvec if bias else 0)
    if relu: ref = np.maximum(ref, 0)
    np.testing.assert_allclose(Y.a.reshape(R, out), ref, rtol=1e-9, atol=1e-9)
    print("ok fwd", R, inn, out, "ks", ks, "kq", kq, "blocks", blocks)

if __name__ == "__main__":
    run(160, 64, 96); run(37, 50, 33, relu=True); run(32, 64, 1); run(300, 10, 64); run(130, 260, 40, bias=False)
    run(70, 512, 24, relu=True); run(1, 1, 1); run(257, 33, 65); run(600, 256, 16)

