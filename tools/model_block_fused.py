"""Index-level model of conv_block_c64_fused_kernel (csrc/conv_block.hip): the flattened-strip geometry, the LDS ring sizes,
the lock-step schedule and the LDS-DMA piece schedule, executed on numpy arrays with poisoned (NaN) rings and explicit
"valid since step" / "overwritten at step" bookkeeping.  Checks the fused BasicBlock against a direct float64 evaluation.
    python tools/model_block_fused.py
Build-container tool (no GPU): it validates the formulas the kernel's host side uses (L, NX, NT, prefill, piece order)."""
import math
import sys

import numpy as np

GX = GT = 8


def geometry(W):
    Wc = W if W <= 128 else 128
    assert W % 8 == 0 and W % Wc == 0
    P = Wc + 8
    L = -(-(2 * P + 129) // 64)
    NT = -(-(64 * (L - 1) + 65) // 16) * 16
    NX = -(-(64 * L + 200) // 16) * 16
    return Wc, P, L, NX, NT


def conv3x3(x, w):
    """x [H,W,C] (zero padded), w [Co,3,3,Ci] -> [H,W,Co], float64"""
    H, W, C = x.shape
    xp = np.zeros((H + 2, W + 2, C))
    xp[1:-1, 1:-1] = x
    out = np.zeros((H, W, w.shape[0]))
    for ky in range(3):
        for kx in range(3):
            out += xp[ky:ky + H, kx:kx + W] @ w[:, ky, kx, :].T
    return out


def run_strip(x, w1, w2, y0, y1, c0, Wc, out, C=4):
    """one strip: rows [y0,y1), columns [c0, c0+Wc) of image x [H,W,C]; writes out[y0:y1, c0:c0+Wc]."""
    H, W, _ = x.shape
    P = Wc + 8
    L = -(-(2 * P + 129) // 64)              # conv2 lag in steps: the epilogue of a conv1 tile runs one step after its MFMAs
    NT = -(-(64 * (L - 1) + 65) // 16) * 16
    NX = -(-(64 * L + 200) // 16) * 16
    R = y1 - y0
    n1 = -(-((R + 2) * P) // 32)
    n2 = -(-(R * P) // 32)
    S = max(-(-n1 // 2), -(-n2 // 2) + L) + 1
    xr = np.full((NX, C), np.nan)          # rings: NaN = never written
    tr = np.full((NT, C), np.nan)
    x_tag = np.full(NX, -10**9)            # absolute index currently held by each ring slot
    t_tag = np.full(NT, -10**9)

    def dma_piece(i):
        """piece i = X indices 8i..8i+7 ; X = GX + xr_row*P + p ; p <-> col c0-2+p ; row = y0-2+xr_row"""
        for l in range(8):
            X = 8 * i + l
            q = X - GX
            val = np.zeros(C)
            if q >= 0:
                row = y0 - 2 + q // P
                col = c0 - 2 + q % P
                if 0 <= row < H and 0 <= col < W:
                    val = x[row, col]
            xr[X % NX] = val
            x_tag[X % NX] = X

    def read_x(X):
        assert X >= 0
        if x_tag[X % NX] != X:             # stale / in-flight / never written: poison (only garbage positions may see this)
            return np.full(C, np.nan)
        return xr[X % NX]

    def read_t(T):
        assert T >= 0
        if t_tag[T % NT] != T:
            return np.full(C, np.nan)
        return tr[T % NT]

    npre = P // 4 + 17                     # pieces 1..npre before step 0 (cover steps 0 and 1)
    for i in range(1, npre + 1):
        dma_piece(i)
    t_prev = []
    pend = []                              # pieces issued at step s land "during" step s+1 at the latest: model the two
    for s in range(S):                     # extremes -- land at once (WAR check) and land late (RAW check via tags)
        # all reads of this step happen against the ring state at the start of the step + early landing of this step's DMA
        issue = [npre + 8 * s + k for k in range(1, 9)]
        for i in issue:                    # earliest landing: the old content of these slots may be gone from now on
            for l in range(8):
                x_tag[(8 * i + l) % NX] = -1
        for wv in range(2):                # t slots written during this step (the PREVIOUS step's tiles) may change under a reader
            if s >= 1 and 2 * (s - 1) + wv < n1:
                for l in range(32):
                    t_tag[(GT + 32 * (2 * (s - 1) + wv) + l) % NT] = -1
        for T, v in t_prev:                # ... and are complete at the end of this step
            pass
        # --- A waves: conv1 tiles 2s, 2s+1
        t_new = []
        for wv in range(2):
            k = 2 * s + wv
            if k >= n1:
                continue
            for l in range(32):
                tq = 32 * k + l
                acc = np.zeros(w1.shape[0])
                for ky in range(3):
                    for kx in range(3):
                        acc += w1[:, ky, kx, :] @ read_x(GX + tq + ky * P + kx - 1)
                rr, p = tq // P, tq % P
                row, col = y0 - 1 + rr, c0 - 2 + p
                v = np.maximum(acc, 0.0)
                if not (0 <= row < H and 0 <= col < W):
                    v = np.zeros_like(v)
                t_new.append((GT + tq, v))
        # --- B waves: conv2 tiles 2(s-L), +1
        for wv in range(2):
            j = 2 * (s - L) + wv
            if s < L or j >= n2:
                continue
            for l in range(32):
                oq = 32 * j + l
                o, p = oq // P, oq % P
                col = c0 - 2 + p
                ok = (0 <= p - 2 < Wc) and o < R
                if not ok:
                    # garbage position: the kernel computes it anyway -- its reads must at least be in-range (they are modulo
                    # the ring) -- nothing to check
                    continue
                acc = np.zeros(w2.shape[0])
                for ky in range(3):
                    for kx in range(3):
                        acc += w2[:, ky, kx, :] @ read_t(GT + oq + ky * P + kx - 1)
                res = read_x(GX + oq + 2 * P)
                out[y0 + o, col] = np.maximum(acc + res, 0.0)
        # end of step: t tiles become visible, this step's DMA lands (everything a later step reads was issued >= 2 steps
        # before it is read; landing it here = "as late as the wait at the end of the NEXT step allows" is modelled by pend)
        for T, v in t_prev:                # epilogue of the previous step's conv1 tiles: visible from the next step on
            tr[T % NT] = v
            t_tag[T % NT] = T
        t_prev = t_new
        for i in pend:
            dma_piece(i)
        pend = issue
    return L, NX, NT, S


def check(H, W, strips, C=4, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((H, W, C))
    w1 = rng.standard_normal((C, 3, 3, C)) * 0.3
    w2 = rng.standard_normal((C, 3, 3, C)) * 0.3
    t = np.maximum(conv3x3(x, w1), 0.0)
    ref = np.maximum(conv3x3(t, w2) + x, 0.0)
    Wc = W if W <= 128 else 128
    out = np.full((H, W, C), np.nan)
    info = None
    for c0 in range(0, W, Wc):
        for i in range(strips):
            y0, y1 = i * H // strips, (i + 1) * H // strips
            info = run_strip(x, w1, w2, y0, y1, c0, Wc, out, C)
    err = np.abs(out - ref).max()
    print("H=%d W=%d strips=%d  L=%d NX=%d NT=%d S=%d  max err %.2e" % ((H, W, strips) + info + (err,)))
    assert err < 1e-9


def magic_check():
    """rr = (q * MAG) >> 22 with MAG = ceil(2^22 / P) must equal q // P for every q the kernel divides"""
    for P in range(16, 144, 8):
        mag = -(-(1 << 22) // P)
        for q in range(0, 40000):
            assert (q * mag) >> 22 == q // P, (P, q)
    print("magic division ok for P in 16..136, q < 40000")


if __name__ == "__main__":
    magic_check()
    check(16, 16, 1)
    check(16, 16, 3)
    check(32, 32, 4)
    check(24, 64, 2)
    check(20, 128, 1)
    check(40, 128, 2)
    check(12, 256, 1)
    print("ok")
