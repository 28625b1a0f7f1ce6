#!/usr/bin/env python3
"""Thread-level numpy model of the two-launch 256^3 FFT (k_fft3d_a / k_fft3d_b in indigo_amd/csrc/ig_fft2.hip).

Not product code: a CPU rehearsal of the kernels' index maps -- every load, every register DFT, every LDS exchange
with its exact addresses -- so that the arithmetic (digit orders, twiddles, exchanges) and the LDS bank behaviour are
checked before the HIP version ever runs (there is no GPU in the build container).

    python tools/fft2pass_model.py        # prints the error against numpy.fft.fftn and the conflict degree per exchange
"""
import numpy as np

N = 256
W = np.exp(-2j * np.pi * np.arange(N) / N)          # w256^k


def dft(v, axis):
    return np.fft.fft(v, axis=axis)


def conflicts(addrs):
    """worst number of lanes of one wave-instruction that fall into the same 8-byte slot modulo 64 (1 = conflict free)"""
    addrs = np.asarray(addrs).reshape(-1, 64) % 64
    return max(np.bincount(row, minlength=64).max() for row in addrs)


def pass_a(vol, z, n2, report):
    """x: full 256-point transform; y: first stage (radix 64 over lines y = 4*n1 + n2).  Returns lines[k1][kx] to be stored at
    y' = k1 + 64*n2."""
    tid = np.arange(512)
    xl, a = tid & 63, tid >> 6
    # role 0 registers v[b][j]: line n1 = a + 8b, x = xl + 64 j
    v = np.empty((512, 8, 4), complex)
    for b in range(8):
        for j in range(4):
            v[:, b, j] = vol[xl + 64 * j, 4 * (a + 8 * b) + n2, z]
    u = dft(v, 1)                                                   # y1: radix 8 over b -> kb
    kb = np.arange(8)
    u *= W[(4 * a[:, None] * kb[None, :]) & 255][:, :, None]        # w64^(a kb)
    # E1 (rounds over j): addr = (a*8 + kb)*64 + xl ; reader (xl, kb'=wave) takes a' = 0..7
    r = np.empty((512, 8, 4), complex)
    for j in range(4):
        lds = np.full(4096, np.nan, complex)
        for k in range(8):
            ad = (a * 8 + k) * 64 + xl
            report('A.E1 write', ad)
            lds[ad] = u[:, k, j]
        for ap in range(8):
            ad = (ap * 8 + a) * 64 + xl            # reader's own wave index plays kb
            report('A.E1 read', ad)
            r[:, ap, j] = lds[ad]
    kbr = a                                                         # role 1: wave index = kb
    d = dft(r, 1)                                                   # y2: radix 8 over a -> ka ; line k1 = kb + 8 ka
    ka = np.arange(8)
    k1 = kbr[:, None] + 8 * ka[None, :]
    d *= W[(n2 * k1) & 255][:, :, None]                             # w256^(n2 k1)
    p = dft(d, 2)                                                   # x1: radix 4 over j -> kj
    kj = np.arange(4)
    p *= W[(xl[:, None] * kj[None, :]) & 255][:, None, :]           # w256^(xl kj)
    # E2 (rounds over kj): addr = line*72 + xl ; reader role 2: c = tid&7, l = (tid>>3)&7, w = tid>>6, line = l + 8w, reads xl = c + 8d
    c, l, w = tid & 7, (tid >> 3) & 7, tid >> 6
    line2 = l + 8 * w
    q = np.empty((512, 8, 4), complex)                              # [d][kj]
    for jj in range(4):
        lds = np.full(64 * 72, np.nan, complex)
        for k in range(8):
            ad = (kbr + 8 * k) * 72 + xl
            report('A.E2 write', ad)
            lds[ad] = p[:, k, jj]
        for dd in range(8):
            ad = line2 * 72 + c + 8 * dd
            report('A.E2 read', ad)
            q[:, dd, jj] = lds[ad]
    s = dft(q, 1)                                                   # x2: radix 8 over d -> kd
    kd = np.arange(8)
    s *= W[(4 * c[:, None] * kd[None, :]) & 255][:, :, None]        # w64^(c kd)
    # E3 (wave local, rounds over kj): addr = line*64 + 8*((c + l)&7) + ((kd + l)&7); reader role 3: kd' = tid&7, same line
    t3 = np.empty((512, 8, 4), complex)                             # [c][kj]
    kd3 = tid & 7
    for jj in range(4):
        lds = np.full(64 * 64, np.nan, complex)
        for k in range(8):
            ad = line2 * 64 + 8 * ((c + l) & 7) + ((k + l) & 7)
            report('A.E3 write', ad)
            lds[ad] = s[:, k, jj]
        for cc in range(8):
            ad = line2 * 64 + 8 * ((cc + l) & 7) + ((kd3 + l) & 7)
            report('A.E3 read', ad)
            t3[:, cc, jj] = lds[ad]
    o = dft(t3, 1)                                                  # x3: radix 8 over c -> kc ; kx = kj + 4 kd + 32 kc
    out = np.empty((64, 256), complex)
    for kc in range(8):
        for jj in range(4):
            out[line2, jj + 4 * kd3 + 32 * kc] = o[:, kc, jj]
    return out


def pass_b(mid, xs, k1, report):
    """y: second stage (radix 4 over n2, two of the four rows per thread, the other two in the neighbouring lane);
    z: full 256-point transform (16 x 16).  mid[x][y'][z] holds pass A's output."""
    tid = np.arange(512)
    h, w, t = tid & 1, (tid >> 1) & 15, tid >> 5
    v = np.empty((512, 16, 2), complex)                             # [k][m]: z = t + 16 k, row n2 = h + 2 m
    for k in range(16):
        for m in range(2):
            v[:, k, m] = mid[xs + w, k1 + 64 * (h + 2 * m), t + 16 * k]
    a = np.stack([v[:, :, 0] + v[:, :, 1], v[:, :, 0] - v[:, :, 1]], axis=2)       # over m -> q
    a[h == 1, :, 1] *= -1j                                          # w4^(h q)
    partner = a[tid ^ 1]                                            # quad_perm [1,0,3,2]
    y = np.where((h == 0)[:, None, None], a + partner, partner - a)  # k2 = q + 2 h
    u = dft(y, 1)                                                   # z1: radix 16 over k -> kk
    kk = np.arange(16)
    u *= W[(t[:, None] * kk[None, :]) & 255][:, :, None]            # w256^(t kk)
    # exchange (rounds over q): addr = t*544 + kk*32 + h*16 + w ; reader role 1 (w, h, kk' = tid>>5) takes t = 0..15
    kk1 = tid >> 5
    out = np.empty((16, 4, 256), complex)                           # [w][k2][kz]
    for q in range(2):
        lds = np.full(16 * 544, np.nan, complex)
        for k in range(16):
            ad = t * 544 + k * 32 + h * 16 + w
            report('B.E write', ad)
            lds[ad] = u[:, k, q]
        r = np.empty((512, 16), complex)
        for tt in range(16):
            ad = tt * 544 + kk1 * 32 + h * 16 + w
            report('B.E read', ad)
            r[:, tt] = lds[ad]
        o = dft(r, 1)                                               # z2: radix 16 over t -> rr ; kz = kk + 16 rr
        for rr in range(16):
            out[w, q + 2 * h, kk1 + 16 * rr] = o[:, rr]
    return out


def main():
    rng = np.random.default_rng(0)
    # a thin volume keeps the model quick: all 256 x and y, a few z for pass A; pass B is checked on a full z column set
    vol = rng.standard_normal((256, 256, 256)) + 1j * rng.standard_normal((256, 256, 256))
    worst = {}

    def report(name, ad):
        if len(ad) % 64 == 0:
            worst[name] = max(worst.get(name, 1), conflicts(ad))
    # pass A on two z planes, all n2
    ref_xy1 = np.fft.fft(vol[:, :, :2], axis=0)
    for z in range(2):
        for n2 in range(4):
            out = pass_a(vol, z, n2, report)
            # expected: full x transform, y stage 1 with its twiddle: U[k1] = w256^(n2 k1) sum_n1 X[4 n1 + n2] w64^(n1 k1)
            lines = ref_xy1[:, n2::4, z]                             # (kx, n1)
            exp = np.fft.fft(lines, axis=1) * W[(n2 * np.arange(64)) & 255][None, :]
            err = np.abs(out - exp.T).max() / np.abs(exp).max()
            assert err < 1e-12, ("pass A", z, n2, err)
    # pass B: build the true intermediate for 16 x values and check the final result against fftn
    full = np.fft.fftn(vol)
    X = np.fft.fft(vol[:32], axis=0) if False else None
    fx = np.fft.fft(vol, axis=0)[:16]                               # kx = 0..15 as the tile's "x" values
    mid = np.empty((16, 256, 256), complex)
    for n2 in range(4):
        lines = fx[:, n2::4, :]                                      # (x, n1, z)
        mid[:, np.arange(64) + 64 * n2, :] = np.fft.fft(lines, axis=1) * W[(n2 * np.arange(64)) & 255][None, :, None]
    for k1 in (0, 5, 63):
        out = pass_b(mid, 0, k1, report)
        exp = full[:16, k1::64, :]                                   # ky = k1 + 64 k2
        err = np.abs(out - exp).max() / np.abs(exp).max()
        assert err < 1e-12, ("pass B", k1, err)
    print("two-launch 256^3 transform: index maps and twiddles agree with numpy.fft.fftn")
    for k in sorted(worst):
        print("  %-12s worst lanes per LDS slot (mod 64): %d" % (k, worst[k]))


if __name__ == "__main__":
    main()
