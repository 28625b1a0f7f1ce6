#!/usr/bin/env python3
"""Stress harness for the adjoint of a coil-chunked SENSE tree (test infrastructure; GPU box only).

    python tests/stress_adjoint.py --reps 500 [--procs 4] [--idle 5] [--rebuild 50] [--coils 7 --chunk 2]

Why: round 3 saw `test_config5_coil_chunks_vs_oracle` (7 coils as 2 + 2 + 2 + 1 on a 256^3 grid: the slot scatter
k_grid_slots<2> / <1> under a chunk VStack) return ONE adjoint 1.6e-2 off the oracle in about thirty suite runs.  This
harness evaluates that adjoint over and over and checks every stage separately, so that a deviation names its kernel:

  scatter    KronI(nc, G')^H of the chunk's k-space rows into a grid POISONED with NaN beforehand.  Compared with the first
             evaluation BIT FOR BIT on every flagged segment of a brick no shared task touches (plain stores: any
             difference is a bug, not summation order), to 2e-6 (relative to the grid's largest value) on bricks whose
             pieces add with float atomics, and the unflagged segments must still be NaN (nobody may write them).  The
             first evaluation itself is checked against scipy's G^H k in complex128.
  transform  ZpadFFT^H of the reference grid with NaN in every unflagged segment (nobody may READ them) -- no atomics:
             bit for bit against the first evaluation.
  chain      the whole A^H k (scratch arena and dynamic scratch alternate), <= 1e-5 against the numpy oracle and 2e-6 against the
             first evaluation.

`--idle S` sleeps S seconds before every 10th evaluation (the one failure came after ~10 s of host-side oracle work: a GPU
waking from idle clocks); `--rebuild N` rebuilds the tree and all formats every N evaluations; `--procs P` runs the whole
thing in P fresh processes one after the other (first-touch / code-object-load effects).  Exit status 1 on any deviation;
every deviation is printed with the bricks it touches (shared or not), the chunk and its coil count.
Used by tests/test_hip_stress.py (a short version in the GPU suite)."""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

C64 = np.dtype('complex64')


def rel(a, b):
    den = np.linalg.norm(np.asarray(b).ravel())
    return float(np.linalg.norm((np.asarray(a) - np.asarray(b)).ravel()) / den) if den else 0.0


class Harness(object):
    def __init__(self, hip, image=160, coils=7, chunk=2, nspokes=300, nreadout=256, oversamp=1.6, seed=5, log=print, width=2):
        from indigo_amd.sense import SenseProblem
        from indigo_amd.util import rand64c
        self.hip, self.log = hip, log
        self.p = SenseProblem.synthetic((image,) * 3, coils, nspokes=nspokes, nreadout=nreadout, width=width, ntable=128,
                                        oversamp=oversamp, seed=seed, lazy_maps=True)
        self.chunk = chunk
        self.k = rand64c(self.p.T * coils, 1, seed=2)
        self.k_d = hip.copy_array(self.k)
        self.ref = None              # per-chunk reference grids / images, the chain's reference image
        self.build()

    # -- the tree and what the checks need to know about its formats ---------------------------------------------------------
    def build(self):
        from indigo_amd import operators as op
        hip, p = self.hip, self.p
        hip._scratch = None
        self.A = A = p.build_zpadfft(hip, chunk=self.chunk)
        self.children = list(A.children) if isinstance(A, op.VStack) else [A]
        n0, n1, n2 = p.oN
        self.P = n0 * n1 * n2
        self.info = []
        x_d = hip.zero_array((A.shape[1], 1), C64)
        A.eval(x_d, self.k_d, forward=False)                             # formats are built on first use
        hip.barrier()
        kz = np.arange(n2)
        for ch in self.children:
            # a chunk padded with zero-weight coils (operators.HeadRows): the tree underneath evaluates `width` coils, its k-space
            # rows beyond the real coils are zeros
            tree = ch.child if isinstance(ch, op.HeadRows) else ch
            G, Z = tree.left.right, tree.right                            # Product(KronI(nc, G'), ZpadFFT): KronI = Kron(Eye, G')
            M = G._matrix_d
            nc = tree.shape[0] // p.T                                     # panel width (with padding coils)
            # flagged segments of the (ky, kz, kx-tile) grid, from the support table THIS chunk's kernels write and read by (16 kx
            # points per entry, or the finer table of the 8- and 4-wide chunks): a tree of several widths carries several
            tile = int(Z._tile_kw.get('support_tile', 16))
            _, _, bits = p.split_support(Z._support_h, tile)
            nt = n0 // tile
            b = bits.reshape(n1, nt, 16)                                 # [ky, kxt, kz % 16] bit kz // 16
            flag = ((b[:, :, kz % 16] >> (kz // 16).astype(np.uint32)) & 1).astype(bool).transpose(0, 2, 1)   # [ky, kz, kxt]
            sl = M._format('_slots', nc, exact=True)
            br = M._format('_bricks', nc, exact=True)
            fmt = sl if sl is not None else br
            # (round 6) wide gridding kernels take the share scatter on the matrix cores (ig_grid_scatter_sep): its own bricks (16 x 4 x 4),
            # its own shared pieces; the rows of its shared table start with the brick id
            shr = getattr(M, '_shares_by', {}).get(nc) if (getattr(M, '_sep', None) is not None and hip.tuning.get('sep_scatter', True)) else None
            if shr is not None:
                fmt = dict(nshared=shr['nshared'], bm=shr['bm'], bs=shr['bs'],
                           shared=hip.copy_array(np.ascontiguousarray(shr['shared'].to_host().reshape(-1, 4)[:max(shr['nshared'], 1), 0].astype(np.int32))))
            shared = np.zeros((n1, n2, nt), dtype=bool)                  # segments of bricks whose pieces add with atomics
            if fmt is not None and fmt['nshared']:
                sb = fmt['shared'].to_host()[:fmt['nshared']].astype(np.int64)
                bm, bs = fmt['bm'], fmt['bs']
                nbx, nbm = n0 // 16, n2 // bm
                bx, bmi, bsi = sb % nbx, (sb // nbx) % nbm, sb // (nbx * nbm)
                for im in range(bm):
                    for is_ in range(bs):
                        for xs in range(16 // tile):
                            shared[bsi * bs + is_, bmi * bm + im, bx * (16 // tile) + xs] = True
            self.info.append(dict(nc=nc, real=ch.shape[0] // p.T, tree=tree, tile=tile, flag=flag,
                                  fmt='shares' if shr is not None else 'slots' if (fmt is sl and sl is not None) else 'bricks' if fmt is not None else 'gather',
                                  shared=shared, layout=Z._layout))

    def chunk_rows(self, i):
        lo = sum(c.shape[0] for c in self.children[:i])
        return lo, lo + self.children[i].shape[0]

    def grid_view(self, g, nc, layout, t):
        """host grid panel (P*nc,) -> [ky, kz, kx tile, point in tile, nc] view; t = kx points per support-table entry of the chunk"""
        n0, n1, n2 = self.p.oN
        if layout == 2:
            return g.reshape(n1, n2, n0 // t, t, nc)
        return g.reshape(nc, n1, n2, n0 // t, t).transpose(1, 2, 3, 4, 0)

    # -- stages ---------------------------------------------------------------------------------------------------------------
    def scatter(self, i, poison_d):
        hip = self.hip
        inf = self.info[i]
        grid_d = hip.empty_array((self.P * inf['nc'], 1), C64)
        grid_d._copy(poison_d[inf['nc']])
        inf['tree'].left.eval(grid_d, self.chunk_k(i), forward=False)
        return grid_d.to_host().reshape(-1)

    def chunk_k(self, i):
        """the chunk's k-space rows on the device, zero rows appended for its padding coils"""
        inf = self.info[i]
        lo, hi = self.chunk_rows(i)
        if inf['real'] == inf['nc']:
            return self.k_d[lo:hi]
        kp = np.zeros((self.p.T * inf['nc'], 1), dtype=C64)
        kp[:hi - lo] = self.k[lo:hi]
        return self.hip.copy_array(kp)

    def transform(self, i, grid_h):
        hip = self.hip
        tree = self.info[i]['tree']
        img_d = hip.empty_array((tree.shape[1], 1), C64)
        img_d._copy(self.nan_img_d)
        g_d = hip.copy_array(grid_h.reshape(-1, 1))
        tree.right.eval(img_d, g_d, forward=False)
        return img_d.to_host().reshape(-1)

    def chain(self, arena):
        from indigo_amd.transforms import reserve_for
        hip = self.hip
        if arena:
            reserve_for(self.A, 1)
        else:
            hip._scratch = None
        y_d = hip.empty_array((self.A.shape[1], 1), C64)
        y_d._copy(self.nan_img_d)
        self.A.eval(y_d, self.k_d, forward=False)
        out = y_d.to_host().reshape(-1)
        hip._scratch = None
        return out

    # -- references ---------------------------------------------------------------------------------------------------------
    def make_references(self, oracle_backend=None):
        hip, p = self.hip, self.p
        self.nan_img_d = hip.copy_array(np.full((self.A.shape[1], 1), np.nan + 1j * np.nan, dtype=C64))
        self.poison = {}
        for nc in sorted({inf['nc'] for inf in self.info}):
            self.poison[nc] = hip.copy_array(np.full((self.P * nc, 1), np.nan + 1j * np.nan, dtype=C64))
        Gm = p.fused_interp(1)
        GH = Gm.conj().T.tocsr().astype(np.complex128)
        self.ref = dict(grid=[], img=[], chain=None)
        for i, inf in enumerate(self.info):
            lo, hi = self.chunk_rows(i)
            flag, tile = inf['flag'], inf['tile']
            g = self.scatter(i, self.poison)
            v = self.grid_view(g, inf['nc'], inf['layout'], tile)
            # nobody writes unflagged segments; flagged ones equal G^H k
            assert np.isnan(v[~flag].real).all(), "chunk %d: the scatter wrote an unflagged segment" % i
            kc = np.zeros((p.T, inf['nc']), dtype=np.complex128)
            kc[:, :inf['real']] = self.k[lo:hi].reshape(p.T, inf['real'], order='F')               # (padding coils: zero rows)
            exp = GH @ kc                                                                           # (P, nc), layout-1 rows
            n0, n1, n2 = p.oN
            e = exp.reshape(n1, n2, n0 // tile, tile, inf['nc'])
            err = np.linalg.norm((v[flag] - e[flag]).ravel()) / np.linalg.norm(e[flag].ravel())
            assert err < 1e-5, "chunk %d (%d coils, %s): first scatter %.3e off scipy's G^H k" % (i, inf['nc'], inf['fmt'], err)
            assert np.count_nonzero(e[~flag]) == 0
            self.ref['grid'].append(g)
            self.ref['img'].append(self.transform(i, g))
            self.log("chunk %d: %d coil(s), layout %d, %s, %d shared segments of %d flagged; scatter vs scipy %.2e"
                     % (i, inf['nc'], inf['layout'], inf['fmt'], int((inf['shared'] & flag).sum()), int(flag.sum()), err))
        self.ref['chain'] = self.chain(arena=False)
        total = np.sum(self.ref['img'], axis=0)
        self.log("chain vs sum of stage images: %.2e" % rel(self.ref['chain'], total))
        assert rel(self.ref['chain'], total) < 2e-6
        if oracle_backend is not None:
            A_o = p.build_zpadfft(oracle_backend, layout=0, support=False)
            exp = (A_o.H * self.k).reshape(-1)
            e = rel(self.ref['chain'], exp)
            self.log("chain vs numpy oracle: %.2e" % e)
            assert e < 1e-5, e

    # -- one round of checks; returns a list of findings ----------------------------------------------------------------------
    def check(self, it):
        bad = []
        for i, inf in enumerate(self.info):
            g = self.scatter(i, self.poison)
            flag = inf['flag']
            v, r = self.grid_view(g, inf['nc'], inf['layout'], inf['tile']), self.grid_view(self.ref['grid'][i], inf['nc'], inf['layout'], inf['tile'])
            if not np.isnan(v[~flag].real).all():
                bad.append("it %d chunk %d (%d coils, %s): %d values written into unflagged segments"
                           % (it, i, inf['nc'], inf['fmt'], int((~np.isnan(v[~flag].real)).sum())))
            own = flag & ~inf['shared']
            same = (v.view(np.uint32) == r.view(np.uint32)).reshape(v.shape[:3] + (-1,)).all(axis=3)     # per segment
            diff = own & ~same
            if diff.any():
                ky, kz, kxt = np.nonzero(diff)
                dv = np.abs(np.nan_to_num(v[diff], nan=1e30) - r[diff]).max()
                bad.append("it %d chunk %d (%d coils, %s): %d NON-SHARED segments differ bitwise (max |d| %.3e, grid max %.3e); first (ky,kz,kxt): %s; nan %d zero %d"
                           % (it, i, inf['nc'], inf['fmt'], int(diff.sum()), dv, np.abs(r[flag]).max(),
                              list(zip(ky[:6].tolist(), kz[:6].tolist(), kxt[:6].tolist())),
                              int(np.isnan(v[diff].real).sum()), int((v[diff] == 0).sum())))
            sh = flag & inf['shared']
            if sh.any():
                scale = np.abs(r[sh]).max()
                d = np.abs(np.nan_to_num(v[sh], nan=1e30) - r[sh]).max() / scale
                if d > 2e-6:
                    seg = sh & ~same
                    ky, kz, kxt = np.nonzero(seg)
                    bad.append("it %d chunk %d (%d coils, %s): SHARED segments off by %.3e of the grid max; %d segments; first %s"
                               % (it, i, inf['nc'], inf['fmt'], d, int(seg.sum()), list(zip(ky[:6].tolist(), kz[:6].tolist(), kxt[:6].tolist()))))
            # the reader: reference grid with NaN in unflagged segments
            img = self.transform(i, self.ref['grid'][i])
            if not np.array_equal(img.view(np.uint32), self.ref['img'][i].view(np.uint32)):
                bad.append("it %d chunk %d (%d coils): cropped transform differs bitwise from its first evaluation: rel %.3e, nan %d"
                           % (it, i, inf['nc'], rel(np.nan_to_num(img), self.ref['img'][i]), int(np.isnan(img.real).sum())))
        out = self.chain(arena=bool(it & 1))
        e = rel(np.nan_to_num(out, nan=1e30), self.ref['chain'])
        if not e < 2e-6:
            bad.append("it %d chain (%s scratch): %.3e off the first evaluation; nan %d" % (it, "arena" if it & 1 else "dynamic", e, int(np.isnan(out.real).sum())))
        return bad


def run(reps, idle=0.0, rebuild=0, oracle=True, log=print, **kw):
    from indigo_amd.backends import get_backend
    hip = get_backend("hip")
    ob = None
    if oracle:
        from oracle.np_backend import NumpyBackend
        ob = NumpyBackend()
    h = Harness(hip, log=log, **kw)
    h.make_references(ob)
    findings = []
    t0 = time.time()
    for it in range(reps):
        if rebuild and it and it % rebuild == 0:
            h.build()
        if idle and it % 10 == 9:
            time.sleep(idle)
        bad = h.check(it)
        for b in bad:
            log("DEVIATION " + b)
        findings += bad
        if it % 50 == 49 or it == reps - 1:
            log("  %d evaluations, %d deviations, %.0f s" % (it + 1, len(findings), time.time() - t0))
    return findings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--idle", type=float, default=0.0)
    ap.add_argument("--rebuild", type=int, default=0)
    ap.add_argument("--image", type=int, default=160)
    ap.add_argument("--coils", type=int, default=7)
    ap.add_argument("--chunk", type=int, default=2)
    ap.add_argument("--width", type=float, default=2, help="half-width of the gridding kernel (3: the share scatter on the matrix cores)")
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    if a.procs > 1:                         # fresh processes, one after the other; this parent never touches the GPU
        rc = 0
        for pi in range(a.procs):
            cmd = [sys.executable, os.path.abspath(__file__), "--reps", str(a.reps), "--idle", str(a.idle), "--rebuild", str(a.rebuild),
                   "--image", str(a.image), "--coils", str(a.coils), "--chunk", str(a.chunk), "--width", str(a.width)] + (["--no-oracle"] if a.no_oracle or pi else [])
            print("== process %d of %d" % (pi + 1, a.procs), flush=True)
            rc |= subprocess.call(cmd)
        sys.exit(rc)
    f = run(a.reps, idle=a.idle, rebuild=a.rebuild, oracle=not a.no_oracle, image=a.image, coils=a.coils, chunk=a.chunk, width=(int(a.width) if a.width == int(a.width) else a.width),
            log=lambda s: print(s, flush=True))
    print("stress_adjoint: %d deviations in %d evaluations" % (len(f), a.reps), flush=True)
    sys.exit(1 if f else 0)


if __name__ == "__main__":
    main()
