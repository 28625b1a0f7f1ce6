"""Non-Cartesian SENSE operators: synthetic problem generator and tree builders.

Two ways to build the forward operator A (k-space <- image) on a backend:

  * `build_tree(...)`      the reference's route: `NUFFT` / `KronI` / `VStack(Diag(maps))` factories
                           (examples/pics.py:92-95), optionally rewritten by `sense_recipe(level)`
                           (pics.py:179-193).  Exact drop-in, scipy does the matrix algebra.
  * `build_fused(...)`     constructs the two CSR factors of the `-O3` tree directly,
                               A = KronI(C, G') * ( KronI(C, UnscaledFFT) * S' )
                               G' = interp * mod * (1/sqrt(P))            (T x P)
                               S' = (I_C (x) mod*zpad*apod) * maps        ((C*P) x N, stored transposed)
                           without any scipy sparse product, so 256^3 x 8-coil problems set up in
                           seconds.  tests/ checks it against `build_tree(level=3)` and the goldens.

Coil sharding: `coils=range(lo, hi)` builds the operator for a subset of coils; the
partial adjoint images of the shards sum to the full A^H k, which is the one all-reduce of AHA.
"""
import numpy as np
import scipy.sparse as spp
from scipy.signal.windows import kaiser

from indigo_amd import fused
from indigo_amd import operators as op
from indigo_amd.interp import interp_csr_arrays, interp_csr_modulated, interp_sep_records
from indigo_amd.noncart import rolloff3
from indigo_amd.transforms import sense_recipe, reserve_for

_C64 = np.dtype('complex64')


def radial_trajectory(nspokes, nreadout, seed=3):
    """3-D radial ("kooshball") trajectory: `nspokes` diameters with seeded uniform-sphere directions,
    `nreadout` samples each on [-0.5, 0.5).  Returns coord of shape (3, nreadout, nspokes)."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(-1.0, 1.0, nspokes)
    phi = rng.uniform(0.0, 2 * np.pi, nspokes)
    s = np.sqrt(1.0 - z * z)
    d = np.stack([s * np.cos(phi), s * np.sin(phi), z])                 # (3, nspokes)
    t = (np.arange(nreadout) - nreadout // 2) / nreadout               # [-0.5, 0.5)
    return d[:, None, :] * t[None, :, None]


class SenseProblem(object):
    """Host-side description of one non-Cartesian SENSE reconstruction."""

    def __init__(self, N, coord, maps, width=2, ntable=128, oversamp=2.0, ncoils=None):
        self.N = tuple(int(n) for n in N)
        self.coord = np.asarray(coord, dtype=np.float64)
        assert self.coord.shape[0] == 3
        # maps: (N0, N1, N2, C) complex64, F-ordered -- or a callable c -> (N0, N1, N2) map of coil c (with `ncoils`),
        # so that a rank of a coil-sharded run only ever materialises its own coils (32 coils x 320^3 = 8.4 GB)
        self.maps = maps
        self.C = int(ncoils) if callable(maps) else int(maps.shape[3])
        self._interp_cache = {}
        self.width, self.ntable, self.oversamp = width, int(ntable), float(oversamp)
        self.oN = tuple(int(n * self.oversamp) for n in self.N)
        self.T = int(np.prod(self.coord.shape[1:]))
        self.ksp_dims = (1,) + tuple(self.coord.shape[1:])
        omin = self.oversamp
        self.beta = np.pi * np.sqrt(((width * 2. / omin) * (omin - 0.5)) ** 2 - 0.8)
        self.table = kaiser(2 * self.ntable + 1, self.beta)[self.ntable:]

    @classmethod
    def synthetic(cls, N, C, nspokes, nreadout, width=2, ntable=128, oversamp=2.0, seed=4, lazy_maps=False):
        """Seeded synthetic problem: radial trajectory + uniform random complex maps.
        lazy_maps: coil c's map is generated on demand from the seed (seed, c) instead of one (N, C) array up front."""
        from indigo_amd.util import rand64c
        coord = radial_trajectory(nspokes, nreadout, seed=seed - 1)
        if lazy_maps:
            N = tuple(int(n) for n in N)
            return cls(N, coord, lambda c: rand64c(*N, seed=[seed, int(c)]), width=width, ntable=ntable,
                       oversamp=oversamp, ncoils=C)
        # coil c's map comes from the seed (seed, c) either way; here all of them are generated up front, a few threads sharing
        # the coils (each with its own generator)
        N = tuple(int(n) for n in N)
        maps = np.empty(N + (C,), dtype=_C64, order='F')

        def one(c):
            maps[:, :, :, c] = rand64c(*N, seed=[seed, int(c)])
        if C > 1 and int(np.prod(N)) >= 1 << 20:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(8, C)) as ex:
                list(ex.map(one, range(C)))
        else:
            for c in range(C):
                one(c)
        return cls(N, coord, maps, width=width, ntable=ntable, oversamp=oversamp)

    def drop_cache(self):
        self._interp_cache = {}

    def coil_map(self, c):
        """(N0, N1, N2) sensitivity map of coil c"""
        return self.maps(c) if callable(self.maps) else self.maps[:, :, :, c]

    # -- the reference's route ------------------------------------------------------------
    def build_tree(self, backend, level=0, coils=None):
        coils = range(self.C) if coils is None else coils
        F1 = backend.NUFFT(self.ksp_dims, self.N, self.coord, width=self.width, n=self.ntable,
                           oversamp=self.oversamp, dtype=_C64)
        F = backend.KronI(len(coils), F1)
        S = backend.VStack([backend.Diag(self.coil_map(c)[..., None]) for c in coils], name='maps')
        A = F * S
        A._name = 'SENSE1'
        if level:
            for Step in sense_recipe(level):
                A = Step().visit(A)
        return A

    # -- direct construction of the -O3 factors ------------------------------------------------
    def fused_interp(self, layout=0, phases=None):
        """G' = interp * mod * (1/sqrt(P)) as a complex64 CSR (T x P).
        layout=1 indexes the grid columns in (x, z, y) memory order (see operators.ZpadFFT).
        phases: per-axis phase arrays to use INSTEAD of the centred transform's modulation -- a leaf whose transform carries the
        modulation of an odd axis itself (HipBackend.fold_axis_shifts) asks for the matrix without it; not cached under `layout`"""
        if phases is not None:
            key = ('folded', layout)
            if key not in self._interp_cache:
                P = int(np.prod(self.oN))
                scale = np.float32(1.0) / np.sqrt(np.float32(P))
                indptr, indices, data = interp_csr_modulated(self.T, self.oN, self.width, self.table, self.coord.reshape(3, -1, order='F'),
                                                             phases, scale, grid_order=1 if layout == 1 else 0)
                self._interp_cache[key] = spp.csr_matrix((data, indices, indptr), shape=(self.T, P))
            return self._interp_cache[key]
        if layout in self._interp_cache:
            return self._interp_cache[layout]
        P = int(np.prod(self.oN))
        n0, n1, n2 = self.oN
        # the native builder numbers the grid columns in either order (sorted within a row both ways) and applies the centred
        # transform's modulation and the 1/sqrt(P) in the same pass (ig_interp3_fill_modulated)
        scale = np.float32(1.0) / np.sqrt(np.float32(P))
        indptr, indices, data = interp_csr_modulated(self.T, self.oN, self.width, self.table, self.coord.reshape(3, -1, order='F'),
                                                     _mod_axis_phases(self.oN), scale, grid_order=1 if layout == 1 else 0)
        G = spp.csr_matrix((data, indices, indptr), shape=(self.T, P))
        self._interp_cache[layout] = G            # 0.6 GB per layout at 5e7 nonzeros; drop_cache() releases them
        return G

    def fused_interp_sep(self, layout=1, phases=None):
        """G' in separable form (indigo_amd.interp.interp_sep_records: one record per sample -- first tap, tap counts and per-axis
        weights with the modulation's sign folded in) for the grid order of `layout`, or None when the grid's modulation is no sign
        per axis (odd axes -- unless the leaf's transform carries their modulation: `phases`, as for fused_interp)"""
        key = ('sep', layout) if phases is None else ('sep-folded', layout)
        if key not in self._interp_cache:
            P = int(np.prod(self.oN))
            scale = np.float32(1.0) / np.sqrt(np.float32(P))
            self._interp_cache[key] = interp_sep_records(self.T, self.oN, self.width, self.table, self.coord.reshape(3, -1, order='F'),
                                                         _mod_axis_phases(self.oN) if phases is None else phases, scale, grid_order=1 if layout >= 1 else 0)
        return self._interp_cache[key]

    def fused_maps_T(self, coils=None):
        """S'^H stored form: CSR of shape (N, C*P) whose adjoint is S' = (I_C (x) mod*zpad*apod) * maps."""
        coils = list(range(self.C) if coils is None else coils)
        Cn = len(coils)
        Nn, P = int(np.prod(self.N)), int(np.prod(self.oN))
        assert Cn * P < 2 ** 31, "int32 column indices: shard the coils"
        from indigo_amd.backends.backend import Backend
        zrows = Backend.zpad_rows(self.oN, self.N)                       # (Nn,) positions in the padded grid
        mod = fftc_mod_box(self.oN, self.N).reshape(-1, order='F')
        apod = rolloff3(self.oversamp, self.width, self.beta, self.N).reshape(-1, order='F').astype(_C64)
        base = (mod * apod).astype(_C64)                                 # per-voxel factor shared by all coils
        # row i holds, for each coil c, conj(base[i]*maps[i,c]) at column c*P + zrows[i]
        indptr = np.arange(0, (Nn + 1) * Cn, Cn, dtype=np.int64)
        indices = np.empty((Nn, Cn), dtype=np.int32)
        data = np.empty((Nn, Cn), dtype=_C64)
        for j, c in enumerate(coils):
            indices[:, j] = (zrows + j * P).astype(np.int32)
            data[:, j] = np.conj(base * self.coil_map(c).reshape(Nn, order='F'))
        return spp.csr_matrix((data.reshape(-1), indices.reshape(-1), indptr), shape=(Nn, Cn * P))

    def fused_weights(self, coils=None, interleaved=False, scale=1.0):
        """w[..., c] = mod(box) * apod * maps[..., c]: the per-voxel, per-coil factor of S' (F-ordered, box + (C,)).
        interleaved: the same values and shape with a voxel's coils side by side in memory (what the coil-interleaved grid
        layout uploads: no transposition of a 2 GB array on the way).
        scale: a complex constant multiplied in (the constant of the k-space modulation that a leaf with a real gridding matrix moves
        over here, HipBackend.split_gridding_constant)"""
        coils = list(range(self.C) if coils is None else coils)
        from indigo_amd.backends.backend import Backend
        bkey = 'weights_base' if complex(scale) == 1.0 else ('weights_base', complex(scale))
        base = self._interp_cache.get(bkey)       # the same for every coil chunk of a tree (config 5: four of them)
        if base is None:
            mod = fftc_mod_box(self.oN, self.N)
            apod = rolloff3(self.oversamp, self.width, self.beta, self.N).astype(_C64)
            base = (mod * apod).astype(_C64) if complex(scale) == 1.0 else (mod * (apod * complex(scale))).astype(_C64)
            self._interp_cache[bkey] = base
        if interleaved and len(coils) > 1:
            w = np.empty(self.N[::-1] + (len(coils),), dtype=_C64).transpose(2, 1, 0, 3)
            nz, nth = self.N[2], (8 if base.size >= 1 << 20 else 1)

            def slab(k):                                 # threads own z slabs (coil-parallel writes would share cache lines)
                z0, z1 = k * nz // nth, (k + 1) * nz // nth
                for j in range(len(coils)):
                    np.multiply(base[:, :, z0:z1], cmaps[j][:, :, z0:z1], out=w[:, :, z0:z1, j])
            if nth > 1:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=nth) as ex:
                    cmaps = list(ex.map(self.coil_map, coils))      # (lazy maps are generated here, coil-parallel)
                    list(ex.map(slab, range(nth)))
            else:
                cmaps = [self.coil_map(c) for c in coils]
                slab(0)
            return w
        w = np.empty(self.N + (len(coils),), dtype=_C64, order='F')

        def one(jc):
            np.multiply(base, self.coil_map(jc[1]), out=w[:, :, :, jc[0]])
        if len(coils) > 1 and base.size >= 1 << 20:      # numpy releases the GIL in the multiply: a few threads share the coils
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(8, len(coils))) as ex:
                list(ex.map(one, enumerate(coils)))
        else:
            for jc in enumerate(coils):
                one(jc)
        return w

    def grid_support(self, G, tile=16, zw=(16, 16)):
        """k-space support table of a layout-1 gridding matrix (see indigo_amd.fused.grid_support)"""
        return fused.grid_support(G, self.oN, tile, zw)

    def split_support(self, table, tile=16, zw_in=None):
        return fused.split_support(table, self.oN, tile, int(zw_in or getattr(self, 'last_support_zw', (16, 16))[0]))

    def zpadfft_pass_bytes(self, ncoils, table=None, fused_sum=False, tile=16):
        """Compulsory HBM bytes of each axis pass of the fused transform (layout 1): what the pass must read
        plus what it must write, given the image box and -- if present -- the k-space support table.
        These are the per-launch "algorithmic bytes" bench.py prices the fused passes with (the reference
        has no equivalent kernels; the rocprofv3 PMC traffic agrees with these figures)."""
        n0, n1, n2 = self.oN
        b0, b1, b2 = self.N
        C, nt = ncoils, n0 // tile                  # `tile` kx points per support-table entry
        bvol, cvol, vol = b0 * b1 * b2, n0 * b1 * b2, n0 * n1 * n2
        if table is None:
            z_sup, z_tiles = vol, n1 * nt
            ylen = np.full(nt, n1, dtype=np.int64)
        else:
            zr, yr, bits = self.split_support(table, tile)
            z_sup = int(np.unpackbits(bits.view(np.uint8)).sum()) * tile   # grid points in flagged segments, per coil
            z_tiles = int(np.count_nonzero(zr[:, 1] > zr[:, 0]))           # (kx tile, ky) columns with any support
            ylen = (yr[:, 1].astype(np.int64) - yr[:, 0]).clip(0)
        y_sup = int(ylen.sum()) * tile * b2          # points the y pass produces / consumes on its grid side
        e = 8 * C
        return {
            "fft_pad_x": bvol * 8 + bvol * e + cvol * e,
            "fft_pad_y": cvol * e + y_sup * e,
            "fft_pad_z": z_tiles * tile * b2 * e + z_sup * e,
            "fft_crop_z": z_sup * e + y_sup * e,            # writes every column the y pass will read (zeros where the hull is empty)
            "fft_crop_y": y_sup * e + cvol * e,
            # (fused_sum: the coil combination happens inside the pass -- one image box is written, not one per coil)
            "fft_crop_x": cvol * e + bvol * e + (bvol * 8 if fused_sum else bvol * e),
        }

    def gridding_pass_bytes(self, ncoils, table=None, tile=16, real_entries=False):
        """Compulsory HBM bytes of the two gridding products of the fused tree for `ncoils` coils: the matrix once, the
        panel rows that are really touched once, the result once.  The adjoint (a gather over G'^T restricted to the
        flagged 16-row segments of the support table) reads the transposed matrix's row pointers only inside flagged
        segments and writes only those segments.  (The reference's model, operators.py:246-256, prices the adjoint
        with a full read-modify-write of the 8.6 GB grid panel; both figures are reported by bench.py.)
        real_entries: the formats store real weights (4 bytes per value instead of 8, see HipBackend weights_are_real)."""
        Gm = self.fused_interp(1)
        T, P = Gm.shape
        nnz = Gm.nnz
        touched = int(fused.touched_columns(Gm).size)
        e = 8 * ncoils
        if table is not None:
            _, _, bits = self.split_support(table, tile)
            sup = int(np.unpackbits(bits.view(np.uint8)).sum()) * tile
        else:
            sup = P
        return {
            "csrmm_gather": nnz * (8 if real_entries else 12) + (T + 1) * 4 + touched * e + T * e,
            "csrmm_rowlane_conj": nnz * 12 + (sup + 1) * 4 + T * e + sup * e,
            # brick-binned scatter: 12 bytes per nonzero (cell + value; 8 when the weights are real and stored so; the padding and the
            # row list of the binned format are overhead, not compulsory), the panel, the flagged rows
            "csrmm_bricks_conj": nnz * (8 if real_entries else 12) + T * e + sup * e,
            # slot-format scatter (1- and 2-coil ranks): 16 bytes per nonzero (cell, value, sample), the panel, the flagged rows
            "csrmm_slots_conj": nnz * (12 if real_entries else 16) + T * e + sup * e,
            "pack_panel": 2 * T * e,
            # round 6, taps computed from the separable records: a 64-byte record per sample instead of the stored taps; the scatter
            # of (sample, brick) shares reads record + panel row once per share (its 8-byte header beside them) -- the shares
            # themselves are the format's choice, so the compulsory figure prices one per sample
            "grid_gather_sep": T * 64 + touched * e + T * e,
            "grid_scatter_sep": T * (64 + 8) + T * e + sup * e,
        }

    @staticmethod
    def locality_order(G):
        """Row order of a gridding matrix that keeps spatial neighbours together: samples sorted by the first
        (lowest) grid column they touch, i.e. lexicographically by grid cell in the grid's memory order."""
        first = np.full(G.shape[0], np.iinfo(np.int64).max, dtype=np.int64)
        nz = np.flatnonzero(np.diff(G.indptr))
        first[nz] = G.indices[G.indptr[nz]]
        return np.argsort(first, kind='stable').astype(np.int32)

    coil_chunks = staticmethod(fused.coil_chunks)

    def build_zpadfft(self, backend, coils=None, layout=None, support=None, reorder=False, chunk=8):
        """A = KronI(C, G') * ZpadFFT: the `-O3` tree with S' and the FFT fused into one leaf
        (zero-pad aware transform; needs backend.supports_padded_fft(grid)).  The oversampled grid is
        private to this pair of leaves, so it may live in the (x, z, y) order (layout=1) that keeps the
        transform's largest pass at a small stride; G' is indexed to match.

        More coils than `chunk` become a VStack of `chunk`-coil trees that share ONE device copy of G' (and of its
        transpose and support table): the 512^3 x 32-coil grid of BASELINE config 5 would be 34 GB, a chunk is 8.6 GB
        and stays the size the kernels were tuned on (indigo_amd.fused.assemble)."""
        coils = list(range(self.C) if coils is None else coils)
        Cn = len(coils)
        single_ok = getattr(backend, 'supports_single_coil_layout', lambda g: True)(self.oN)
        tuning = getattr(backend, 'tuning', {})
        layout, chunks = fused.choose_layout(Cn, chunk, layout, single_ok, tuning.get('chunk_cost'), tuning.get('chunk_pad', True))
        # odd axes whose transform pass can carry the centred transform's modulation itself (chirp-z axes: a circular shift folded into
        # the pass's tables): the gridding matrix is then built without it and keeps real weights (HipBackend.fold_axis_shifts)
        kshift, folded, gconst = None, None, 1.0
        if layout == 2 and hasattr(backend, 'fold_axis_shifts'):
            kshift, folded = backend.fold_axis_shifts(self.oN, _mod_axis_phases(self.oN))
            # ... and what is left of the modulation as (a constant) x (a sign per axis): the constant goes to the transform's weights
            gconst, split = backend.split_gridding_constant(folded if folded is not None else _mod_axis_phases(self.oN))
            folded = split if split is not None else folded
        Gm = self.fused_interp(1 if layout == 2 else layout, phases=folded)      # layout 2 = layout 1 with the coils interleaved below
        table = None
        zw = fused.support_words(backend, self.oN)      # words per entry of the table's bitmaps: follows from the z pass's kernel
        if (support is None or support) and layout >= 1 and zw is not None and (layout == 2 or zw == (16, 16)):
            # restrict the transform's z pass and the adjoint gridding to the k-space support of G'
            table = self.grid_support(Gm, 16, zw)
        zw = zw or (16, 16)
        self.last_support_table = table
        self.last_support_zw = zw
        order = self.locality_order(Gm) if reorder and Cn <= 8 else None
        widths = {lo: w for lo, _, w in chunks}
        A = fused.assemble(backend, Gm, self.oN, self.N, lambda lo, hi: self.fused_weights(coils[lo:hi], interleaved=(widths.get(lo, 0) > 1), scale=gconst), Cn,
                           layout, chunks, table=table, row_order=order, zw=zw,
                           sep=self.fused_interp_sep(1, phases=folded) if (layout == 2 and order is None) else None, kshift=kshift)
        self.last_support_fine = getattr(A, '_support_fine', None)       # (table, tile) when the tree took a finer table
        return A

    def build_fused(self, backend, coils=None):
        coils = list(range(self.C) if coils is None else coils)
        Cn = len(coils)
        G = backend.SpMatrix(self.fused_interp(), name='interp*mod*scale')
        F = backend.UnscaledFFT(self.oN, dtype=_C64, name='fft')
        St = backend.SpMatrix(self.fused_maps_T(coils), name='((x)mod*zpad*apod)*maps.H')
        A = backend.KronI(Cn, G) * (backend.KronI(Cn, F) * St.H)
        A._name = 'SENSE-O3'
        return A


def backend_mod(ft_shape):
    from indigo_amd.backends.backend import Backend
    return Backend.fftc_mod(ft_shape, _C64)


def _mod_axis_phases(ft_shape):
    """per-axis terms (idx - c/2) * (c/n), c = n // 2, of the centred FFT's modulation phase (Backend.fftc_mod)"""
    out = []
    for n in ft_shape:
        c = n // 2
        out.append((np.arange(n) - c / 2.0) * (c / n))
    return out


def fftc_mod_at(ft_shape, kx, ky, kz):
    """Backend.fftc_mod(ft_shape)[kx, ky, kz] without the full grid: the phase is a sum of per-axis terms, added in
    the same order (x, then y, then z) and precision as the full-grid formula, so the values are bit-identical."""
    px, py, pz = _mod_axis_phases(ft_shape)
    phase = 0 + px[kx]
    phase += py[ky]
    phase += pz[kz]
    return np.exp(1j * 2.0 * np.pi * phase).astype(_C64)


def fftc_mod_box(ft_shape, box):
    """the modulation on the centred box (Backend.Zpad's placement) of the grid, shape `box`.  The phase is a sum of per-axis
    terms: the product of three per-axis exponentials (complex128, rounded once) replaces 1.7e7 complex exponentials."""
    sl = [slice(m // 2 + int(np.ceil(-n / 2)), m // 2 + int(np.ceil(n / 2))) for m, n in zip(ft_shape, box)]
    ex, ey, ez = (np.exp(1j * 2.0 * np.pi * ph[s]) for ph, s in zip(_mod_axis_phases(ft_shape), sl))
    return np.asfortranarray((ex[:, None, None] * ey[None, :, None] * ez[None, None, :]).astype(_C64))


def normal_operator(A, lamda=0.0, ncols=1):
    """AHA = A^H A (+ lamda I), with the scratch arena sized for it (examples/pics.py:195)."""
    AHA = A.H * A
    if lamda:
        # lamda*I + A^H A rather than A^H A + lamda*I: a Sum evaluates its right child first, with the caller's beta,
        # and its left child on top (operators.py Sum._eval, reference operators.py:566-567), so this order lets the
        # adjoint's last leaf run with beta = 0 -- the form its fused kernels (coil combination inside the last
        # transform pass) take -- and adds lamda*x with one axpby
        AHA = lamda * A._backend.Eye(A.shape[1]) + AHA
    AHA._name = 'SENSE'
    reserve_for(AHA, ncols)
    return AHA
