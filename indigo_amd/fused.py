"""Assembly of the fused SENSE tree  A = KronI(C, G'_perm) * ZpadFFT  from its ingredients.

Shared by `SenseProblem.build_zpadfft` (which builds the ingredients directly) and by the `FuseZpadFFT`
transform (which recovers them from the reference's `-O3` tree, examples/pics.py:104-193).  The oversampled
k-space grid exists only between these two leaves, so its memory order is theirs to choose:

  layout 0   (x, y, z) per coil      the reference's Fortran order
  layout 1   (x, z, y) per coil      keeps the transform's largest pass at a small stride
  layout 2   (c, x, z, y)            layout 1 with the coils interleaved below x

and the gridding matrix's columns are renumbered to match.
"""
import numpy as np
import scipy.sparse as spp

_C64 = np.dtype('complex64')


def permute_grid_columns(G, oN):
    """Renumber the columns of a gridding matrix from the (x, y, z) grid order to (x, z, y); sorted CSR."""
    n0, n1, n2 = (int(n) for n in oN)
    G = G.tocsr()
    idx = G.indices.astype(np.int64)
    kx = idx % n0
    ky = (idx // n0) % n1
    kz = idx // (n0 * n1)
    out = spp.csr_matrix((G.data, (kx + n0 * (kz + n2 * ky)).astype(np.int32), G.indptr), shape=G.shape)
    out.sort_indices()
    return out


def touched_columns(G):
    """sorted indices of the columns of G that hold a nonzero (the grid points a gridding matrix touches); one flag pass
    instead of a sort of all column indices, cached on the matrix"""
    c = getattr(G, '_ig_touched_columns', None)
    if c is None:
        mark = np.zeros(G.shape[1], dtype=bool)
        mark[G.indices] = True
        c = np.flatnonzero(mark)
        try:
            G._ig_touched_columns = c
        except AttributeError:
            pass
    return c


def grid_support(G, oN, tile=16, zw=(16, 16)):
    """k-space support table of a layout-1 gridding matrix (see grid_support_numpy for the format): the library's native
    host routine, one threaded pass over the column indices (ig_grid_support).  zw = (words per entry of the bitmaps as the
    z pass reads them on its input side, ... on its output side): (16, 16) for 256- and 512-point z axes, (B, A) for an axis the
    A x B kernel transforms (backend.support_words(n2))."""
    from indigo_amd import _lib
    n0, n1, n2 = (int(n) for n in oN)
    zi, zo = (int(v) for v in zw)
    assert n0 % tile == 0 and n2 <= 32 * min(zi, zo) and tile in (2, 4, 8, 16)
    nt = n0 // tile
    idx = np.ascontiguousarray(G.indices, dtype=np.int32)
    table = np.empty(2 * (n1 * nt + nt) + 2 * n1 * nt * (zi + (zo if zo != zi else 0)), dtype=np.int16)
    _lib.check(_lib.lib().ig_grid_support(idx.size, idx.ctypes.data, n0, n1, n2, int(tile), zi, zo, table.ctypes.data), None, "ig_grid_support")
    return table


def grid_support_numpy(G, oN, tile=16, zw=(16, 16)):
    """k-space support of a layout-1 gridding matrix G (T x P) as the flat int16 table ig_fft_exec_padded,
    ig_fft_exec_cropped and ig_ccsrmm_t_grid take.  Three parts:
      1. [z_lo, z_hi) per (ky, 16-wide kx tile): the kz range outside which no sample touches the grid;
      2. [y_lo, y_hi) per kx tile: the ky range with a non-empty z range (the transform's y pass);
      3. 16 uint32 words per (ky, kx tile): bit m of word t is set iff the 16-row segment (kx tile, ky,
         kz = t + 16*m) holds a nonzero of G.  Segments without a nonzero are never gridded from, never
         written by the adjoint gridding and read as zero by the cropped transform.
    A radial trajectory fills a ball (half of the grid cube lies outside) and, away from the centre, leaves
    gaps between spokes: 30 % of the 16-row segments of the 512^3 grid of the headline problem are flagged."""
    n0, n1, n2 = (int(n) for n in oN)
    zi, zo = (int(v) for v in zw)          # words per entry: bit kz // z of word kz % z (input-side form, then the output-side form)
    assert n0 % tile == 0 and n2 <= 32 * min(zi, zo) and tile in (2, 4, 8, 16)
    nt = n0 // tile               # `tile` kx points per entry (16 unless the caller asked for a finer table, see ig_fft_set_support_tile)
    cols = touched_columns(G)
    kx = cols % n0
    kz = (cols // n0) % n2
    ky = cols // (n0 * n2)
    key = ky * nt + kx // tile
    bits = np.zeros((n1 * nt, zi), dtype=np.uint32)
    np.bitwise_or.at(bits, (key, kz % zi), np.uint32(1) << (kz // zi).astype(np.uint32))
    bits_out = None
    if zo != zi:
        bits_out = np.zeros((n1 * nt, zo), dtype=np.uint32)
        np.bitwise_or.at(bits_out, (key, kz % zo), np.uint32(1) << (kz // zo).astype(np.uint32))
    order = np.argsort(key, kind='stable')
    key, kz = key[order], kz[order]
    ranges = np.zeros((n1 * nt + nt, 2), dtype=np.int16)
    if key.size:
        starts = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
        ranges[key[starts], 0] = np.minimum.reduceat(kz, starts)
        ranges[key[starts], 1] = np.maximum.reduceat(kz, starts) + 1
    nonempty = (ranges[:n1 * nt, 1] > ranges[:n1 * nt, 0]).reshape(n1, nt)
    for t in range(nt):
        ys = np.flatnonzero(nonempty[:, t])
        if ys.size:
            ranges[n1 * nt + t] = (ys[0], ys[-1] + 1)
    parts = [ranges.reshape(-1), bits.reshape(-1).view(np.int16)]
    if bits_out is not None:
        parts.append(bits_out.reshape(-1).view(np.int16))
    return np.concatenate(parts)


def support_words(backend, oN):
    """(zw_in, zw_out) of the k-space support table for this grid on this backend -- the words per entry of its bitmaps, which
    follow from the kernel that transforms the z axis (oN[2]): 16 / 16 (256, 512), B / A (an A x B length), B / B (a chirp-z axis
    over an A x B length: 410 = 2 * 5 * 41 of the reference driver's default grid runs over 864 = 27 x 32) --, or None where the
    backend takes no table for the grid (the grid then runs without one: every grid row is written and read)"""
    f = getattr(backend, 'support_words', None)
    if f is None or int(oN[0]) % 16:
        return None
    return f(int(oN[2]))


def split_support(table, oN, tile=16, zw_in=16):
    """(z ranges (n1*nt, 2), y ranges (nt, 2), segment bits (n1*nt, zw_in) uint32: the input-side form) views of a support table"""
    n0, n1, n2 = (int(n) for n in oN)
    nt = n0 // tile
    table = np.ascontiguousarray(table, dtype=np.int16).reshape(-1)
    a, b = 2 * n1 * nt, 2 * (n1 * nt + nt)
    return table[:a].reshape(-1, 2), table[a:b].reshape(-1, 2), table[b:b + 2 * n1 * nt * zw_in].view(np.uint32).reshape(n1 * nt, zw_in)


def pow2_divisor(n, cap):
    """the largest power of two <= cap that divides n"""
    b = int(cap)
    while b > 1 and int(n) % b:
        b //= 2
    return max(b, 1)


# What one chunk of w interleaved coils costs per evaluation of the headline problem (ms, MI355X, DESIGN.md section 5: the per-rank
# timings of a coil-sharded run; w = 1 is the per-coil grid layout).  Only the ratios matter: they decide how a coil count
# that is no power of two is cut up -- 7 coils are cheaper as ONE 8-wide chunk with a zero-weight coil (6.6) than as 4 + 2 + 1
# (8.1), 12 coils as 8 + 4, 3 coils as a 4-wide chunk (3.9) rather than 2 + 1 (4.2).
CHUNK_COST = {16: 13.0, 8: 6.6, 4: 3.9, 2: 2.45, 1: 1.7}


def plan_chunks(Cn, chunk=8, single_ok=True, cost=None, pad=True):
    """Cut Cn coils into chunks the coil-interleaved kernels take: [(lo, hi, width)], width in {16, 8, 4, 2} (<= chunk) coils
    interleaved below the grid -- of which hi - lo are real and the rest, if any, zero-weight padding -- or width 1: one coil
    in the per-coil layout (only where the backend has those kernels for the grid: `single_ok`).  The cheapest cover by
    `cost` (default CHUNK_COST: measured on the headline problem on one MI355X; backend.tuning['chunk_cost'] overrides it for other
    grids / devices); every chunk of every coil count reaches the binned adjoint gridding and the fused transform leaf.
    pad=False (backend.tuning['chunk_pad']): no chunk may carry zero-weight padding coils -- a padded chunk allocates up to a third
    more grid and scratch than its real coils need (7 coils then run as 4 + 2 + 1 instead of one 8-wide chunk)."""
    cost = dict(CHUNK_COST, **(cost or {}))
    widths = [w for w in (16, 8, 4, 2) if w <= max(int(chunk), 2)] + ([1] if single_ok else [])
    best = [(0.0, [])]
    for c in range(1, int(Cn) + 1):
        cand = [(best[max(c - w, 0)][0] + cost[w], best[max(c - w, 0)][1] + [w]) for w in widths if pad or w <= c]
        if not cand:          # (no padding allowed and nothing fits: a single coil without the per-coil kernels -- pad after all)
            cand = [(best[max(c - w, 0)][0] + cost[w], best[max(c - w, 0)][1] + [w]) for w in widths]
        best.append(min(cand, key=lambda t: (round(t[0], 6), len(t[1]))))
    out, lo = [], 0
    for w in sorted(best[int(Cn)][1], reverse=True):
        hi = min(lo + w, int(Cn))
        out.append((lo, hi, w))
        lo = hi
    return out


def coil_chunks(Cn, chunk=8):
    """(lo, hi) of the chunks of plan_chunks"""
    return [(lo, hi) for lo, hi, _ in plan_chunks(Cn, chunk)]


def choose_layout(Cn, chunk=8, layout=None, single_ok=True, cost=None, pad=True):
    """(layout, chunks) for Cn coils on one rank: chunks = [(lo, hi, width)] (plan_chunks); layout 2 when any chunk is
    coil-interleaved.  An explicit per-coil layout (0 or 1) keeps all coils in one chunk."""
    if layout in (0, 1):
        return layout, [(0, int(Cn), 0)]                  # width 0: per-coil layout, any coil count
    chunks = plan_chunks(Cn, chunk, single_ok, cost, pad)
    return (2 if any(w > 1 for _, _, w in chunks) else 1), chunks


def pad_coils(w, width, interleaved=True):
    """weights box + (c,) -> box + (width,) with zero weights for the padding coils, in the memory form ZpadFFT keeps as it is
    (layout 2: a voxel's coils side by side)"""
    w = np.asarray(w)
    box, c = w.shape[:3], w.shape[3]
    if c == width:
        return w
    out = np.zeros(box[::-1] + (width,), dtype=_C64).transpose(2, 1, 0, 3) if interleaved else np.zeros(box + (width,), dtype=_C64, order='F')
    out[..., :c] = w
    return out


def assemble(backend, Gm, oN, N, weights_of, Cn, layout, chunks, table=None, box_lo=None, row_order=None,
             name='SENSE-fusedFFT', zw=(16, 16), sep=None, kshift=None):
    """A = KronI(C, G') * ZpadFFT, or a VStack of such trees over coil chunks sharing ONE device copy of G'.

    Gm          gridding matrix (T x P, complex64 CSR) with its columns in the order of `layout` (see
                permute_grid_columns; layout 2 uses layout 1's numbering)
    sep         the same matrix in separable form (interp_sep_records, grid_order 1) or None
    kshift      per-axis circular shifts the leaf's transform carries (HipBackend.fold_axis_shifts: Gm and sep were then built without the
                modulation of those axes) or None
    weights_of  (lo, hi) -> weights array box + (hi - lo,) for that run of coils (maps * roll-off * modulation)
    chunks      [(lo, hi, width)] from choose_layout: width coils interleaved, hi - lo of them real
    More coils than a chunk holds (the reference's `batch` hint, indigo/operators.py:15-17,341: evaluate a wide
    KronI a few columns at a time) become a VStack: the k-space rows come out coil-major exactly as from KronI(C, G'),
    and the adjoint accumulates the chunks' images (VStack, operators.py:440-447).  Chunks of different widths share the
    matrix too: it carries one binned adjoint format and one fine support table PER WIDTH present (8 and 4: brick rounds;
    2: slots), so 12 coils (8 + 4) or 6 (4 + 2) run the same kernels as 8, 4 and 2 coils do alone."""
    from indigo_amd import operators as op
    tuning = getattr(backend, 'tuning', {})          # format choices of the backend (HipBackend.tuning)
    chunks = [(c[0], c[1], (c[2] if len(c) > 2 else (c[1] - c[0] if layout == 2 else 0))) for c in chunks]
    il_widths = sorted({w for _, _, w in chunks if w > 1}, reverse=True)
    x16 = int(oN[0]) % 16 == 0

    # A finer k-space support table for the coil-interleaved trees (backends that take one): `tile` kx points per entry instead
    # of 16 -- 22 % instead of 30 % of the headline problem's grid is flagged at 8, 16 % at 4.  The gather routes keep the
    # 16-point table: whatever they write is a superset of what a reader with the finer table reads.
    # (measured on the headline problem: 7.80 ms at 16, 7.52 ms at 8; coils * tile >= 32 keeps the transform's 32-column tiles,
    # which need two tiles per table entry.  Round 4: 4 kx points per entry for 8-coil trees -- 16.4 % of the headline grid flagged
    # instead of 22.2 %; the scatter flushes the 4-point segments in pairs, one wave store as before: 6.92 -> 6.77 ms.)
    # One table per width present: a chunk's scatter writes by the table its own transform reads by.
    tile_of = tuning.get('support_tile', 8)
    fine = {}
    by_tile = {}
    for w in il_widths:
        tile = int(tile_of.get(w, 8) if isinstance(tile_of, dict) else tile_of)
        if (table is not None and tile in (4, 8) and w * tile >= 32 and getattr(backend, 'supports_support_tile', False)
                and (w in tuning.get('bricks', ()) or (sep is not None and w in tuning.get('shares', ())))):
            if tile not in by_tile:
                by_tile[tile] = grid_support(Gm, oN, tile, zw)
            fine[w] = (by_tile[tile], tile)

    def gridding(interleaved, ncols=0):
        # ncols: panel columns of the trees that use this matrix (per-coil layout: the chunk's coil count)
        G = backend.SpMatrix(Gm, name='interp*mod*scale')
        if interleaved:
            G._grid_interleaved = True
            if sep is not None and tuning.get('separable', True):
                # the same matrix as one record per sample (indigo_amd.interp.interp_sep_records): the interleaved products compute
                # their taps instead of streaming the stored ones
                G._grid_separable = sep
        if table is not None:
            G._grid_support = (table, int(oN[0]), int(oN[2]), int(zw[0]))
        if row_order is not None:
            G._row_order = row_order
            return G
        bricks, slots, fines, shares = {}, {}, {}, {}
        for w in (il_widths if interleaved else [ncols]):
            if (interleaved and sep is not None and tuning.get('separable', True) and w in tuning.get('shares', ()) and x16
                    and sep['tw'] >= tuning.get('shares_min_tw', 6)):
                # adjoint gridding as a scatter of (sample, brick) SHARES with the taps computed from the separable records
                # (ig_grid_scatter_sep): no stored taps at all.  A share costs the same whatever the number of taps inside it, so
                # this is the route of wide kernels -- the reference's default half-width 3 (125 taps per sample: 1.6 ms against
                # 2.2 ms for the stored-tap bricks on the headline trajectory); at half-width 2 (27 taps) the stored taps win
                # (0.62 against 0.88 ms), measured in profiles/r06_scatter_forms.txt
                sshape = tuning.get('share_shape', {})
                sshape = tuple(sshape.get(w, (8, 2, 1024, 1024)) if isinstance(sshape, dict) else sshape)
                # (share bricks need not divide the grid: 16 x 4 x 4 also on 640 x 277 x 410)
                shares[w] = (int(sshape[0]), int(sshape[1])) + sshape[2:]
                if w in fine:
                    fines[w] = fine[w]
            elif interleaved and w in tuning.get('bricks', ()) and x16:
                # adjoint gridding by grid bricks (a scatter binned on the host) instead of a gather over the transposed matrix.
                # Measured (config 4): 8 coils 0.91 ms against 1.82 ms (gather + its deferred long rows); 4 coils 0.68 against
                # 1.05 ms.  Two coils or one would pad every sample's share of a brick to 32 / 64 entries: they take the slots.
                shape = tuning.get('brick_shape', {})
                shape = shape.get(w, (2, 2, 4096, 4096)) if isinstance(shape, dict) else shape
                # bricks of 16 x bm x bs cells must tile the grid: halve a side until it divides (277 x 410, the reference driver's
                # default grid: 16 x 2 x 1)
                bricks[w] = (int(oN[0]), int(oN[2]), int(oN[1]), w, pow2_divisor(oN[2], shape[0]), pow2_divisor(oN[1], shape[1])) + tuple(shape[2:])
                if w in fine:
                    fines[w] = fine[w]
            elif w in tuning.get('slots', ()) and x16:
                # 1, 2 (or 4) columns: the same scatter with SLOTS in place of rounds (ig_ccsrmm_t_slots) -- no padding, no G'^T.
                sshape = tuple(tuning.get('slot_shape', (4, 4, 256, 64)))
                slots[w] = (int(oN[0]), int(oN[2]), int(oN[1]), w, pow2_divisor(oN[2], sshape[0]), pow2_divisor(oN[1], sshape[1])) + sshape[2:]
        if fines:
            G._grid_support_fine = fines
        if shares:
            G._grid_shares = shares
        if bricks:
            G._grid_bricks = bricks
        if slots:
            G._grid_slots = slots
        return G

    G_il = gridding(True) if il_widths else None
    kskw = {'kshift': tuple(int(v) for v in kshift)} if (kshift is not None and any(kshift)) else {}
    assert not kskw or all(w > 1 for _, _, w in chunks), "a transform that carries an axis shift exists for the coil-interleaved layout only"
    G_pc = None
    trees = []
    for lo, hi, w in chunks:
        real = hi - lo
        if w > 1:
            wts = pad_coils(weights_of(lo, hi), w)
            if w in fine:
                Z = backend.ZpadFFT(oN, N, wts, box_lo=box_lo, layout=2, support=fine[w][0], support_tile=fine[w][1], name='fft*zpad*apod*maps', **kskw)
            else:
                Z = backend.ZpadFFT(oN, N, wts, box_lo=box_lo, layout=2, support=table, name='fft*zpad*apod*maps', **kskw)
            tree = backend.KronI(w, G_il) * Z
            if real < w:
                # zero-weight padding coils: their k-space rows come last, are exact zeros on the way out and read as zeros on the way in
                tree = op.HeadRows(backend, tree, real * Gm.shape[0], name='coils %d..%d of a %d-wide chunk' % (lo, hi, w))
        else:
            lay = layout if layout in (0, 1) else 1          # one coil (or an explicit per-coil layout): the per-coil kernels
            G_pc = G_pc or gridding(False, real)
            Z = backend.ZpadFFT(oN, N, weights_of(lo, hi), box_lo=box_lo, layout=lay, support=table if lay == 1 else None, name='fft*zpad*apod*maps')
            tree = backend.KronI(real, G_pc) * Z
        trees.append(tree)
    A = trees[0] if len(trees) == 1 else backend.VStack(trees, name='coil-chunks')
    A._name = name
    A._support_fine = fine.get(il_widths[0]) if il_widths else None          # (the widest chunks' table: what most of the tree runs by)
    A._support_fine_by_width = fine
    A._support_zw = tuple(int(v) for v in zw)
    A._coil_chunks = chunks
    return A


def decode_zpad_maps(St, C, P, oN):
    """Recover (box_lo, box_dims, weights) from the stored form of the `-O3` tree's S' factor:
    St = S'^H as an N x (C*P) CSR whose row i holds, for each coil c, conj(w[i, c]) at column c*P + zrow(i), the
    zrow(i) enumerating a box of the grid in Fortran order (SenseProblem.fused_maps_T; the reference builds the same
    matrix from Zpad, FFTc's modulation, the roll-off and the maps, examples/pics.py:104-177).
    Entries may be MISSING: scipy's sparse products drop exact zeros, so coil maps that vanish outside the body (masked
    ESPIRiT maps) leave rows with fewer than C entries, or none -- a missing (voxel, coil) entry is a weight of 0.  The box
    is then recovered from the rows that are present: i = ix + d0*(iy + d1*iz) with (kx, ky, kz) = lo + (ix, iy, iz).
    Returns None if the matrix does not have that structure."""
    St = St.tocsr()
    if St.shape[1] != C * P or St.nnz == 0:
        return None
    rows = np.repeat(np.arange(St.shape[0], dtype=np.int64), np.diff(St.indptr))
    return decode_zpad_entries(rows, St.indices.astype(np.int64), St.data, St.shape[0], C, P, oN)


def decode_zpad_entries(rows, cols, data, Nn, C, P, oN):
    """decode_zpad_maps on the entries themselves, St[rows[j], cols[j]] = data[j] in any order (what the structured realisation of
    the recipe hands over, indigo_amd.structured.SelectS)"""
    rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
    Nn = int(Nn)
    if rows.size == 0 or cols.min() < 0 or cols.max() >= C * P or rows.min() < 0 or rows.max() >= Nn:
        return None
    coil, pos = cols // P, cols % P
    # one grid position per row (all its coils at the same point), at most one entry per (row, coil)
    zrow = np.full(Nn, -1, dtype=np.int64)
    zrow[rows] = pos
    if not np.array_equal(pos, zrow[rows]):
        return None
    if np.bincount(rows * C + coil, minlength=Nn * C).max() > 1:
        return None
    present = np.flatnonzero(zrow >= 0)
    n0, n1, n2 = (int(n) for n in oN)
    z = zrow[present]
    kx, ky, kz = z % n0, (z // n0) % n1, z // (n0 * n1)
    if rows.size == Nn * C:
        dims = (int(kx.max() - kx.min()) + 1, int(ky.max() - ky.min()) + 1, int(kz.max() - kz.min()) + 1)
    else:
        # present rows only: solve i = kx + d0*ky + d0*d1*kz + const for the box's pitches
        Amat = np.stack([ky, kz, np.ones_like(ky)], axis=1).astype(np.float64)
        if np.linalg.matrix_rank(Amat) < 3:
            return None
        sol = np.linalg.lstsq(Amat, (present - kx).astype(np.float64), rcond=None)[0]
        d0, d01 = int(round(sol[0])), int(round(sol[1]))
        if d0 < 1 or d01 < d0 or d01 % d0 or Nn % d01:
            return None
        dims = (d0, d01 // d0, Nn // d01)
    if dims[0] * dims[1] * dims[2] != Nn:
        return None
    ix, iy, iz = present % dims[0], (present // dims[0]) % dims[1], present // (dims[0] * dims[1])
    lo = (int(kx[0] - ix[0]), int(ky[0] - iy[0]), int(kz[0] - iz[0]))
    if not (np.array_equal(kx, lo[0] + ix) and np.array_equal(ky, lo[1] + iy) and np.array_equal(kz, lo[2] + iz)):
        return None
    if min(lo) < 0 or lo[0] + dims[0] > n0 or lo[1] + dims[1] > n1 or lo[2] + dims[2] > n2:
        return None
    w = np.zeros((Nn, C), dtype=_C64)
    w[rows, coil] = np.conj(data)
    return lo, dims, np.asfortranarray(w).reshape(dims + (C,), order='F')
