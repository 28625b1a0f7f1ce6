#!/usr/bin/env python3
"""Benchmarks of the SENSE hot path on MI355X (BASELINE.json metric and configs).

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4,5}]

Default (`--config 4`, BASELINE.json's metric): one "step" is one evaluation y = A^H A x of the composed
non-Cartesian SENSE normal operator (the reference's `-O3` tree S' -> FFT -> G' -> G'^H -> IFFT -> S'^H, with S' and
the FFT fused into the zero-pad-aware `ZpadFFT` leaf; `--tree o3` runs the reference's leaves one by one, `--tree
recipe` reaches the fused leaf through the reference's own recipe + `FuseZpadFFT`) on synthetic inputs already
resident in HBM: image 256^3, 8 coils, oversampled grid 512^3, 3-D radial trajectory with 1,851,904 samples,
width-4 (indigo width=2) Kaiser-Bessel gridding.  For N > 1 (one rank per GPU: launched by
torch.distributed.run, or -- a plain `python bench.py --gpus N` -- by this script itself, which then starts the N ranks as
child processes) the coils are sharded over the ranks and each evaluation ends in one RCCL all-reduce of the image: strong scaling.

Other configs (each prints its own JSON line, same contract fields):
    --config 1   examples/spmm.py: random 1e4 x 1e4 CSR (1 % nnz) x 8 RHS -- the reference's CPU-runnable case
    --config 2   batched 3-D C2C FFT, 256^3 x 16 (the plain fftn/ifftn contract)
    --config 3   3-D radial gridding CSR (5e7 nnz) x 64-column panel, forward and adjoint
    --config 5   SENSE 320^3 x 32 coils on the 512^3 grid (oversampling 1.6), coils sharded over the ranks, 8-coil
                 chunks on a rank (the reference's `batch` hint); `--shard R/W` times rank R's share of a W-rank
                 run on one GPU without communication (per-rank cost of a run that cannot be launched here)
The default run also reports config 5 for the same N in the `config5` object (`--no-config5` skips it), so the
driver's N = 1, 2, 4, 8 series carries the 32-coil problem's scaling next to the headline's.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline         : the dominant kernel: algorithmic (compulsory) bytes per launch / average launch duration measured
                     live with HIP events on the backend's stream; `traffic` = HBM bytes per launch from the
                     committed rocprofv3 PMC summary of this same command (profiles/)
  cpu_baseline     : the numpy oracle (restatement of the reference's numpy backend) timed on the host on ONE coil
                     of the same problem (warm-up + min of 5), scaled to evals/s; single-threaded, baseline only
  parity_rel_err   : the benchmarked operator with all coils but one switched off vs that oracle evaluation and vs a
                     double-precision evaluation of the same operator (the complex64 oracle is itself only good to
                     ~2.6e-5 on this DC-heavy input, oracle/precise.py)
  eval_traffic_*   : bytes the whole evaluation really moves (PMC summary) and the fraction of the 8 TB/s peak that
                     is; `reference_model_*`: the same evaluation priced with the reference's leaf-by-leaf model
                      -- a speed-up measure, NOT a roofline fraction (fusion removed those bytes)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_SUMMARIES = {4: os.path.join("profiles", "r06_cfg4_pmc_traffic.json"),
                 2: os.path.join("profiles", "r06_cfg2_pmc_traffic.json"),
                 3: os.path.join("profiles", "r06_cfg3_pmc_traffic.json"),
                 5: os.path.join("profiles", "r06_cfg5_pmc_traffic.json")}
PMC_NOTE = " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 per the gfx950 note)"


# profile-scope name (call site) -> device kernel symbol as rocprofv3 prints it.  The template arguments of the FFT
# kernel are <R1, R2, T, W, AXIS0, WMODE, BOXED, HALF> (indigo_amd/csrc/ig_fft.hip); which instantiation a pass runs
# depends on the grid layout, the axis length and whether the image box is the middle half of the axis.
def bricks_symbol(ncoils, support_tile, real_entries=False):
    """k_grid_bricks<NC, NSEG, PAIR, REALW>: segments per 16 x 2 x 2 brick unrolled for the 8-coil kernel (4 at 16 kx points per table
    entry, 8 at 8; the 4-point table's 16 segments are flushed in 8 pairs); REALW: 8-byte entries of a matrix with real weights"""
    rw = "true" if real_entries else "false"
    if ncoils != 8:
        return "k_grid_bricks<%d, 0, false, %s>" % (ncoils, rw)
    return {16: "k_grid_bricks<8, 4, false, %s>", 8: "k_grid_bricks<8, 8, false, %s>", 4: "k_grid_bricks<8, 8, true, %s>"}.get(support_tile, "k_grid_bricks<8, 0, false, %s>") % rw


def tree_real_entries(A):
    """does the tree's gridding matrix carry the 8-byte brick entries of a real-weight matrix?"""
    stack = [A]
    while stack:
        node = stack.pop()
        md = getattr(node, '_matrix_d', None)
        br, sl = getattr(md, '_bricks', None), getattr(md, '_slots', None)
        if br is not None:
            return br.get('words', 3) == 2
        if sl is not None:
            return sl.get('words', 4) == 3
        stack.extend(getattr(node, '_children', None) or [])
    return False


def kernel_symbols(layout, ncoils, half_box=True, n=512, support_tile=16, real_entries=False, tw=4):
    r1 = 32 if n == 512 else 16
    f = "k_fft_2stage<%d, 16, 16, %%s>" % r1
    m = {
        "fft_2stage_axis0": f % "16, true, 0, false, 0",
        "fft_2stage_axis1": f % "16, false, 0, false, 0",
        "fft_2stage_axis2": f % "16, false, 0, false, 0",
        "csrmm_gather": "k_csrmm_gather<8, 8, false, 0>",
    }
    lg = ncoils.bit_length() - 1
    ab = {160: "10, 16, 1", 192: "12, 16, 1", 240: "15, 16, 1", 320: "16, 20, 1", 384: "16, 24, 1", 400: "20, 20, 1", 432: "18, 24, 1", 480: "20, 24, 1", 640: "20, 32, 1"}
    if layout == 2 and n in ab:
        # the reference driver's own grids: zero-pad-aware passes on the A x B kernel <A, B, ROUNDS, WMODE> (ig_fft_ab.h)
        g = "anyfft::k_fft_ab_desc<%s, %%d>" % ab[n]
        m.update({"fft_pad_x": g % 1, "fft_pad_y": g % 0, "fft_pad_z": g % 0, "fft_crop_z": g % 0, "fft_crop_y": g % 0, "fft_crop_x": g % (3 + lg)})
        rw = "true" if real_entries else "false"
        gv = {8: "k_csrmm_gather_v<4, 2, 8, false, 0, %s>" % rw, 4: "k_csrmm_gather_v<2, 2, 8, false, 0, %s>" % rw,
              2: "k_csrmm_gather_v<2, 1, 8, false, 0, %s>" % rw}.get(ncoils, "k_csrmm_gather")
        m.update({"csrmm_rowlane_conj": "k_csrmm_dense64<%d, true, true>" % ncoils, "csrmm_gather": gv,
                  "csrmm_bricks_conj": bricks_symbol(ncoils, support_tile, real_entries)})
    elif layout == 2:
        rw = "true" if real_entries else "false"
        gv = {8: "k_csrmm_gather_v<4, 2, 8, false, 0, %s>" % rw, 4: "k_csrmm_gather_v<2, 2, 8, false, 0, %s>" % rw,
              2: "k_csrmm_gather_v<2, 1, 8, false, 0, %s>" % rw}.get(ncoils, "k_csrmm_gather")
        if half_box and n == 512:
            m.update({"fft_pad_x": f % "16, false, 1, true, 3",
                      "fft_pad_y": f % "32, false, 0, true, 1", "fft_pad_z": f % "32, false, 0, true, 1",
                      "fft_crop_z": f % "32, false, 0, true, 2", "fft_crop_y": f % "32, false, 0, true, 2",
                      "fft_crop_x": f % ("16, false, %d, true, 4" % (3 + lg))})
        else:
            # no compile-time half box (config 5: 320 of 512): run-time box predicates; the y and z passes take 32-column tiles
            # (the z passes where a support bitmap gates their loads / stores, i.e. in the fused tree)
            zt = "32" if n == 512 else "16"
            m.update({"fft_pad_x": f % "16, false, 1, true, 0",
                      "fft_pad_y": f % "32, false, 0, true, 0", "fft_pad_z": f % (zt + ", false, 0, true, 0"),
                      "fft_crop_z": f % (zt + ", false, 0, true, 0"), "fft_crop_y": f % "32, false, 0, true, 0",
                      "fft_crop_x": f % ("16, false, %d, true, 0" % (3 + lg))})
        m.update({"csrmm_rowlane_conj": "k_csrmm_dense64<%d, true, true>" % ncoils, "csrmm_gather": gv, "csrmm_slots_conj": "k_grid_slots<%d, %s>" % (ncoils, "true" if real_entries else "false"),
                  "csrmm_bricks_conj": bricks_symbol(ncoils, support_tile, real_entries)})
    if layout == 2:
        # round 6: the taps computed from the separable records (tw = weights per axis of a record: 4 up to kernel half-width 2, 6 up to 3)
        m.update({"grid_gather_sep": "k_grid_gather_sep<%d, %d, 0>" % (ncoils, tw), "grid_scatter_sep": "k_grid_scatter_mfma<%d, %d, 4>" % (ncoils, tw)})
    elif layout == 1:
        h = (3, 1, 1, 2, 2, 4) if half_box else (0,) * 6
        w = 32 if (half_box and n == 512) else 16         # compile-time half box: 32-column tiles (launch_2stage, ig_fft.hip)
        m.update({"fft_pad_x": f % ("16, true, 1, true, %d" % h[0]), "fft_pad_y": f % ("%d, false, 0, true, %d" % (w, h[1])),
                  "fft_pad_z": f % ("%d, false, 0, true, %d" % (w, h[2])), "fft_crop_z": f % ("%d, false, 0, true, %d" % (w, h[3])),
                  "fft_crop_y": f % ("%d, false, 0, true, %d" % (w, h[4])), "fft_crop_x": f % ("16, true, 2, true, %d" % h[5]),
                  "csrmm_rowlane_conj": "k_csrmm_dense64<%d, true, false>" % ncoils, "csrmm_slots_conj": "k_grid_slots<%d, %s>" % (ncoils, "true" if real_entries else "false")})
    return m


def load_pmc(cfg):
    path = os.path.join(ROOT, PMC_SUMMARIES.get(cfg, ""))
    if not os.path.isfile(path):
        return None, None
    return json.load(open(path)), PMC_SUMMARIES[cfg] + PMC_NOTE


def pmc_stale(pmc, symbols=()):
    """True when a committed PMC summary no longer describes the kernels that just ran: it was collected from other kernel
    sources (tools/pmc_summary.py records their hash), or a kernel symbol it is asked for is not in it"""
    if not pmc:
        return None
    from indigo_amd.build import source_hash
    meta = pmc.get("_meta") or {}
    return meta.get("csrc_sha16") != source_hash() or any(s not in pmc for s in symbols)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=4, choices=[1, 2, 3, 4, 5])
    ap.add_argument("--image", default="", help="image edge, or X,Y,Z for a non-cubic image (config 4: e.g. 480,208,308 --osf 1.3333 is the "
                    "reference driver's own scan on the 640 x 277 x 410 grid its sizing rule gives, backend.py:427-430); default 256 / 320 for config 4 / 5")
    ap.add_argument("--spokes-scale", type=float, default=1.0, help="config 4: multiply the number of radial spokes (8: a densely sampled "
                    "scan -- the k-space support table then flags most of the ball)")
    ap.add_argument("--no-dense", action="store_true", help="default run at N = 1: skip the extra dense-trajectory measurement")
    ap.add_argument("--coils", type=int, default=0, help="coils (default 8 / 32 for config 4 / 5)")
    ap.add_argument("--width", type=float, default=2.0, help="configs 4 / 5: half-width of the Kaiser-Bessel gridding kernel in grid cells (indigo's "
                    "`width`): 2 = BASELINE's 'width-4' kernel, 27 taps per sample; 3 = the default of the reference's Backend.NUFFT "
                    "(indigo/backends/backend.py:403; examples/pics.py:92 passes none): 125 taps per sample")
    ap.add_argument("--osf", type=lambda v: (float(v.split("/")[0]) / float(v.split("/")[1])) if "/" in v else float(v), default=0.0, help="config 4: oversampling factor of the gridding (default 2.0; 1.25 puts the 256^3 "
                    "image on the 320^3 grid the reference's own driver would pick, examples/pics.py:87-90)")
    ap.add_argument("--tree", choices=["zpadfft", "o3", "recipe"], default="zpadfft",
                    help="zpadfft: S' and the FFT fused into one zero-pad-aware leaf (default); o3: the reference's -O3 leaves; "
                         "recipe: the reference's factories + pics.py recipe + FuseZpadFFT (reaches the same fused leaf)")
    ap.add_argument("--layout", type=int, default=-1, help="grid layout of the fused tree: 1 = (x,z,y) per coil, 2 = coils interleaved "
                    "(default: 2 when this rank holds 2, 4, 8 or more coils, else 1)")
    ap.add_argument("--shard", default="", help="R/W: time rank R's coils of a W-rank run on this one GPU, no communication")
    ap.add_argument("--comm", choices=["auto", "rccl", "torch", "direct"], default="auto",
                    help="all-reduce provider for N > 1: the library's own RCCL binding (ig_comm_*), torch.distributed, or direct: the "
                         "library's own reduce-scatter + all-gather over HIP IPC windows (no RCCL; every rank talks to all peers at once)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity", action="store_true", help="configs 4 / 5 with --no-cpu-baseline: still check the benchmarked operator against the "
                    "float64 evaluation of a one-coil operator (oracle/precise.py)")
    ap.add_argument("--no-config5", action="store_true", help="default run: skip the extra config-5 measurement")
    ap.add_argument("--no-leaf-configs", action="store_true", help="default run at N = 1: skip the extra config-2 / config-3 measurements")
    ap.add_argument("--batch", type=int, default=16, help="config 2: number of volumes")
    ap.add_argument("--ncol", type=int, default=64, help="config 3: panel columns")
    ap.add_argument("--no-extras", action="store_true", help="only the selected config: --no-config5 --no-leaf-configs (profiling runs)")
    args = ap.parse_args()
    if args.no_extras:
        args.no_config5 = args.no_leaf_configs = args.no_dense = True
    dims = [int(v) for v in str(args.image).split(",") if v.strip()] if args.image else []
    assert len(dims) in (0, 1, 3), "--image N or --image X,Y,Z"
    args.image_dims = tuple(dims) if len(dims) == 3 else None          # non-cubic image (config 4 only)
    args.image = dims[0] if len(dims) == 1 else 0
    # the standard problem of the selected config: what the committed PMC summaries and the headline's name describe
    args.standard = not (args.image or args.image_dims or args.coils or args.osf or args.spokes_scale != 1.0 or args.width != 2.0) and args.tree == "zpadfft"
    return args


RANK = int(os.environ.get("RANK", "0"))
# fd 1 carries rank 0's ONE JSON line and nothing else -- under torchrun too, where all ranks share the launcher's stdout and
# libraries print there (gloo announces its connections on stdout): everything else this process or its libraries write to
# stdout goes to stderr (claim_stdout, called by main before any library is loaded)
REAL_STDOUT = sys.stdout


def claim_stdout():
    global REAL_STDOUT
    sys.stdout.flush()
    REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr


def log(*a):
    if RANK == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def timed_steps(B, comm, fn, steps, warmup):
    """W untimed + K timed calls of fn between barrier + stream sync on both sides; returns (seconds of the K steps -- max
    over ranks --, per-call-site profile of the timed region from stream events)"""
    for _ in range(warmup):
        fn()
    B.barrier()
    if comm:
        comm.barrier()
    B.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    B.barrier()
    if comm:
        comm.barrier()
    elapsed = time.perf_counter() - t0
    B.profile(False)
    prof = B.profile_report()
    if comm:
        elapsed = comm.max(elapsed)
    return elapsed, prof


def tree_support(A):
    """(support table, kx points per entry) of the fused leaf the tree evaluates with -- read off the tree, so that trees built by
    the reference's recipe + FuseZpadFFT are priced by the table THEY carry (None, 16 without one)"""
    from indigo_amd.operators import ZpadFFT
    stack = [A]
    while stack:
        node = stack.pop()
        if isinstance(node, ZpadFFT):
            if node._support_h is None:
                return None, 16
            return node._support_h, int(node._tile_kw.get('support_tile', 16))
        stack.extend(getattr(node, '_children', None) or [])
    return None, 16


def fused_leaves(A):
    """the ZpadFFT leaves of a fused tree in evaluation order (one per coil chunk): [dict(width, layout, table, tile)] -- read off the
    tree, so that trees built by the reference's recipe + FuseZpadFFT are priced by the tables THEY carry"""
    from indigo_amd.operators import ZpadFFT
    out = []

    def walk(node):
        if isinstance(node, ZpadFFT):
            out.append(dict(width=node._C, layout=node._layout, table=node._support_h,
                            tile=int(node._tile_kw.get('support_tile', 16)) if node._support_h is not None else 16))
        for c in (getattr(node, '_children', None) or []):
            walk(c)
    walk(A)
    return out


def check_fractions(obj, where="line", errors=None):
    """A fraction of the HBM peak above 1 is a pricing bug (bytes the kernel never moved) or a problem served from the caches, never
    a roofline result: the field is set to null and named in the returned list -- the rest of the line, the measured headline
    included, is still printed (the caller adds the list as `pricing_error` and exits non-zero after printing)."""
    errors = [] if errors is None else errors
    if isinstance(obj, dict):
        for k, v in list(obj.items()):
            if isinstance(v, (int, float)) and not isinstance(v, bool) and (k == "frac" or k.endswith("_frac") or k.endswith("frac_of_peak")) and v > 1.0:
                errors.append("%s.%s = %.3f > 1.0: the byte model prices traffic the kernels do not move (or the problem fits the caches)" % (where, k, v))
                obj[k] = None
            else:
                check_fractions(v, where + "." + str(k), errors)
    return errors


def roofline_of(prof, symbols, cfg, pick=None, traffic_ok=True):
    """merge call sites that launch the same device kernel, pick the dominant one, price it.  traffic_ok: the committed PMC summary
    of this config was collected on launches of THIS shape (8 coils per launch, the standard image and grid); otherwise `traffic`
    stays null -- a per-launch byte count of another shape is not this kernel's traffic"""
    kernels = {}
    for name, d in prof.items():
        sym = symbols.get(name, name)
        k = kernels.setdefault(sym, dict(launches=0, total_ms=0.0, bytes=0.0, ref_bytes=0.0, sites=[]))
        k['launches'] += d['launches']
        k['total_ms'] += d['total_ms']
        k['bytes'] += d['bytes']
        k['ref_bytes'] += d.get('ref_bytes', 0.0)
        k['sites'].append(name)
    if not kernels:
        return None, kernels
    dom = pick if pick in kernels else max(kernels, key=lambda k: kernels[k]['total_ms'])
    d = kernels[dom]
    avg_ms = d['total_ms'] / d['launches']
    per_launch = d['bytes'] / d['launches'] if d['bytes'] else None
    achieved = per_launch / (avg_ms * 1e-3) / 1e9 if per_launch else None
    pmc, src = load_pmc(cfg) if traffic_ok else (None, None)
    traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch") if pmc else None
    out = dict(bound="hbm", kernel=dom, call_sites=d['sites'], achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
               frac=(achieved / HBM_PEAK_GBS) if achieved else None, traffic=traffic,
               traffic_source=src if traffic else None, avg_launch_ms=avg_ms, launches=d['launches'],
               algorithmic_bytes_per_launch=per_launch,
               bytes_model="compulsory bytes of the pass (box, k-space support and coil sum taken into account), see DESIGN.md 3")
    if traffic:
        out["traffic_frac_of_peak"] = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out["traffic_stale"] = pmc_stale(pmc, [dom])
    if d['ref_bytes']:
        rb = d['ref_bytes'] / d['launches']
        out["reference_model"] = dict(bytes_per_launch=rb, equiv_GBps=rb / (avg_ms * 1e-3) / 1e9,
                                      note="SURVEY 8(d) accounting of the unfused leaf this pass replaces (4*x.nbytes per 3-D "
                                           "transform / 3 passes; csrmm nnz*12+(M+1)*4+X*read_frac+Y); not a roofline fraction")
    return out, kernels


def kernel_table(prof, steps):
    def row(v):
        r = {"launches_per_step": v['launches'] / steps, "avg_ms": round(v['avg_ms'], 4),
             "GBps": round(v['bytes'] / v['launches'] / (v['avg_ms'] * 1e-3) / 1e9, 1) if v['bytes'] else None,
             "frac_of_peak": round(v['bytes'] / v['launches'] / (v['avg_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) if v['bytes'] else None,
             "bytes_per_launch": v['bytes'] / v['launches'] if v['bytes'] else None}
        if v.get('ref_bytes'):
            r["reference_model_GBps"] = round(v['ref_bytes'] / v['launches'] / (v['avg_ms'] * 1e-3) / 1e9, 1)
        return r
    return {k: row(v) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['total_ms'])}


def host_info():
    return dict(host_cores=os.cpu_count(), note="single-threaded: pocketfft and scipy csr_matvecs do not thread; the reference's "
                                                 "get_max_threads() is 1 (indigo/backends/backend.py:243-244)")


# ---------------------------------------------------------------------------------------------------------
# communication
# ---------------------------------------------------------------------------------------------------------
def make_comm(args, B, world, rank, local_rank):
    """all-reduce provider for world > 1: the library's own RCCL communicator (no torch in the product path); torch's
    process group only if that cannot be brought up (or --comm torch)"""
    if world == 1:
        return None
    from indigo_amd import dist as igdist
    if args.comm == "direct":
        c = igdist.DirectComm(B, rank, world)
        log("communicator: %s" % c.describe())
        return c
    if args.comm in ("auto", "rccl") and os.environ.get("INDIGO_BENCH_DIST_BACKEND", "nccl") == "nccl":
        # (RcclComm raises on ALL ranks together or on none: what can fail on one rank alone -- loading RCCL -- is voted on in
        # the id handshake before anyone enters ncclCommInitRank, and the handshake itself times out everywhere at once; so
        # the fallback below is taken by every rank or by none)
        try:
            c = igdist.RcclComm(B, rank, world)
            log("communicator: ig_comm (RCCL through the C ABI), %d ranks" % world)
            return c
        except Exception as e:           # noqa: BLE001 -- any failure here falls back to the torch process group, loudly
            if args.comm == "rccl":
                raise
            print("[bench] rank %d: ig_comm bring-up FAILED (%s: %s); falling back to torch.distributed" % (rank, type(e).__name__, e),
                  file=sys.stderr, flush=True)
    os.environ["INDIGO_HIP_WITH_TORCH"] = "1"
    import torch
    import torch.distributed as dist
    dist_backend = os.environ.get("INDIGO_BENCH_DIST_BACKEND", "nccl")
    if dist_backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=dist_backend)     # gloo: several ranks may share one GPU (rehearsal only)
    log("communicator: torch.distributed (%s), %d ranks" % (dist_backend, world))
    return igdist.TorchComm(B)


# ---------------------------------------------------------------------------------------------------------
# configs 4 and 5: SENSE A^H A
# ---------------------------------------------------------------------------------------------------------
def sense_problem(cfg, img, C, osf=0.0, dims=None, spokes_scale=1.0, width=2.0):
    from indigo_amd.sense import SenseProblem
    if cfg == 4:
        osf = osf or 2.0
        N = tuple(dims) if dims else (img,) * 3
        nreadout = int(N[0] * osf)                           # samples per spoke = oversampled grid edge along the readout
        # 3617 spokes at 256^3 -> T = 1,851,904 at oversampling 2; scaled with the area of the (y, z) face for other images
        nspokes = int(round(3617 * (N[1] * N[2]) / 256.0 ** 2 * spokes_scale))
        return SenseProblem.synthetic(N, C, nspokes=nspokes, nreadout=nreadout, width=width, ntable=128,
                                      oversamp=osf, seed=4)
    # config 5: 320^3 in 512^3 (oversampling 1.6), maps generated per coil so that a rank only materialises its own
    grid = int(img * 1.6)
    nspokes = int(round(3617 * (img / 256.0) ** 2))          # 5652 spokes at 320^3 -> T = 2,893,824
    return SenseProblem.synthetic((img,) * 3, C, nspokes=nspokes, nreadout=grid, width=width, ntable=128,
                                  oversamp=1.6, seed=5, lazy_maps=True)


def run_sense(args, cfg, B, comm, world, rank, steps, warmup, want_cpu, quiet=False, want_parity=False):
    import numpy as np
    from indigo_amd.dist import ShardedNormalOperator, coil_range
    from indigo_amd.sense import normal_operator
    from indigo_amd.transforms import reserve_for
    from indigo_amd.util import Trace, rand64c

    t_setup = time.time()
    img = args.image or (256 if cfg == 4 else 320)
    C = args.coils or (8 if cfg == 4 else 32)
    dims = getattr(args, 'image_dims', None) if cfg == 4 else None
    scale = float(getattr(args, 'spokes_scale', 1.0)) if cfg == 4 else 1.0
    width = float(getattr(args, 'width', 2.0))
    width = int(width) if width == int(width) else width
    p = sense_problem(cfg, img, C, args.osf if cfg == 4 else 0.0, dims, scale, width)
    shard = None
    if args.shard:
        r, w = (int(v) for v in args.shard.split("/"))
        shard = (r, w)
        coils = list(coil_range(C, r, w))
    else:
        coils = list(coil_range(C, rank, world))
    log("config %d: image %s, %d coils (%d here%s), grid %s, T=%d (%.1fs)" % (
        cfg, "x".join(str(n) for n in p.N), C, len(coils), " = shard %d/%d" % shard if shard else "", p.oN, p.T, time.time() - t_setup))
    tree = args.tree if cfg == 4 else "zpadfft"
    fused_fft = tree in ("zpadfft", "recipe") and B.supports_padded_fft(p.oN, len(coils))
    layout = args.layout if args.layout >= 0 else None          # None: indigo_amd.fused.choose_layout cuts the coils into interleaved chunks
    if tree == "recipe":
        from indigo_amd.transforms import FuseZpadFFT, sense_recipe
        A = p.build_tree(B, level=0, coils=coils)
        for Step in sense_recipe(3) + [FuseZpadFFT]:
            A = Step().visit(A)
        layout = FuseZpadFFT.layout_of(A)
    elif fused_fft:
        A = p.build_zpadfft(B, coils=coils, layout=layout)
        layout = max([lf['layout'] for lf in fused_leaves(A)] or [0])
    else:
        A = p.build_fused(B, coils=coils)
        layout = 0
    nchunks = len(getattr(A, 'children', [])) if type(A).__name__ == "VStack" else 1
    log("tree:", {"zpadfft": "KronI(G') * ZpadFFT (S' folded into a zero-pad-aware FFT)", "recipe": "reference recipe + FuseZpadFFT",
                  "o3": "-O3: KronI(G') * (KronI(FFT) * S')"}[tree if fused_fft or tree == "o3" else "o3"],
        "layout %d, %d chunk(s)" % (layout, nchunks))
    c64 = np.dtype('complex64')
    Nvox = A.shape[1]
    x = B.copy_array(rand64c(Nvox, 1, seed=1))
    y = B.zero_array((Nvox, 1), c64)
    if comm is not None:
        reserve_for(A, 1)
        AHA = ShardedNormalOperator(A, comm)
    else:
        AHA = normal_operator(A)

    # one traced evaluation: algorithmic bytes by the reference's own model, and first touch of all buffers
    B.trace = Trace()
    AHA.eval(y, x)
    B.barrier()
    trace = B.trace
    B.trace = None
    ev = trace.by_event()
    ref_bytes_rank = trace.total_bytes()
    setup_s = time.time() - t_setup
    log("setup %.1fs; reference-model bytes/eval on this rank: %.2f GB %s" % (
        setup_s, ref_bytes_rank / 1e9, {k: round(v['nbytes'] / 1e9, 2) for k, v in ev.items()}))

    elapsed, prof = timed_steps(B, comm, lambda: AHA.eval(y, x), steps, warmup)
    ms_per_step = elapsed / steps * 1e3
    value = steps / elapsed

    # ---- per call site: compulsory bytes (ours) and reference-model bytes (the leaf it replaces).  A tree of several coil chunks
    # launches every call site once per chunk; chunks may differ in width (12 coils: 8 + 4) and then in support table too, so the
    # bytes are summed chunk by chunk, each priced by the table and tile ITS leaf carries.
    half_box = all(2 * b == n for b, n in zip(p.N, p.oN))
    leaves_z = fused_leaves(A) if fused_fft else []
    cpr = max([lf['width'] for lf in leaves_z] or [len(coils)])          # widest chunk: names the kernel symbols
    sup_tab, sup_tile = (leaves_z[0]['table'], leaves_z[0]['tile']) if leaves_z else (None, 16)
    p.last_support_zw = getattr(A, '_support_zw', None) or getattr(p, 'last_support_zw', (16, 16))     # (recipe trees carry their own)
    real_entries = tree_real_entries(A)
    recs = {True: [r['nbytes'] for r in trace.records if r['event'] == 'csrmm' and not r.get('fused') and r['name'] == 'interp*mod*scale' and r['forward']],
            False: [r['nbytes'] for r in trace.records if r['event'] == 'csrmm' and not r.get('fused') and r['name'] == 'interp*mod*scale' and not r['forward']]}
    acc = {}

    def add(site, nbytes, ref):
        if site in prof:
            a = acc.setdefault(site, [0.0, 0.0])
            a[0] += float(nbytes)
            a[1] += float(ref)
    if fused_fft:
        for ci, lf in enumerate(leaves_z):
            w = lf['width']
            for name, nbytes in p.zpadfft_pass_bytes(w, lf['table'], fused_sum=(lf['layout'] == 2), tile=lf['tile']).items():
                # 4 * x.nbytes per 3-D transform, a third per pass; x = grid x coils
                add(name, nbytes, 4.0 * np.prod(p.oN) * 8.0 * w / 3.0)
            gb = p.gridding_pass_bytes(w, lf['table'], tile=lf['tile'], real_entries=real_entries)
            # SpMM GB/s two ways: the reference's model (operators.py:246-256) and the bytes this kernel must move
            add("csrmm_gather", gb["csrmm_gather"], recs[True][ci] if ci < len(recs[True]) else 0.0)
            add("grid_gather_sep", gb["grid_gather_sep"], recs[True][ci] if ci < len(recs[True]) else 0.0)
            adj_ref = recs[False][ci] if ci < len(recs[False]) else 0.0
            for site in ("grid_scatter_sep",) + (("csrmm_bricks_conj",) if w >= 4 else ("csrmm_slots_conj",)) + ("csrmm_rowlane_conj", "csrmm_gather_conj"):
                if site in prof:
                    add(site, gb.get(site, gb["csrmm_rowlane_conj"]), adj_ref)
                    break
            add("pack_panel", gb["pack_panel"], 0.0)
    else:
        for fwd, sites in ((True, ("csrmm_gather",)), (False, ("csrmm_rowlane_conj", "csrmm_gather_conj", "csrmm_bricks_conj", "csrmm_slots_conj"))):
            for nb in recs[fwd]:
                for site in sites:
                    if site in prof:
                        add(site, nb, nb)
                        break
    for site, (nb, ref) in acc.items():          # (sums over the chunks of ONE evaluation; the profile covers `steps` of them)
        prof[site]['bytes'] = nb * steps
        if ref:
            prof[site]['ref_bytes'] = ref * steps
    tw = 4 if 2 * p.width <= 4 else 6 if 2 * p.width <= 6 else 8
    symbols = kernel_symbols(layout if fused_fft else 0, cpr, half_box, p.oN[0], sup_tile if fused_fft else 16, real_entries, tw)
    standard = args.standard or (cfg == 5 and not args.image and not args.coils)
    uniform8 = bool(leaves_z) and all(lf['width'] == 8 for lf in leaves_z)
    roofline, kernels = roofline_of(prof, symbols, cfg, traffic_ok=(standard and uniform8))
    if not quiet:
        for k in sorted(prof, key=lambda k: -prof[k]['total_ms']):
            log("  %-24s %4d launches  avg %8.3f ms  total %9.2f ms  %s" % (
                k, prof[k]['launches'], prof[k]['avg_ms'], prof[k]['total_ms'],
                "%6.0f GB/s" % (prof[k]['bytes'] / prof[k]['launches'] / prof[k]['avg_ms'] / 1e6) if prof[k]['bytes'] else ""))
    log("config %d: %.3f ms/step, %.2f evals/s on %d GPU(s)" % (cfg, ms_per_step, value, world))

    # ---- traffic of the whole evaluation: PMC summary where one exists for this exact configuration, else compulsory bytes
    comp_bytes = sum(v['bytes'] / steps for v in prof.values())
    pmc, pmc_src = load_pmc(cfg)
    traffic_bytes, traffic_src = None, None
    if pmc and world == 1 and not shard and standard:
        tot = 0.0
        for sym, k in kernels.items():
            if sym in pmc:
                tot += pmc[sym]["hbm_bytes_per_launch"] * k['launches'] / steps
        if tot:
            traffic_bytes, traffic_src = tot, pmc_src
    if traffic_bytes is None:
        traffic_bytes, traffic_src = comp_bytes, "sum of the kernels' compulsory bytes (no PMC summary for this configuration)"

    out = {
        "ms_per_step": ms_per_step, "value": value, "setup_s": round(setup_s, 2),
        "config": {"workload": "non-Cartesian SENSE A^H A, image %s, %d coils, grid %s (osf %.4g), radial T=%d, KB width %g; "
                               "-O3 tree S'->FFT->G'->G'^H->IFFT->S'^H%s" % (
                                   ("%d^3" % p.N[0]) if len(set(p.N)) == 1 else "x".join(str(n) for n in p.N), C,
                                   ("%d^3" % p.oN[0]) if len(set(p.oN)) == 1 else "x".join(str(n) for n in p.oN), p.oversamp, p.T, 2 * p.width,
                                   " (BASELINE config %d)" % cfg),
                   "parallelism": ("coil-sharded x%d (%d coils per rank), one all-reduce of the image per eval" % (world, len(coils))
                                   if world > 1 else ("rank %d of %d alone (no communication)" % shard if shard else "single GPU")),
                   "grid_layout": layout, "tree": tree, "coil_chunks_per_rank": nchunks,
                   "coil_chunk_widths": [lf['width'] for lf in leaves_z] or None, "spokes_scale": scale,
                   "support_table_kx_points_per_entry": (sup_tile if sup_tab is not None else None),
                   # large arrays are placed by probing: the best 1 GB window of an allocation 24 GB longer than the array (HipBackend.tuning[placement_window_gb / _allocs],
                   # DESIGN.md 3.1): bytes, the candidates' probe times in ms, the one kept -- part of set-up, outside the timed region
                   "placement": [list(e) for e in getattr(B, "_placement_log", [])][-4:] or None},
        "roofline": roofline,
        "eval_traffic_GB": traffic_bytes / 1e9,
        "eval_traffic_frac": traffic_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "eval_traffic_source": traffic_src,
        "eval_traffic_stale": pmc_stale(pmc, [k for k in kernels if k.startswith("k_")]) if (pmc and traffic_src == pmc_src) else None,
        "eval_compulsory_GB": comp_bytes / 1e9,
        "reference_model_GB_per_eval_per_gpu": ref_bytes_rank / 1e9,
        "reference_model_equiv": ref_bytes_rank / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "reference_model_note": "the reference's unfused leaf-by-leaf bytes (SURVEY 8d) / time / 8 TB/s: above 1 because the fusions "
                                "removed bytes -- a speed-up measure against the 0.6 target (41 evals/s), not a roofline fraction",
        "kernels": kernel_table(prof, steps),
    }
    if comm is not None:
        out["comm"] = comm.describe()
    if want_cpu:
        cpu, parity = cpu_baseline_and_parity(p, C, B, layout if fused_fft else None, y)
        out["cpu_baseline"] = cpu
        out["parity_rel_err"] = parity
    elif want_parity and fused_fft:
        out["parity_rel_err"] = parity_vs_float64(p, C, B, layout, y, rand64c(Nvox, 1, seed=1))
    if fused_fft and leaves_z and leaves_z[0]['table'] is not None:
        # what the k-space support tables flag: the fraction of the grid's rows every z pass and both gridding products touch
        lf = leaves_z[0]
        _, _, bits = p.split_support(lf['table'], lf['tile'])
        out["config"]["support_flagged_frac"] = float(np.unpackbits(bits.view(np.uint8)).sum()) * lf['tile'] / float(np.prod(p.oN))
    del AHA, A, x, y
    B._scratch = None
    p.drop_cache()
    return out


def cpu_baseline_and_parity(p, C, B, layout, y_dev):
    """numpy oracle on ONE coil of the same problem: warm-up + min of 5 evaluations; evals/s = 1 / (C * t_one_coil).
    The same oracle result checks the benchmarked operator: coils 1..C-1 switched off, identical kernels."""
    import numpy as np
    from indigo_amd.sense import SenseProblem, normal_operator
    from indigo_amd.util import rand64c
    from oracle.np_backend import NumpyBackend
    t0 = time.time()
    O = NumpyBackend()
    A1 = p.build_fused(O, coils=[0])
    AHA1 = normal_operator(A1)
    xh = rand64c(A1.shape[1], 1, seed=1)
    x = O.copy_array(xh)
    y = O.zero_array((A1.shape[1], 1), np.dtype('complex64'))
    times = []
    for i in range(6):                      # first = warm-up (scipy/pocketfft plan caches, page faults); ~40 s of CPU work in all
        t1 = time.perf_counter()
        AHA1.eval(y, x)
        times.append(time.perf_counter() - t1)
    t = min(times[1:])
    log("cpu baseline: one coil, warm-up %.2f s, then %s s (setup %.1f s)" % (times[0], ["%.2f" % v for v in times[1:]], time.time() - t0 - sum(times)))
    ref = y.to_host()
    cpu = dict(value=1.0 / (C * t), unit="evals/s", cores=1, kind="port", **host_info(),
               sample="numpy oracle (restatement of indigo/backends/np.py: np.fft.fftn + scipy csr @), 1 of %d coils of the same problem, "
                      "1 warm-up + min of 5 evaluations (%.1f s each), scaled linearly in coils" % (C, t))
    parity = parity_vs_float64(p, C, B, layout, y_dev, xh, ref) if layout is not None else None
    return cpu, parity


def parity_vs_float64(p, C, B, layout, y_dev, xh, ref_c64=None):
    """the benchmarked operator (same tree, same kernels) with every coil but the first switched off, against the
    double-precision evaluation of that one-coil operator (oracle/precise.py) -- and, where given, the complex64 oracle's result"""
    import numpy as np
    from indigo_amd.sense import SenseProblem, normal_operator
    zero = np.zeros(p.N, dtype=np.complex64, order='F')
    q = SenseProblem(p.N, p.coord, lambda c: p.coil_map(c) if c == 0 else zero, width=p.width, ntable=p.ntable,
                     oversamp=p.oversamp, ncoils=min(C, 8))
    q._interp_cache = p._interp_cache
    B._scratch = None
    A0 = q.build_zpadfft(B, layout=layout)
    AHA0 = normal_operator(A0)
    AHA0.eval(y_dev, B.copy_array(xh))
    got = y_dev.to_host()
    # the complex64 oracle's own error on this DC-heavy input is ~2.6e-5 (oracle/precise.py): the double-precision
    # evaluation of the same operator is the arbiter, the distance to the complex64 oracle is reported beside it
    from oracle.precise import CoilOperatorF64
    exact = CoilOperatorF64(p, 0).normal(xh).reshape(-1, 1)
    nrm = np.linalg.norm(exact)
    parity = dict(vs_float64_evaluation=float(np.linalg.norm(got - exact) / nrm), tolerance=1e-5)
    if ref_c64 is not None:
        parity.update(vs_complex64_oracle=float(np.linalg.norm(got - ref_c64) / np.linalg.norm(ref_c64)),
                      complex64_oracle_own_error=float(np.linalg.norm(ref_c64 - exact) / nrm))
    log("parity: benchmarked operator (coils 1.. switched off) vs float64 evaluation %.3e%s" % (
        parity["vs_float64_evaluation"], ("; vs the complex64 oracle %.3e (the oracle's own error: %.3e)" % (
            parity["vs_complex64_oracle"], parity["complex64_oracle_own_error"])) if ref_c64 is not None else ""))
    del A0, AHA0
    B._scratch = None
    return parity


def bench_sense(args, world, rank, local_rank):
    from indigo_amd.backends import get_backend
    B = get_backend("hip", device_id=local_rank)
    comm = make_comm(args, B, world, rank, local_rank)
    log("device:", B.device_name(), "world", world)
    cfg = args.config
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and not args.shard and cfg == 4
    res = run_sense(args, cfg, B, comm, world, rank, args.steps, args.warmup, want_cpu, want_parity=(args.parity and world == 1 and not args.shard))

    import threading
    emit_lock = threading.Lock()
    emitted = []

    def emit(extra5, leaves):
        """prints the ONE line (at most once: the main thread and the give-up timer may both get here); returns the pricing errors"""
        with emit_lock:
            if emitted:
                return []
            emitted.append(True)
            name = ("SENSE AHA evals/sec (256^3 x 8-coil non-Cartesian)" if cfg == 4 and args.standard
                    else "SENSE AHA evals/sec (%s)" % res["config"]["workload"].split(",", 1)[1].split(";")[0].strip())
            out = {"metric": name, "value": res["value"], "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                   "dtype": "complex64 (f32)", "data": "synthetic"}
            out.update({k: v for k, v in res.items() if k not in ("value", "ms_per_step")})
            out.setdefault("cpu_baseline", None)
            if extra5 is not None:
                out["config5"] = extra5
            out.update(leaves)
            errs = check_fractions({k: v for k, v in out.items() if k not in ("reference_model_equiv",)}, "line")
            if errs:
                out["pricing_error"] = errs
            print(json.dumps(out), file=REAL_STDOUT, flush=True)
            return errs

    extra5 = None
    if cfg == 4 and not args.no_config5 and not args.shard and args.standard:
        # Multi-rank runs: the extra is a second distributed job.  If ONE rank fails in it alone, the others wait for it inside a
        # collective -- the headline, already measured, must not go down with that: after a limit every rank gives up, rank 0
        # prints the line with the extra marked as failed, and the processes leave without the clean-up a hung collective would block.
        finished = threading.Event()
        limit = int(os.environ.get("INDIGO_BENCH_EXTRA_LIMIT", "300") or 0)

        def give_up():
            if finished.is_set():
                return
            try:
                print("[bench] rank %d: the config-5 extra did not finish within %d s" % (rank, limit), file=sys.stderr, flush=True)
                if rank == 0:
                    emit({"error": "the config-5 extra did not finish within %d s on every rank" % limit}, {})
                sys.stderr.flush()
            finally:
                # the headline line is out (rank 0); the run as a whole FAILED -- some rank hung in the extra's collective: every
                # rank leaves non-zero, the launcher keeps rank 0's line (self_launch relays it whatever the exit status)
                os._exit(3)
        timer = None
        if world > 1 and limit > 0:
            timer = threading.Timer(limit, give_up)
            timer.daemon = True
            timer.start()
        try:
            r5 = run_sense(args, 5, B, comm, world, rank, max(2, min(args.steps, 5)), min(args.warmup, 2), False, quiet=True)
            extra5 = {"evals_per_s": r5["value"], "ms_per_step": r5["ms_per_step"], "setup_s": r5["setup_s"], "n_gpus": world, "config": r5["config"],
                      "eval_traffic_frac": r5["eval_traffic_frac"], "kernels": r5["kernels"],
                      "note": "BASELINE config 5 (320^3 x 32 coils, grid 512^3) at the same N: strong scaling of the 32-coil problem"}
        except Exception as e:             # noqa: BLE001 -- the extra measurement must not cost the headline its line
            extra5 = {"error": "%s: %s" % (type(e).__name__, e)}
            print("[bench] config-5 extra failed on rank %d: %s" % (rank, extra5["error"]), file=sys.stderr, flush=True)
        finally:
            finished.set()
            if timer is not None:
                timer.cancel()
    leaves = {}
    if rank == 0 and world == 1 and cfg == 4 and not args.no_leaf_configs and not args.shard and args.standard:
        # the other half of BASELINE.json's metric ("SpMM HBM GB/s vs peak") and the plain FFT contract, in the same driver-run
        # line: BASELINE configs 2 and 3 with their own roofline / cpu_baseline / parity objects (`--config 2|3` alone prints
        # the same objects as full lines)
        import copy
        for c, fn in ((2, bench_fft), (3, bench_spmm)):
            a2 = copy.copy(args)
            a2.config, a2.image, a2.steps, a2.warmup = c, 0, max(5, min(args.steps, 10)), min(args.warmup, 3)
            try:
                B._scratch = None
                r = fn(a2, local_rank, B)
                leaves["config%d" % c] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "roofline",
                                                          "cpu_baseline", "parity_rel_err", "kernels")}
            except Exception as e:             # noqa: BLE001 -- an extra must not cost the headline its line
                leaves["config%d" % c] = {"error": "%s: %s" % (type(e).__name__, e)}
                print("[bench] config-%d extra failed: %s" % (c, leaves["config%d" % c]["error"]), file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and cfg == 4 and not args.no_dense and not args.shard and args.standard:
        # What a densely sampled scan costs: the same image, coils and grid with 8 x the spokes.  The headline's radial trajectory
        # touches 11 % of the grid's points and its support table flags 16 % of the rows -- every z pass and both gridding products
        # profit; here the table flags most of the k-space ball.  Same kernels, same pricing, parity against the float64 evaluation.
        import copy
        a2 = copy.copy(args)
        a2.spokes_scale, a2.standard = 8.0, False
        try:
            B._scratch = None
            r = run_sense(a2, 4, B, None, 1, 0, max(3, min(args.steps, 5)), min(args.warmup, 2), False, quiet=True, want_parity=True)
            leaves["dense_trajectory"] = {"evals_per_s": r["value"], "ms_per_step": r["ms_per_step"], "setup_s": r["setup_s"], "config": r["config"],
                                          "roofline": r["roofline"], "eval_compulsory_GB": r["eval_compulsory_GB"],
                                          "eval_compulsory_frac": r["eval_compulsory_GB"] / (r["ms_per_step"] * 1e-3) / HBM_PEAK_GBS,
                                          "parity_rel_err": r.get("parity_rel_err"), "kernels": r["kernels"],
                                          "note": "the headline problem with 8 x the radial spokes (python bench.py --spokes-scale 8): bounds what a well-sampled scan costs"}
        except Exception as e:             # noqa: BLE001 -- an extra must not cost the headline its line
            leaves["dense_trajectory"] = {"error": "%s: %s" % (type(e).__name__, e)}
            print("[bench] dense-trajectory extra failed: %s" % leaves["dense_trajectory"]["error"], file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and cfg == 4 and not args.no_dense and not args.shard and args.standard:
        # The reference's own default kernel: Backend.NUFFT(width=3) (indigo/backends/backend.py:403; examples/pics.py:92 passes no width)
        # -- 125 taps per sample instead of 27.  Same image, coils, grid and trajectory as the headline.
        import copy
        a3 = copy.copy(args)
        a3.width, a3.standard = 3.0, False
        try:
            B._scratch = None
            r = run_sense(a3, 4, B, None, 1, 0, max(3, min(args.steps, 5)), min(args.warmup, 2), False, quiet=True, want_parity=True)
            leaves["width3"] = {"evals_per_s": r["value"], "ms_per_step": r["ms_per_step"], "setup_s": r["setup_s"], "config": r["config"],
                                "eval_compulsory_GB": r["eval_compulsory_GB"],
                                "eval_compulsory_frac": r["eval_compulsory_GB"] / (r["ms_per_step"] * 1e-3) / HBM_PEAK_GBS,
                                "parity_rel_err": r.get("parity_rel_err"), "kernels": r["kernels"],
                                "note": "the headline problem with the reference's default kernel half-width 3 (python bench.py --width 3): 125 taps per sample"}
        except Exception as e:             # noqa: BLE001 -- an extra must not cost the headline its line
            leaves["width3"] = {"error": "%s: %s" % (type(e).__name__, e)}
            print("[bench] width-3 extra failed: %s" % leaves["width3"]["error"], file=sys.stderr, flush=True)
    errs = emit(extra5, leaves) if rank == 0 else []
    if comm is not None:
        comm.close()
    if errs:
        for e in errs:
            print("[bench] pricing error:", e, file=sys.stderr, flush=True)
        sys.exit(4)


# ---------------------------------------------------------------------------------------------------------
# config 2: batched 3-D C2C FFT, 256^3 x 16 (the reference contract Backend.fftn/ifftn, benchmark.py:36-62)
# ---------------------------------------------------------------------------------------------------------
def bench_fft(args, local_rank, B=None):
    import numpy as np
    from indigo_amd.backends import get_backend
    from indigo_amd.util import rand64c
    B = B or get_backend("hip", device_id=local_rank)
    n = args.image or 256
    batch = args.batch
    shape = (n, n, n, batch)
    c64 = np.dtype('complex64')
    x = B.empty_array(shape, c64)
    for j in range(batch):
        x[:, :, :, j:j + 1].copy_from(rand64c(n, n, n, 1, seed=2 + j))
    yv = B.zero_array(shape, c64)
    log("config 2: %s; plan: %s" % (shape, B.fft_describe(shape)))
    nbytes = x.nbytes
    elapsed, prof = timed_steps(B, None, lambda: B.fftn(yv, x), args.steps, args.warmup)
    ms = elapsed / args.steps * 1e3
    tot_kernel_ms = sum(v['total_ms'] for v in prof.values()) / args.steps
    pmc, src = load_pmc(2) if n == 256 and batch == 16 else (None, None)
    traffic = sum(pmc[k]["hbm_bytes_per_launch"] * 1.0 for k in pmc if k.startswith("k_fft")) if pmc else None
    inv_elapsed, _ = timed_steps(B, None, lambda: B.ifftn(yv, yv), max(2, args.steps // 2), 1)
    # parity on the spot: one volume (the fourth when there are that many) against numpy
    B.fftn(yv, x)
    pv = min(3, batch - 1)
    got = yv[:, :, :, pv:pv + 1].to_host()[..., 0]
    ref = np.fft.fftn(x[:, :, :, pv:pv + 1].to_host()[..., 0])
    perr = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    cpu = None
    if not args.no_cpu_baseline:
        v = x[:, :, :, 0:1].to_host()[..., 0]
        ts = []
        for _ in range(6):
            t1 = time.perf_counter()
            np.fft.fftn(v)
            ts.append(time.perf_counter() - t1)
        cpu = dict(value=1.0 / (batch * min(ts[1:])), unit="transforms/s", cores=1, kind="port", **host_info(),
                   sample="np.fft.fftn (the reference numpy backend's fftn, np.py:102-115) on 1 of %d volumes, warm-up + min of 5 (%.2f s), scaled" % (batch, min(ts[1:])))
    out = {"metric": "batched 3-D C2C FFT %d^3 x %d (complex64) transforms/sec" % (n, batch), "value": args.steps / elapsed,
           "unit": "transforms/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "complex64 (f32)", "data": "synthetic",
           "config": {"workload": "BASELINE config 2: forward fftn of %d^3 x %d, out of place (Backend.fftn contract)" % (n, batch),
                      "plan": B.fft_describe(shape), "inverse_in_place_ms": inv_elapsed / max(2, args.steps // 2) * 1e3},
           "roofline": dict(bound="hbm", kernel="whole transform (%d launches)" % sum(v['launches'] // args.steps for v in prof.values()),
                            achieved=4.0 * nbytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=4.0 * nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            algorithmic_bytes_per_launch=4.0 * nbytes,
                            bytes_model="SURVEY 8(d) / benchmark.py:55: 4 * x.nbytes per multi-dimensional transform",
                            traffic=traffic, traffic_source=src if traffic else None, traffic_stale=pmc_stale(pmc),
                            traffic_frac_of_peak=(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                            kernel_ms_per_transform=tot_kernel_ms),
           "cpu_baseline": cpu, "parity_rel_err": perr, "kernels": kernel_table(prof, args.steps)}
    del x, yv
    return out


# ---------------------------------------------------------------------------------------------------------
# config 3: 3-D radial gridding CSR (T x 256^3, 27 taps/row, ~5e7 nnz) x 64-column panel (benchmark.py:65-97)
# ---------------------------------------------------------------------------------------------------------
def bench_spmm(args, local_rank, B=None):
    import numpy as np
    import scipy.sparse as spp
    from scipy.signal.windows import kaiser
    from indigo_amd.backends import get_backend
    from indigo_amd.interp import interp_csr_arrays
    from indigo_amd.sense import radial_trajectory
    from indigo_amd.util import rand64c, Trace
    B = B or get_backend("hip", device_id=local_rank)
    n = args.image or 256
    ncol = args.ncol
    N = (n, n, n)
    coord = radial_trajectory(int(round(3617 * (n / 256.0) ** 2)), 2 * n, seed=3)
    T = int(np.prod(coord.shape[1:]))
    beta = np.pi * np.sqrt(((2 * 2.0 / 2.0) * (2.0 - 0.5)) ** 2 - 0.8)
    table = kaiser(2 * 128 + 1, beta)[128:]
    t0 = time.time()
    indptr, indices, w = interp_csr_arrays(T, N, 2, table, coord.reshape(3, -1, order='F'), dtype=np.float32)
    G = spp.csr_matrix((w.astype(np.complex64), indices, indptr), shape=(T, n ** 3))
    S = B.SpMatrix(G, name='gridding')
    c64 = np.dtype('complex64')
    P = n ** 3
    X = B.empty_array((P, ncol), c64)
    for j0 in range(0, ncol, 8):
        X[:, j0:j0 + min(8, ncol - j0)].copy_from(rand64c(P, min(8, ncol - j0), seed=100 + j0))
    Y = B.zero_array((T, ncol), c64)
    Z = B.zero_array((P, ncol), c64)
    B.trace = Trace()
    S.eval(Y, X)
    S.eval(Z, Y, forward=False)
    B.barrier()
    tr = B.trace
    B.trace = None
    fb = [r['nbytes'] for r in tr.records if r['forward']][0]
    ab = [r['nbytes'] for r in tr.records if not r['forward']][0]
    M = S._matrix_d
    log("config 3: G %d x %d, nnz %d, col_frac %.3f; reference-model bytes fwd %.2f GB, adj %.2f GB (setup %.1f s)" % (
        T, P, G.nnz, M._col_frac, fb / 1e9, ab / 1e9, time.time() - t0))
    elapsed, prof = timed_steps(B, None, lambda: S.eval(Y, X), args.steps, args.warmup)
    ms = elapsed / args.steps * 1e3
    a_elapsed, a_prof = timed_steps(B, None, lambda: S.eval(Z, Y, forward=False), args.steps, args.warmup)
    a_ms = a_elapsed / args.steps * 1e3
    pmc, src = load_pmc(3) if n == 256 and ncol == 64 else (None, None)
    traffic = traffic_note = None
    a_traffic = None
    ktab_f, ktab_a = kernel_table(prof, args.steps), kernel_table(a_prof, args.steps)
    if pmc:
        # The repacking kernel serves both directions under ONE symbol: its PMC figure is the mean over a forward and an adjoint
        # launch.  The adjoint's repack is dense (every byte of the panel read once and written once: its compulsory bytes ARE its
        # traffic), so the forward's share is 2 x mean - that.
        sym = {"csrmm_gather": "k_csrmm_gather", "csrmm_runs": "k_csrmm_runs64r", "csrmm_bricks_wide_conj": "k_bricks_wide64", "bricks_wide_zero": "k_wide_zero_unowned"}
        packs = [k for k in pmc if k.startswith("k_pack_panel_tiled")]
        a_pack = ktab_a.get("pack_panel", {}).get("bytes_per_launch")
        if packs and a_pack and "pack_panel" in ktab_f:
            ktab_f["pack_panel"]["pmc_bytes_per_launch"] = 2.0 * pmc[packs[0]]["hbm_bytes_per_launch"] - a_pack
            traffic_note = "PMC of the product kernel + (2 x the repack kernel's mean over a forward and an adjoint launch - the adjoint repack's dense bytes)"
        for tab in (ktab_f, ktab_a):
            for site, ent in tab.items():
                ks = [k for k in pmc if isinstance(pmc[k], dict) and sym.get(site) and k.startswith(sym[site])]
                if ks and ent.get("avg_ms"):
                    ent["pmc_bytes_per_launch"] = pmc[ks[0]]["hbm_bytes_per_launch"]
                    ent["pmc_GBps"] = round(pmc[ks[0]]["hbm_bytes_per_launch"] / (ent["avg_ms"] * 1e-3) / 1e9, 1)
        a_traffic = sum(e.get("pmc_bytes_per_launch", e.get("bytes_per_launch") or 0.0) * e["launches_per_step"] for e in ktab_a.values()) or None
        if all("pmc_bytes_per_launch" in e for e in ktab_f.values()):
            traffic = sum(e["pmc_bytes_per_launch"] * e["launches_per_step"] for e in ktab_f.values())
    # parity: column 5 against scipy, forward and adjoint
    x5 = X[:, 5:6].to_host()
    y5 = Y[:, 5:6].to_host()
    ref = G @ x5
    perr = float(np.linalg.norm(y5 - ref) / np.linalg.norm(ref))
    z5 = Z[:, 5:6].to_host()
    refa = G.conj().T.astype(np.complex128) @ y5.astype(np.complex128)
    perr_a = float(np.linalg.norm(z5 - refa) / np.linalg.norm(refa))
    cpu = None
    if not args.no_cpu_baseline:
        xs = np.asfortranarray(X[:, 0:8].to_host())
        ts = []
        for _ in range(6):
            t1 = time.perf_counter()
            G @ xs
            ts.append(time.perf_counter() - t1)
        cpu = dict(value=1.0 / (min(ts[1:]) * ncol / 8.0), unit="products/s", cores=1, kind="port", **host_info(),
                   sample="scipy csr @ dense (np.py:120-127) on 8 of the %d columns, warm-up + min of 5 (%.2f s), scaled" % (ncol, min(ts[1:])))
    # What this product can reach THROUGH the reference's boundary (column-major panels): the model prices every touched panel row once,
    # but a column-major panel is read in 128-byte lines of ONE column -- 16 rows -- and a gather of `ncol` columns a whole panel apart
    # needs the transposition: read the lines that hold a touched row, write the touched rows packed, read them packed, the matrix,
    # the result; all at the rate the memory system moves such a pattern without any arithmetic (tools/stride_probe.hip,
    # profiles/r05_stride_probe.txt: 5.4 - 5.6 TB/s).  The adjoint: the k-space panel packed (read, write, read), every row of the
    # result written once (beta = 0 defines it), the matrix with 8-byte entries.
    PROBE_GBS = 5500.0
    touched = np.zeros(P, dtype=bool)
    touched[G.indices] = True
    n_touched = int(touched.sum())
    n_lines16 = int(np.count_nonzero(touched.reshape(-1, 16).any(axis=1))) * 16 if P % 16 == 0 else P
    e = 8.0 * ncol
    floor_fwd = n_lines16 * e + 2.0 * n_touched * e + (G.nnz * 12.0 + (T + 1) * 4.0) + T * e
    floor_adj = 3.0 * T * e + P * e + G.nnz * 8.0
    floor = dict(bytes=floor_fwd, ms=floor_fwd / PROBE_GBS / 1e6, frac_by_model=fb / (floor_fwd / PROBE_GBS / 1e6 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 rate_GBps=PROBE_GBS,
                 derivation="lines of 16 panel rows that hold a touched row x %d columns (%.2f GB) + the touched rows written and read packed "
                            "(2 x %.2f GB) + matrix (%.2f GB) + result (%.2f GB), at the stride probe's %.1f TB/s" % (
                                ncol, n_lines16 * e / 1e9, n_touched * e / 1e9, (G.nnz * 12.0 + (T + 1) * 4.0) / 1e9, T * e / 1e9, PROBE_GBS / 1e3),
                 adjoint=dict(bytes=floor_adj, ms=floor_adj / PROBE_GBS / 1e6, frac_by_model=ab / (floor_adj / PROBE_GBS / 1e6 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              derivation="k-space panel read, written packed, read packed (3 x %.2f GB) + every row of the result written once "
                                         "(%.2f GB) + 8-byte entries (%.2f GB)" % (T * e / 1e9, P * e / 1e9, G.nnz * 8.0 / 1e9)))
    out = {"metric": "gridding CSR (%d x %d^3, nnz %.2e) x %d-column SpMM products/sec" % (T, n, G.nnz, ncol), "value": args.steps / elapsed,
           "unit": "products/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "complex64 (f32)", "data": "synthetic",
           "config": {"workload": "BASELINE config 3: forward csrmm through Backend.ccsrmm (column-major panels), adjoint reported beside it",
                      "col_frac": M._col_frac, "adjoint_ms": a_ms,
                      "adjoint_GBps_reference_model": ab / (a_ms * 1e-3) / 1e9,
                      "adjoint_frac_of_peak_reference_model": ab / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "adjoint_traffic": a_traffic,
                      "adjoint_traffic_frac_of_peak": (a_traffic / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if a_traffic else None,
                      "adjoint_parity_rel_err_vs_float64": perr_a},
           "roofline": dict(bound="hbm", kernel="forward product (%s)" % " + ".join(sorted(prof)), achieved=fb / (ms * 1e-3) / 1e9,
                            peak=HBM_PEAK_GBS, unit="GB/s", frac=fb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            algorithmic_bytes_per_launch=fb,
                            bytes_model="SURVEY 8(d) / operators.py:246-256: nnz*12 + (M+1)*4 + K*n*8*col_frac + M*n*8",
                            traffic=traffic, traffic_source=src if traffic else None, traffic_note=traffic_note if traffic else None,
                            traffic_stale=pmc_stale(pmc),
                            traffic_frac_of_peak=(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                            floor_ms=floor["ms"], floor=floor),
           "cpu_baseline": cpu, "parity_rel_err": perr,
           "kernels": {"forward": ktab_f, "adjoint": ktab_a}}
    del X, Y, Z, S
    return out


# ---------------------------------------------------------------------------------------------------------
# config 1: examples/spmm.py on the numpy backend (CPU only in the reference); HIP timed beside the oracle
# ---------------------------------------------------------------------------------------------------------
def bench_spmm_example(args, local_rank):
    import numpy as np
    import scipy.sparse as spp
    from indigo_amd.backends import get_backend
    from indigo_amd.util import rand64c
    from oracle.np_backend import NumpyBackend
    B = get_backend("hip", device_id=local_rank)
    A = spp.random(10000, 10000, density=0.01, format='csr', random_state=np.random.default_rng(1), dtype=np.float32).astype(np.complex64)
    xh = rand64c(10000, 8, seed=1)
    S = B.SpMatrix(A)
    x = B.copy_array(xh)
    y = B.zero_array((10000, 8), np.dtype('complex64'))
    S.eval(y, x)
    elapsed, prof = timed_steps(B, None, lambda: S.eval(y, x), max(args.steps, 50), args.warmup)
    steps = max(args.steps, 50)
    ms = elapsed / steps * 1e3
    O = NumpyBackend()
    So = O.SpMatrix(A)
    xo, yo = O.copy_array(xh), O.zero_array((10000, 8), np.dtype('complex64'))
    ts = []
    for _ in range(11):
        t1 = time.perf_counter()
        So.eval(yo, xo)
        ts.append(time.perf_counter() - t1)
    perr = float(np.linalg.norm(y.to_host() - yo.to_host()) / np.linalg.norm(yo.to_host()))
    nbytes = A.nnz * 12 + 10001 * 4 + 10000 * 8 * 8 * 2
    out = {"metric": "examples/spmm.py: 1e4 x 1e4 CSR (1 % nnz) x 8 RHS SpMM products/sec", "value": steps / elapsed, "unit": "products/s",
           "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "complex64 (f32)", "data": "synthetic",
           "config": {"workload": "BASELINE config 1 (the reference's CPU-runnable case), forward SpMatrix.eval, nnz %d" % A.nnz},
           "roofline": dict(bound="hbm", kernel="+".join(sorted(prof)), achieved=nbytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, algorithmic_bytes_per_launch=nbytes, traffic=None,
                            note="13 MB: launch-latency bound, fits the L2"),
           "cpu_baseline": dict(value=1.0 / min(ts[1:]), unit="products/s", cores=1, kind="port", **host_info(),
                                sample="numpy oracle SpMatrix.eval, 1 warm-up + min of 10 (%.2f ms)" % (min(ts[1:]) * 1e3)),
           "parity_rel_err": perr, "kernels": kernel_table(prof, steps)}
    return out


def visible_gpu_count():
    """GPUs of this node WITHOUT initialising the HIP runtime in this process (the launcher must stay clear of the GPU: it
    starts the ranks as child processes): KFD topology nodes with compute units, cut down by a *_VISIBLE_DEVICES list."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([v for v in os.environ[var].split(",") if v.strip()]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` started without a launcher: this process becomes the launcher.  It never touches the GPU;
    it starts N fresh ranks of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would
    set them, the RCCL id file in a private directory), relays rank 0's JSON line, and fails if any rank does."""
    import shutil
    import socket
    import subprocess
    import tempfile
    n = args.gpus
    rehearsal = os.environ.get("INDIGO_BENCH_DIST_BACKEND", "nccl") != "nccl"       # gloo: all ranks on GPU 0
    have = visible_gpu_count()
    if not rehearsal and have is not None and have < n:
        print("[bench] --gpus %d but this node shows %d GPU(s): one rank per GPU" % (n, have), file=sys.stderr, flush=True)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    rdv = tempfile.mkdtemp(prefix="indigo_bench_")
    procs = []
    rc = 0
    kill_at = 0.0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), INDIGO_COMM_ID_FILE=os.path.join(rdv, "rccl_id"),
                       INDIGO_BENCH_LAUNCHER="self")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
        print("[bench] launcher: started %d ranks (pids %s)" % (n, [p.pid for p in procs]), file=sys.stderr, flush=True)
        out0 = None
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                if r == 0 and out0 is None:
                    # (rank 0 prints one line at the very end: reading its pipe to EOF is the wait)
                    try:
                        out0, _ = procs[0].communicate(timeout=0.5)
                    except subprocess.TimeoutExpired:
                        pass
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print("[bench] launcher: rank %d exited with %d; stopping the others" % (r, code), file=sys.stderr, flush=True)
                    for q in pending:
                        procs[q].terminate()
                    kill_at = time.time() + 10.0          # a rank inside ncclCommInitRank / a collective may ignore SIGTERM
            if rc and pending and time.time() > kill_at:
                for q in pending:
                    if procs[q].poll() is None:
                        procs[q].kill()
            time.sleep(0.2)
        if out0:
            # rank 0's stdout is its ONE JSON line -- plus whatever a library chose to print there (gloo announces its
            # connections on stdout): only the JSON goes on, the rest to stderr
            for line in out0.decode().splitlines():
                if line.lstrip().startswith("{"):
                    sys.stdout.write(line + "\n")
                else:
                    print(line, file=sys.stderr)
            sys.stdout.flush()
            sys.stderr.flush()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(rdv, ignore_errors=True)
    sys.exit(rc if rc else 0)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.shard:
        assert args.config in (4, 5), "configs 1-3 are single-GPU leaf benchmarks"
        self_launch(args)
    claim_stdout()
    rank = RANK
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Watchdog: after that many seconds every thread's stack is dumped and the rank exits non-zero (the launcher then stops the
    # others).  ON by default for multi-rank runs (900 s; INDIGO_BENCH_WATCHDOG=0 disables): a bring-up or a collective that
    # hangs on one rank must end with stacks and an exit status, not eat the caller's whole time limit in silence.
    wd = int(os.environ.get("INDIGO_BENCH_WATCHDOG", "900" if world > 1 else "0") or 0)
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True, file=sys.stderr)
    # (--shard R/W times one rank's share in this ONE process, whatever --gpus says the full run would use)
    assert world == args.gpus or (args.shard and world == 1), "--gpus %d but WORLD_SIZE is %d" % (args.gpus, world)
    if world > 1 and args.comm in ("auto", "torch"):
        # The collective is the library's own RCCL binding (ig_comm_*).  Under `--comm auto` torch is imported FIRST --
        # before libindigo_hip.so loads -- only so that the fallback to torch.distributed stays possible: torch ships its own
        # HIP runtime and RCCL, and whichever copy is loaded first must serve both (the library then reuses torch's copies).
        # `--comm rccl` runs without torch in the process (tests/test_hip_dist.py proves that path).
        os.environ["INDIGO_HIP_WITH_TORCH"] = "1"
        import torch                                     # noqa: F401
    if world > 1 and os.environ.get("INDIGO_BENCH_DIST_BACKEND", "nccl") == "nccl":
        from indigo_amd import _lib
        import ctypes
        n = ctypes.c_int()
        _lib.lib().ig_device_count(ctypes.byref(n))
        assert local_rank < n.value, "LOCAL_RANK %d but only %d GPU(s) visible: one rank per GPU" % (local_rank, n.value)
    elif world > 1:
        local_rank = 0                      # gloo rehearsal: all ranks share GPU 0
    if args.config in (4, 5):
        bench_sense(args, world, rank, local_rank)
    else:
        assert world == 1, "configs 1-3 are single-GPU leaf benchmarks"
        out = {1: bench_spmm_example, 2: bench_fft, 3: bench_spmm}[args.config](args, local_rank)
        errs = check_fractions(out, "line")
        if errs:
            out["pricing_error"] = errs
        print(json.dumps(out), file=REAL_STDOUT, flush=True)
        if errs:
            sys.exit(4)


if __name__ == "__main__":
    main()
