#!/usr/bin/env python3
"""Headline benchmark: SENSE A^H A evaluations per second (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" is one evaluation y = A^H A x of the composed non-Cartesian SENSE
normal operator (the reference's `-O3` tree: S' -> FFT -> G' -> G'^H -> IFFT -> S'^H;
by default with S' and the FFT fused into the zero-pad-aware `ZpadFFT` leaf, `--tree o3`
runs the reference's leaves one by one) on synthetic inputs already resident in HBM: image 256^3, 8 coils, oversampled
grid 512^3, 3-D radial trajectory with 1,851,904 samples, width-4 (indigo
width=2) Kaiser-Bessel gridding (BASELINE config 4).  For N > 1 (launched by
torch.distributed.run, one rank per GPU) the 8 coils are sharded over the ranks
and each evaluation ends in one RCCL all-reduce of the image: strong scaling.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel's algorithmic bytes per launch / its average launch duration,
                 measured live with HIP events on the backend's stream during the timed steps
  cpu_baseline : the numpy oracle (restatement of the reference's numpy backend) timed on the host
                 on ONE coil of the same problem, scaled to evals/s (single-threaded, baseline only)
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


# profile-scope name (call site) -> device kernel symbol as rocprofv3 prints it (512-point axes).  The template
# arguments are <R1, R2, T, W, AXIS0, WMODE, BOXED, HALF> (indigo_amd/csrc/ig_fft.hip); which instantiation a pass
# runs depends on the grid layout (bench default: 2 = coils interleaved when the rank holds 2, 4 or 8 coils).
def kernel_symbols(layout, ncoils):
    f = "k_fft_2stage<32, 16, 16, %s>"
    m = {
        "fft_2stage_axis0": f % "16, true, 0, false, 0",
        "fft_2stage_axis1": f % "16, false, 0, false, 0",
        "fft_2stage_axis2": f % "16, false, 0, false, 0",
        "csrmm_gather": "k_csrmm_gather<8, 8, false, 0>",
    }
    if layout == 2:
        # Strided passes over the combined (coil, kx) index.  HALF 1: half input box + run-time output support (pad y/z),
        # HALF 2: run-time input support + half output box (crop z/y); 32-column tiles (W = 32) for the half-input
        # variants and for the y pass at its 16 MB stride (launch_2stage in ig_fft.hip); the last pass sums the coils
        # (WMODE = 3 + log2(coils)).
        m.update({"fft_pad_x": f % "16, false, 1, true, 3",
                  "fft_pad_y": f % "32, false, 0, true, 1", "fft_pad_z": f % "32, false, 0, true, 1",
                  "fft_crop_z": f % "16, false, 0, true, 2", "fft_crop_y": f % "32, false, 0, true, 2",
                  "fft_crop_x": f % ("16, false, %d, true, 4" % (3 + ncoils.bit_length() - 1)),
                  "csrmm_rowlane_conj": "k_csrmm_dense64<%d, true, true>" % ncoils,
                  "csrmm_gather": {8: "k_csrmm_gather_v<4, 2, 8, false, 0>", 4: "k_csrmm_gather_v<2, 2, 8, false, 0>",
                                   2: "k_csrmm_gather_v<2, 1, 8, false, 0>"}.get(ncoils, "k_csrmm_gather")})
    else:
        m.update({"fft_pad_x": f % "16, true, 1, true, 3", "fft_crop_x": f % "16, true, 2, true, 4",
                  "fft_pad_y": f % "16, false, 0, true, 1", "fft_pad_z": f % "16, false, 0, true, 1",
                  "fft_crop_y": f % "16, false, 0, true, 2", "fft_crop_z": f % "16, false, 0, true, 2",
                  "csrmm_rowlane_conj": "k_csrmm_dense64<%d, true, false>" % ncoils})
    return m


PMC_SUMMARY = os.path.join("profiles", "r01i_pmc_traffic.json")


def pmc_traffic(kernel, grid, ncoils, image):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/), valid only
    for the configuration it was taken on (image 256^3, grid 512^3, 8 coils on one GPU); otherwise None.
    PMC counters cannot be read from inside the benchmark process, so this figure comes from the
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same script."""
    path = os.path.join(ROOT, PMC_SUMMARY)
    if not os.path.exists(path) or tuple(grid) != (512, 512, 512) or ncoils != 8 or image != 256:
        return None, None
    d = json.load(open(path)).get(kernel)
    if not d:
        return None, None
    return d["hbm_bytes_per_launch"], PMC_SUMMARY + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 per the gfx950 note)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--image", type=int, default=256, help="image edge (debug: smaller problems)")
    ap.add_argument("--coils", type=int, default=8)
    ap.add_argument("--tree", choices=["zpadfft", "o3"], default="zpadfft",
                    help="zpadfft: S' and the FFT fused into one zero-pad-aware leaf (default); o3: the reference's -O3 leaves")
    ap.add_argument("--layout", type=int, default=-1, help="grid layout of the fused tree: 1 = (x,z,y) per coil, 2 = coils interleaved "
                    "(default: 2 when this rank holds 2, 4 or 8 coils, else 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-budget", type=float, default=60.0, help="skip the CPU leg if setup says it will exceed this many seconds")
    return ap.parse_args()


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "--gpus must match WORLD_SIZE"
    comm = None
    if world > 1:
        os.environ["INDIGO_HIP_WITH_TORCH"] = "1"
        import torch
        import torch.distributed as dist
        # one rank per GPU over RCCL; INDIGO_BENCH_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal only)
        dist_backend = os.environ.get("INDIGO_BENCH_DIST_BACKEND", "nccl")
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=dist_backend)

    import numpy as np
    from indigo_amd.backends import get_backend
    from indigo_amd.dist import ShardedNormalOperator, TorchComm, coil_range
    from indigo_amd.sense import SenseProblem, normal_operator
    from indigo_amd.util import Trace, rand64c

    t_setup = time.time()
    B = get_backend("hip", device_id=local_rank)
    if world > 1:
        comm = TorchComm(B)
    log(rank, "device:", B.device_name(), "world", world)

    img, C = args.image, args.coils
    nreadout = 2 * img                                   # samples per spoke = oversampled grid edge
    nspokes = int(round(3617 * (img / 256.0) ** 2))      # 3617 spokes at 256^3 -> T = 1,851,904
    p = SenseProblem.synthetic((img,) * 3, C, nspokes=nspokes, nreadout=nreadout, width=2, ntable=128,
                               oversamp=2.0, seed=4)
    coils = list(coil_range(C, rank, world))
    log(rank, "problem: image %d^3, %d coils (%d on this rank), grid %s, T=%d (%.1fs)" % (img, C, len(coils), p.oN, p.T, time.time() - t_setup))
    fused_fft = args.tree == "zpadfft" and B.supports_padded_fft(p.oN)
    layout = args.layout if args.layout >= 0 else (2 if len(coils) in (2, 4, 8) else 1)
    A = p.build_zpadfft(B, coils=coils, layout=layout) if fused_fft else p.build_fused(B, coils=coils)
    log(rank, "tree:", "KronI(G') * ZpadFFT (S' folded into a zero-pad-aware FFT)" if fused_fft
        else "-O3: KronI(G') * (KronI(FFT) * S')")
    c64 = np.dtype('complex64')
    Nvox = A.shape[1]
    x = B.copy_array(rand64c(Nvox, 1, seed=1))
    y = B.zero_array((Nvox, 1), c64)
    if world > 1:
        # size the arena for A and A^H on this rank's coils
        from indigo_amd.transforms import reserve_for
        reserve_for(A, 1)
        AHA = ShardedNormalOperator(A, comm)
    else:
        AHA = normal_operator(A)

    # one traced evaluation: algorithmic bytes by the reference's own model, and first-touch of all buffers
    B.trace = Trace()
    AHA.eval(y, x)
    B.barrier()
    ev = B.trace.by_event()
    alg_bytes_rank = B.trace.total_bytes()
    B.trace = None
    log(rank, "setup %.1fs; algorithmic bytes/eval on this rank: %.2f GB %s" % (
        time.time() - t_setup, alg_bytes_rank / 1e9, {k: round(v['nbytes'] / 1e9, 2) for k, v in ev.items()}))
    log(rank, "fft plan:", B.fft_describe(p.oN + (len(coils),)))

    for _ in range(args.warmup):
        AHA.eval(y, x)
    B.barrier()
    if comm:
        comm.barrier()
    B.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        AHA.eval(y, x)
    B.barrier()
    if comm:
        comm.barrier()
    t1 = time.perf_counter()
    B.profile(False)
    prof = B.profile_report()
    elapsed = t1 - t0
    if comm:
        elapsed = comm.max(elapsed)
    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed

    # dominant kernel and its roofline point.  The event brackets are per call site; call sites that launch
    # the same device kernel (the two strided FFT axes) are merged so the figure matches rocprofv3's row.
    if fused_fft:
        # price the fused passes with their exact compulsory bytes (box and k-space support taken into account)
        exact = p.zpadfft_pass_bytes(len(coils), getattr(p, 'last_support_table', None), fused_sum=(layout == 2))
        for name, nbytes in exact.items():
            if name in prof:
                prof[name]['bytes'] = float(nbytes) * prof[name]['launches']
    kernels = {}
    KERNEL_SYMBOL = kernel_symbols(layout if fused_fft else 0, len(coils))
    for name, d in prof.items():
        sym = KERNEL_SYMBOL.get(name, name)
        k = kernels.setdefault(sym, dict(launches=0, total_ms=0.0, bytes=0.0))
        k['launches'] += d['launches']
        k['total_ms'] += d['total_ms']
        k['bytes'] += d['bytes']
    dom = max(kernels, key=lambda k: kernels[k]['total_ms']) if kernels else None
    roofline = None
    if dom:
        d = kernels[dom]
        avg_ms = d['total_ms'] / d['launches']
        per_launch_bytes = d['bytes'] / d['launches'] if d['bytes'] else None
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9 if per_launch_bytes else None
        traffic, traffic_src = pmc_traffic(dom, p.oN, len(coils), img)
        roofline = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=(achieved / HBM_PEAK_GBS) if achieved else None, traffic=traffic,
                        traffic_source=traffic_src, avg_launch_ms=avg_ms, launches=d['launches'],
                        algorithmic_bytes_per_launch=per_launch_bytes)
    for k in sorted(prof, key=lambda k: -prof[k]['total_ms']):
        log(rank, "  %-24s %4d launches  avg %8.3f ms  total %9.2f ms" % (k, prof[k]['launches'], prof[k]['avg_ms'], prof[k]['total_ms']))
    log(rank, "eval: %.3f ms/step, %.2f evals/s; whole-eval algorithmic rate %.0f GB/s per GPU" % (
        ms_per_step, value, alg_bytes_rank / (ms_per_step * 1e-3) / 1e9))

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(p, C, log)

    if rank == 0:
        out = {
            "metric": "SENSE AHA evals/sec (256^3 x 8-coil non-Cartesian)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "complex64 (f32)", "data": "synthetic",
            "config": {"workload": "non-Cartesian SENSE A^H A, image %d^3, %d coils, grid %d^3 (osf 2), radial T=%d, KB width 4; "
                                   "-O3 tree S'->FFT->G'->G'^H->IFFT->S'^H" % (img, C, p.oN[0], p.T),
                       "parallelism": "coil-sharded x%d, one all-reduce per eval" % world if world > 1 else "single GPU",
                       "grid_layout": (layout if fused_fft else 0),
                       "algorithmic_GB_per_eval_per_gpu": alg_bytes_rank / 1e9},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            # whole-eval rate in the reference's own bytes model, and the per-call-site breakdown
            "eval_algorithmic_GBps_per_gpu": alg_bytes_rank / (ms_per_step * 1e-3) / 1e9,
            "eval_frac_of_hbm_peak": alg_bytes_rank / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "kernels": {k: {"launches_per_eval": v['launches'] / args.steps, "avg_ms": round(v['avg_ms'], 4),
                            "GBps": round(v['bytes'] / v['launches'] / (v['avg_ms'] * 1e-3) / 1e9, 1) if v['bytes'] else None}
                        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['total_ms'])},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def run_cpu_baseline(p, C, log):
    """numpy oracle on ONE coil of the same problem; evals/s = 1 / (C * t_one_coil)."""
    import numpy as np
    from indigo_amd.sense import normal_operator
    from indigo_amd.util import rand64c
    from oracle.np_backend import NumpyBackend
    t0 = time.time()
    O = NumpyBackend()
    A1 = p.build_fused(O, coils=[0])
    AHA1 = normal_operator(A1)
    x = O.copy_array(rand64c(A1.shape[1], 1, seed=1))
    y = O.zero_array((A1.shape[1], 1), np.dtype('complex64'))
    t1 = time.perf_counter()
    AHA1.eval(y, x)
    t = time.perf_counter() - t1
    log(0, "cpu baseline: one coil in %.2f s (setup %.1f s)" % (t, t1 - t0 if False else time.time() - t0 - t))
    return dict(value=1.0 / (C * t), unit="evals/s", cores=1, kind="port",
                sample="numpy oracle (restatement of indigo/backends/np.py), 1 of %d coils of the same problem, "
                       "one evaluation (%.1f s), scaled linearly in coils; single-threaded pocketfft + scipy csr_matvecs" % (C, t))


if __name__ == "__main__":
    main()
