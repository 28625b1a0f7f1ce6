#!/usr/bin/env python3
"""CG on the SENSE normal equations (the caller of the hot path, SURVEY 8f / indigo/backends/backend.py:639-689):
iterations per second on the headline problem, next to the bare A^H A evaluation rate.

    python tools/cg_bench.py [iterations]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from indigo_amd.backends import get_backend
from indigo_amd.sense import SenseProblem, normal_operator
from indigo_amd.util import rand64c

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = get_backend("hip")
p = SenseProblem.synthetic((256,) * 3, 8, nspokes=3617, nreadout=512, width=2, ntable=128, oversamp=2.0, seed=4)
A = p.build_zpadfft(B)
AHA = normal_operator(A, lamda=0.01)
b = A.H * rand64c(A.shape[0], 1, seed=2)            # right-hand side A^H k
x0 = np.zeros_like(b)
B.cg(AHA, b, x0.copy(order='F'), maxiter=2)          # warm-up: plans, scratch, transposes
b_d = B.copy_array(b)                                # device-resident right-hand side and iterate
x_d = B.zero_array(b.shape, b.dtype)
B.barrier()
t0 = time.perf_counter()
hist = B.cg(AHA, b_d, x_d, maxiter=iters)
B.barrier()
t = time.perf_counter() - t0
# the same solve on plain launches (no HIP graph), and the steady state of the graph route: a long solve minus a short one
def solve(n, graph):
    B.tuning['cg_graph'] = graph
    xx = B.zero_array(b.shape, b.dtype)
    B.barrier()
    t1 = time.perf_counter()
    B.cg(AHA, b_d, xx, maxiter=n, tol=0.0)
    B.barrier()
    return time.perf_counter() - t1
for graph in (False, True):
    ta, tb = solve(30, graph), solve(130, graph)
    print("cg_graph=%s: 30 iterations %.1f ms, 130 iterations %.1f ms -> steady state %.3f ms/iteration" % (graph, ta * 1e3, tb * 1e3, (tb - ta) * 1e3 / 100))
B.tuning['cg_graph'] = False          # (the product's default: the replay saves 0.03 ms per iteration and costs 5 ms per solve to record)
print("history:", " ".join("%.3f" % h for h in hist)); print("CG: %d iterations in %.1f ms -> %.2f ms/iteration (%.1f it/s); relative residual %.3e -> %.3e" % (
    len(hist), t * 1e3, t * 1e3 / max(len(hist), 1), len(hist) / t, hist[0], hist[-1]))

if os.environ.get("CG_PROFILE"):
    B.profile(True)
    B.cg(AHA, b, x0.copy(order='F'), maxiter=iters)
    B.barrier()
    B.profile(False)
    rep = B.profile_report()
    tot = sum(v['total_ms'] for v in rep.values())
    print("profiled kernel time per iteration: %.2f ms" % (tot / iters))
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]['total_ms']):
        print("  %-24s %5d launches  avg %7.3f ms  per-iteration %6.3f ms" % (k, v['launches'], v['avg_ms'], v['total_ms'] / iters))
