"""`python bench.py --gpus N` without a launcher starts its own ranks (bench.py:self_launch).  What can be checked without a
GPU: the launcher itself never needs one, every rank gets the environment torch.distributed.run would give it, and a rank
that fails (here: all of them -- there is no GPU, and the product has no CPU fallback) makes the launcher exit non-zero."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_propagates_rank_failure_and_environment(tmp_path):
    probe = tmp_path / "sitecustomize.py"
    probe.write_text(
        "import os\n"
        "if os.environ.get('INDIGO_BENCH_LAUNCHER') == 'self':\n"
        "    open(os.path.join(os.environ['PROBE_DIR'], 'rank%s' % os.environ['RANK']), 'w').write(\n"
        "        ' '.join(os.environ.get(k, '?') for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'INDIGO_COMM_ID_FILE')))\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""), PROBE_DIR=str(tmp_path),
               INDIGO_BENCH_DIST_BACKEND="gloo")          # (rehearsal switch: no GPU count check in the launcher)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--image", "32"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    seen = sorted(f for f in os.listdir(tmp_path) if f.startswith("rank"))
    assert seen == ["rank0", "rank1"], (seen, r.stderr[-2000:])
    f0 = (tmp_path / "rank0").read_text().split()
    f1 = (tmp_path / "rank1").read_text().split()
    assert f0[:4] == ["0", "0", "2", "127.0.0.1"] and f1[:4] == ["1", "1", "2", "127.0.0.1"]
    assert f0[4] == f1[4] and not os.path.exists(os.path.dirname(f0[4])), "one private rendezvous directory, removed afterwards"
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "launcher: rank" in r.stderr, r.stderr[-2000:]
        assert r.stdout.strip() == ""
