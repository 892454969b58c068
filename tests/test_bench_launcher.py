"""`python bench.py --gpus N` starts N rank processes by itself (no torchrun): rehearsed here on CPU with the
gloo backend and no kernels (--rehearse prints a line without a value), so the launcher, the environment it
gives the ranks and the process group they form are covered without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=240, env=env)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_gpus_2_starts_two_ranks_that_form_one_group():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    rc, out, err = _run(["--gpus", "2", "--rehearse"], env)
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out                    # one JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_in_group"] == 2 and d["max_rank_seen"] == 1
    assert d["rehearsal"] is True and d["value"] is None


def test_failed_rank_fails_the_launcher():
    """Without --rehearse and without a GPU every rank refuses to run: the launcher must report failure."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    rc, out, err = _run(["--gpus", "2", "--legs", "", "--no-cpu-baseline"], env)
    assert rc != 0 and out.strip() == ""
    assert "needs a GPU" in err or "liborbx" in err or "HIP" in err
