"""bench.py's launcher logic without a GPU: `--gpus N` with no launcher starts N rank processes itself; every rank that
cannot get its GPU exits with code 3 before any rendezvous, and the parent reports that code (nothing hangs, no CPU
fallback runs).  On a GPU box the same entry point is exercised by tests/test_gpu_parallel.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="needs a box without GPUs (the GPU path is tests/test_gpu_parallel.py)")
def test_bench_without_gpus_fails_loudly_in_every_launch_form():
    env = dict((k, v) for k, v in os.environ.items()
               if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'))
    # one rank, no launcher
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '1', '--warmup', '0'], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and 'needs GPU' in r.stderr and r.stdout.strip() == ''
    # two ranks started by bench.py itself: both exit 3, the parent relays it and prints no record
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0'], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    assert r.stderr.count('needs GPU') == 2 and 'rank exit codes [3, 3]' in r.stderr and r.stdout.strip() == ''
    # under a launcher (WORLD_SIZE set): the rank itself exits 3
    env2 = dict(env, WORLD_SIZE='2', RANK='1', LOCAL_RANK='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0'], cwd=ROOT, env=env2,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and 'rank 1 needs GPU 1' in r.stderr


def test_self_launch_takes_the_other_ranks_down_when_one_fails(tmp_path):
    """One rank dies (exit code 5) while the others would sit in a collective: the parent ends them (its own children,
    by PID) within seconds and reports the failing code; a clean run relays rank 0's last line."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    script = tmp_path / 'rank.py'
    script.write_text(
        "import os, sys, time\n"
        "r = int(os.environ['RANK']); mode = sys.argv[1]\n"
        "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "if mode == 'fail' and r == 1:\n"
        "    sys.exit(5)\n"
        "if mode == 'fail':\n"
        "    time.sleep(120)\n"
        "print('banner of rank %d' % r)\n"
        "print('{\"record\": %d}' % r)\n")
    t0 = time.time()
    assert bench.self_launch(3, script=str(script), argv=['fail']) == 5
    assert time.time() - t0 < 30.0
    assert bench.self_launch(3, script=str(script), argv=['ok']) == 0
