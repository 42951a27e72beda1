"""`bench.py --gpus N` must never print a line for fewer GPUs than it was asked for: without a launcher it starts
N ranks itself (torch.distributed.run) or fails; with a WORLD_SIZE that disagrees it fails.  In this container there
is no GPU, so both cases must exit non-zero without a JSON line.  The launching parent stays GPU-free."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK") + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, env=env, timeout=300)


def test_gpus_2_without_launcher_spawns_or_fails():
    import torch

    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--width", "64", "--spp", "1", "--grid", "2",
              "--no-cpu-baseline"])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if torch.cuda.device_count() >= 2:  # a multi-GPU box: the launcher must have produced a 2-GPU line
        import json

        assert r.returncode == 0, r.stderr[-2000:]
        assert json.loads(lines[-1])["n_gpus"] == 2
    else:
        assert r.returncode != 0 and not lines
        assert "visible" in r.stderr


def test_launcher_parent_stays_gpu_free():
    """The parent that starts the ranks must not touch a GPU runtime (a process that has may not start other GPU programs
    on the pool's boxes): it counts devices from sysfs and imports neither torch nor the library."""
    code = ("import sys, bench; n = bench.visible_gpu_count(); "
            "assert 'torch' not in sys.modules and 'rayz_amd' not in sys.modules, sorted(sys.modules); "
            "print('count', n)")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("count"), r.stderr[-1000:]
    src = open(BENCH).read()
    body = src[src.index("def launch_ranks"):src.index("def main()")]
    assert "import torch" not in body and "device_count" not in body
    # HIP_VISIBLE_DEVICES narrows the count without any runtime call
    r = subprocess.run([sys.executable, "-c", "import bench; print(bench.visible_gpu_count())"], capture_output=True, text=True,
                       cwd=ROOT, env=dict(os.environ, HIP_VISIBLE_DEVICES=""), timeout=120)
    assert r.stdout.strip() in ("0", "None")


def test_world_size_mismatch_fails():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    r = _run(["--gpus", "0"])
    assert r.returncode != 0
