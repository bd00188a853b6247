"""`python bench.py --gpus N` must start its own ranks (the reference's driver, `render_parallel_iterations`
tracing/src/worker.rs:84-119, spawns its own workers): with N > 1 and no WORLD_SIZE the script - before torch or a GPU
is touched - starts N fresh rank processes as children, relays rank 0's ONE JSON line and exits with their status.
Rehearsed here without a GPU: PTX_BENCH_DRYRUN=1 makes the ranks use gloo and a stand-in for the renderer (every
sample adds 1.0), so launcher, rendezvous, reduce and the one-line contract all run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_gpus_2_without_a_launcher_starts_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--scene", "cornell", "--size", "48x32", "--spp-per-step", "5"],
             {"PTX_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # ONE line on stdout, whatever the ranks and their libraries print
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["dryrun"] is True and out["dryrun_sum_ok"] is True  # both ranks' buffers arrived on rank 0
    assert abs(out["value"] * out["ms_per_step"] * 3e-3 - 2 * 3 * 5 * 48 * 32) < 1e-3  # whole-job samples of both ranks
    assert out["scaling"] == "weak" and out["vs_baseline"] is None and out["dtype"] == "f32"
    assert "starting 2 ranks" in r.stderr


def test_the_parent_touches_neither_torch_nor_a_gpu():
    # the parent's whole job happens before torch is imported: a process that has initialised the GPU must not be the
    # one that starts other programs
    code = ("import sys; sys.path.insert(0, %r); import bench; a = bench.parse_args(['--gpus', '4']); "
            "assert 'torch' not in sys.modules and 'pathtracer_amd' not in sys.modules; print(a.gpus)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "4", r.stderr


def test_a_failing_rank_fails_the_parent():
    # WORLD_SIZE that disagrees with --gpus is a launch mistake: the rank refuses, and so does the script
    r = _run(["--gpus", "2", "--scene", "cornell", "--size", "16x16"], {"PTX_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
    assert r.stdout.strip() == ""
