"""bench.py's multi-rank plumbing without a GPU: `python bench.py --gpus 2 --stub` must start two ranks ITSELF (a
torch.distributed.run child on 127.0.0.1), run barrier / max-over-ranks on gloo and print one JSON line whose `n_gpus`
equals --gpus; a world size that does not match --gpus is an error, never a silent one-rank run."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_spawns_its_own_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 3 and d["data"] == "stub" and d["value"] is None


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    r = _run(["--gpus", "2", "--stub"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1", "--stub"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_single_rank_stub_line():
    r = _run(["--stub", "--steps", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1


def test_collectives_selftest_on_gloo_one_and_two_ranks():
    """`--nccl-selftest` with the CPU backend: every kind of exchange the multi-rank path issues, in a group
    of ONE rank (forced through the collectives) and of two (bench.py starts the ranks itself)."""
    for gpus in (1, 2):
        r = _run(["--gpus", str(gpus), "--nccl-selftest", "--selftest-backend", "gloo"])
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"rccl_selftest"')]
        assert len(lines) == 1, r.stdout
        rec = json.loads(lines[0])["rccl_selftest"]
        assert rec["ok"] and rec["world"] == gpus and rec["backend"] == "gloo" and rec["error"] is None
        want = {"store_rendezvous", "all_reduce_x3", "all_gather_object", "padded_gather", "hessian_all_reduce", "streamed_gather", "barrier"}
        assert want <= set(rec["steps"]) and ("rank0_handshake" in rec["steps"]) == (gpus > 1)
