"""The multi-rank flow of bench.py executed for real on ONE GPU (`OQ_BENCH_REHEARSAL=1`: ranks share cuda:0, collectives on
gloo with host staging): LPT plans, the sharded GPTQ run with batched factors, the wave-by-wave streamed gather, max-over-ranks timing,
every rank verifying its own first layers.  Not a scaling measurement -- the scaling bench is the driver's, on 8 GPUs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_runs_the_multi_rank_path_on_one_gpu(ranks):
    env = dict(os.environ, OQ_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29650 + ranks))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1",
                        "--gptq-layers", "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == ranks and "rehearsal" in line
    assert line["verified_vs_reference_digest"] is True
    g = line["gather"]
    assert g["ranks_seen"] == ranks and g["rank0_result_intact"] is True and g["gather_bytes"] > 0
    q = line["gptq"]
    assert q["n_gpus"] == ranks and q["ranks_seen"] == ranks
    assert q["verified"] is True, q["verification"]
    assert sorted(b["rank"] for b in q["verification"]["by_rank"]) == list(range(ranks))
    assert all(b["ok"] for b in q["verification"]["by_rank"])
    assert q["verification"]["gathered_equals_senders"] is True          # the streamed gather delivered the senders' bytes
    assert q["gather_bytes"] > 0 and q["cpu_baseline" if "cpu_baseline" in q else "value"] is not None


@pytest.mark.gpu
def test_bench_gptq_padded_gather_switch_on_one_gpu():
    """`--gather padded`: the fallback of the streamed point-to-point gather -- one padded collective gather
    after the last kernel -- through the same two-rank run; what arrives on rank 0 equals what the senders hold."""
    env = dict(os.environ, OQ_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29661")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gptq-layers", "2",
                        "--gptq-gather", "padded", "--gptq-extra-passes", "", "--no-model-rtn", "--no-awq", "--no-calibration", "--no-seam"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    q = json.loads(r.stdout.strip().splitlines()[-1])["gptq"]
    assert q["n_gpus"] == 2 and q["ranks_seen"] == 2 and q["verified"] is True
    assert "padded" in q["gather_how"] and q["verification"]["gathered_equals_senders"] is True and q["gather_bytes"] > 0
