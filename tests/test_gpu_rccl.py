"""-m gpu: RCCL executes on this box's one GPU.  Every other multi-rank test runs on gloo (two ranks
sharing the GPU, which RCCL refuses); here the process group is ``nccl`` with ONE rank and the step runs in its sharded, captured
form (tests/rccl_one_rank_worker.py) -- bit-identical to the non-distributed step.  The record lands in gpurun_out/ for profiles/."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_captured_step_on_a_one_rank_rccl_group_equals_the_plain_step(tmp_path):
    from allrank_amd.launch import free_port
    out = str(tmp_path / "rccl.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py"), out], env=env, capture_output=True,
                       text=True, timeout=420)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-4000:])
    rec = json.load(open(out))
    assert rec["backend"] == "nccl" and rec["world"] == 1
    assert [j["loss"] for j in rec["jobs"]] == ["approxNDCGLoss", "neuralNDCG", "lambdaLoss", "listNet"]
    assert all(j["graph_segments"] >= 1 + j["buckets"] for j in rec["jobs"]) and rec["short_batch_captures"] == 2
    # neuralNDCG / lambdaLoss(mean) cut one more segment than the plain losses: the normaliser all-reduce between the loss phases
    seg = {j["loss"]: j["graph_segments"] for j in rec["jobs"]}
    assert seg["neuralNDCG"] > seg["approxNDCGLoss"] and seg["lambdaLoss"] > seg["approxNDCGLoss"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_one_rank.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
