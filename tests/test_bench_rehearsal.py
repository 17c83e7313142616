"""bench.py's N > 1 path rehearsed by ONE rank on the RCCL backend: the process group on the device, the barriers, the
exchange's library reduce-scatter / all-gather, its direct all-to-all form and the all-reduce exchange all execute (with
one participant), and standard output carries exactly one JSON line although RCCL prints a banner of its own."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("form", ["library", "direct", "allreduce"])
def test_one_rank_takes_the_multi_gpu_path_on_rccl(gpu, form):
    env = dict(os.environ, POGS_BENCH_FORCE_DIST="1", POGS_EXCHANGE_SINGLE_RANK_COLLECTIVES="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29541 + ["library", "direct", "allreduce"].index(form)))
    env.pop("POGS_EXCHANGE_DIRECT", None)
    if form == "direct":
        env["POGS_EXCHANGE_DIRECT"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--gaussians", "200000", "--width", "640", "--height", "360", "--views", "2"]
    if form == "allreduce":
        cmd += ["--exchange", "allreduce"]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=170)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, done.stdout[:500]
    rec = json.loads(lines[0])
    assert rec["value"] > 0 and rec["n_gpus"] == 1 and "dp1" in rec["config"]["parallelism"]
    assert ("all-reduce" in rec["config"]["parallelism"]) == (form == "allreduce")
    assert ("direct all-to-all" in rec["config"]["parallelism"]) == (form == "direct")
