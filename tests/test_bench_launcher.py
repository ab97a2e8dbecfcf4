"""CPU: `python bench.py --gpus N` started WITHOUT a launcher around it must start its N ranks itself (bench.launch_ranks), relay exactly ONE
JSON line and the children's failure; and the two-phase agreement of bench.try_rccl_comm must keep the control group's collectives matched
when ONE rank fails alone.  No GPU: the ranks here are stand-in scripts on the gloo backend."""
import io
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _script(tmp_path, body):
    p = tmp_path / "rank_script.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


RANK_OK = """
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    print("a banner some library printed on stdout, rank", rank)          # noise on stdout from every rank
    print(json.dumps({"not": "the result line"}))                          # JSON, but not a bench line
    if rank == 0:
        print(json.dumps({"metric": "m", "value": float(t.item()), "n_gpus": world, "argv": sys.argv[1:]}))
    dist.destroy_process_group()
"""


def test_launcher_relays_exactly_one_json_line(tmp_path, capfd):
    out = io.StringIO()
    rc = bench.launch_ranks(2, [_script(tmp_path, RANK_OK), "--gpus", "2", "--steps", "3"], out=out, timeout=300)
    assert rc == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["metric"] == "m" and res["value"] == 3.0 and res["n_gpus"] == 2
    assert res["argv"] == ["--gpus", "2", "--steps", "3"]  # the ranks see the parent's own flags
    err = capfd.readouterr().err
    assert "a banner some library printed" in err  # what else reached the children's stdout went to stderr, not into the line


def test_launcher_reports_a_failing_rank(tmp_path):
    body = """
        import os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")
        if int(os.environ["RANK"]) == 1:
            sys.exit(7)
        dist.barrier()
    """
    out = io.StringIO()
    rc = bench.launch_ranks(2, [_script(tmp_path, body)], out=out, timeout=300)
    assert rc != 0
    assert out.getvalue() == ""


def test_launcher_without_a_result_line_is_an_error(tmp_path):
    out = io.StringIO()
    rc = bench.launch_ranks(1, [_script(tmp_path, "print('nothing to report')\n")], out=out, timeout=300)
    assert rc == 1 and out.getvalue() == ""


def test_bench_main_takes_the_launcher_path_before_any_gpu_call(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent must go through torch.distributed.run.  Here there is no GPU, so the ranks fail
    at their first GPU call — AFTER having been started as ranks (WORLD_SIZE = 2 in their environment), which the error text shows; the parent
    relays a non-zero exit code and prints no line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "C1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run the real bench")
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "launch with torch.distributed.run" not in p.stderr  # the old refusal is gone
    assert "torch.distributed.run exited with" in p.stderr


AGREE = """
    import os, sys
    sys.path.insert(0, {root!r})
    import torch, torch.distributed as dist
    import bench
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    comm, path, why = bench.try_rccl_comm(torch, dist, rank, world)
    assert comm is None and why, (comm, why)
    # the control group is still usable and matched: one more collective, same result everywhere
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    assert t.item() == 3.0
    if rank == 0:
        import json
        print(json.dumps({{"metric": "agree", "why": why}}))
    dist.destroy_process_group()
"""


@pytest.mark.parametrize("where", ["rank1", "rank0", "1"])
def test_comm_setup_failure_on_one_rank_keeps_the_ranks_matched(tmp_path, where, monkeypatch):
    """BENCH_FORCE_COMM_FAIL=rank1: only rank 1 fails, in phase 1 (before any broadcast) — the asymmetric case that used to leave rank 0 in
    broadcast and rank 1 in all_reduce.  With the two-phase agreement both ranks fall back together."""
    monkeypatch.setenv("BENCH_FORCE_COMM_FAIL", where)
    out = io.StringIO()
    rc = bench.launch_ranks(2, [_script(tmp_path, AGREE.format(root=ROOT))], out=out, timeout=300)
    assert rc == 0, "the ranks hung or disagreed"
    res = json.loads(out.getvalue())
    assert res["metric"] == "agree"
