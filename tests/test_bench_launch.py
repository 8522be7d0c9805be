"""`python bench.py --gpus N` must be runnable from a plain shell: the parent starts N rank
processes with torch.distributed.run before touching the GPU and relays rank 0's JSON line.  CPU test of that
launcher with a stand-in rank script (gloo, world_size 2); the real thing runs in tests/test_gpu_forward.py."""
import json
import os
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(dist.get_rank() + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if "--fail" in sys.argv and dist.get_rank() == 1:
        sys.exit(3)
    if "--atomic-noise" in sys.argv:
        sys.stdout.write("noise from rank %d\\n" % dist.get_rank()); sys.stdout.flush()      # one write(): cannot split another line
    else:
        print("noise", "from", "rank", dist.get_rank(), "x" * 200, flush=True)                 # several writes: may land INSIDE rank 0's line
    if dist.get_rank() == 0:
        line = json.dumps({"metric": "m", "n_gpus": dist.get_world_size(), "max": t.item(), "argv": sys.argv[1:]})
        print(line, flush=True)
        if "--no-file" not in sys.argv:
            with open(os.environ["SAGE_BENCH_RESULT_FILE"], "w") as fh:
                fh.write(line + "\\n")
    dist.barrier()
    dist.destroy_process_group()
""")


def test_self_launch_relays_rank0_line(tmp_path, capfd):
    """Rank 0's line reaches the parent through the result file: the ranks share one stdout pipe, where another rank's output
    can land in the middle of the line (seen here with a multi-write print, ~3 % of the runs)."""
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    rc = bench.self_launch(2, ["--gpus", "2", "--steps", "3"], script=str(script), timeout=300)
    cap = capfd.readouterr()
    out = cap.out.strip().splitlines()
    assert rc == 0 and len(out) == 1, cap.err[-2000:]
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["max"] == 2.0 and line["argv"] == ["--gpus", "2", "--steps", "3"]


def test_self_launch_falls_back_to_stdout_without_a_result_file(tmp_path, capfd):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    rc = bench.self_launch(2, ["--no-file", "--atomic-noise"], script=str(script), timeout=300)
    cap = capfd.readouterr()
    out = cap.out.strip().splitlines()
    assert rc == 0 and len(out) == 1, cap.err[-2000:]
    assert json.loads(out[0])["argv"] == ["--no-file", "--atomic-noise"]


def test_self_launch_fails_loudly_when_a_rank_dies(tmp_path, capfd):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    rc = bench.self_launch(2, ["--fail"], script=str(script), timeout=300)
    assert rc != 0
    assert capfd.readouterr().out.strip() == ""


def test_host_plan_turns_the_role_threads_on_only_where_every_rank_has_five_cores(monkeypatch):
    """four spinning role threads + the submitter per rank: on only if usable cores // ranks >= 5; the line says which mode ran."""
    import types
    args = types.SimpleNamespace(host_threads=1, roles="SGDL")
    monkeypatch.setattr(bench, "usable_host_cores", lambda: 16)
    h1, h2, h4 = bench.host_plan(args, 1), bench.host_plan(args, 2), bench.host_plan(args, 4)
    assert (h1["cores_per_rank"], h1["role_threads"]) == (16, True) and (h2["cores_per_rank"], h2["role_threads"]) == (8, True)
    assert (h4["cores_per_rank"], h4["role_threads"], h4["enqueue_mode"]) == (4, False, "submitting thread only") and "oversubscribe" in h4["why"]
    monkeypatch.setattr(bench, "usable_host_cores", lambda: 192)
    assert bench.host_plan(args, 8)["role_threads"] is True and bench.host_plan(args, 8)["cores_per_rank"] == 24
    assert bench.host_plan(types.SimpleNamespace(host_threads=0, roles="SGDL"), 1)["role_threads"] is False
    assert bench.host_plan(types.SimpleNamespace(host_threads=1, roles="SGDD"), 1)["role_threads"] is False      # three streams: no role threads
    assert 1 <= bench.usable_host_cores.__wrapped__() if hasattr(bench.usable_host_cores, "__wrapped__") else True
