"""rdycore_amd.launch: N ranks from a plain command (what `bench.py --gpus N` does before any GPU call)."""
import io
import os
import sys
import textwrap

from rdycore_amd.launch import launch_ranks, rank_environment


def _script(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_ranks_get_the_rendezvous_environment_and_rank0_is_relayed(tmp_path):
    argv = _script(tmp_path, """
        import json, os
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "RDY_TEST_PASSTHROUGH")
        print(json.dumps({k: os.environ.get(k) for k in keys}), flush=True)
    """)
    out = io.StringIO()
    os.environ["RDY_TEST_PASSTHROUGH"] = "yes"
    try:
        assert launch_ranks(3, argv, timeout=60, out=out) == 0
    finally:
        del os.environ["RDY_TEST_PASSTHROUGH"]
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1                       # only rank 0's stdout is relayed
    import json
    d = json.loads(lines[0])
    assert d["RANK"] == "0" and d["LOCAL_RANK"] == "0" and d["WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1"
    assert int(d["MASTER_PORT"]) > 0
    # the caller's environment reaches the ranks unchanged; HSA_ENABLE_IPC_MODE_LEGACY (RCCL's intra-node transport depends
    # on it) is passed through as it is -- the launcher invents no default for it
    assert d["RDY_TEST_PASSTHROUGH"] == "yes" and d["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    assert "HSA_ENABLE_IPC_MODE_LEGACY" not in rank_environment(0, 2, 1, base={})


def test_a_failing_rank_ends_the_others_and_sets_the_return_code(tmp_path):
    argv = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(120)      # would wait in a collective for ever
    """)
    import time
    t0 = time.time()
    rc = launch_ranks(3, argv, timeout=100, out=io.StringIO())
    assert rc == 7 and time.time() - t0 < 60


def test_ranks_form_a_gloo_group(tmp_path):
    argv = _script(tmp_path, """
        import os, torch, torch.distributed as dist
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(int(t.item()), flush=True)
        dist.destroy_process_group()
    """)
    out = io.StringIO()
    assert launch_ranks(2, argv, timeout=120, out=out, keep=lambda ln: ln.strip().isdigit()) == 0     # gloo prints a connection note on stdout
    assert out.getvalue().strip() == "3"


def test_bench_parent_does_not_touch_the_gpu():
    """`bench.py --gpus N` as a plain command becomes the launcher before any torch.cuda call: the parent's code path
    imports neither torch nor the operator"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    main_src = src[src.index("def main("):]
    launcher = main_src[:main_src.index("run_rank(args, argv)")]
    assert "launch_ranks" in launcher and "torch" not in launcher.replace("torch.cuda.*", "").replace("torch.distributed.run", "")
    head = src[:src.index("def parse(")]
    assert "import torch" not in head             # module import time is GPU-free as well
    env = rank_environment(1, 4, 12345, base={})
    assert env["RANK"] == "1" and env["WORLD_SIZE"] == "4" and env["MASTER_PORT"] == "12345"
