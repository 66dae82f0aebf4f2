"""bench.py's launch contract, the part that needs no GPU: a world size that does not match --gpus is refused
before anything touches a device (driver contract: one rank per GPU, `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N`), and `--help` names the contract's flags."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=120)


def test_world_size_mismatch_is_refused_without_a_json_line():
    res = _run(["--gpus", "2"], dict(WORLD_SIZE="1", RANK="0"))
    assert res.returncode == 2 and "WORLD_SIZE" in res.stderr and not res.stdout.strip()
    res = _run(["--gpus", "1"], dict(WORLD_SIZE="4", RANK="0"))
    assert res.returncode == 2 and not res.stdout.strip()


def test_help_lists_the_contract_flags():
    res = _run(["--help"], {})
    assert res.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in res.stdout


def test_launch_ranks_restarts_without_graph_after_a_failed_capture(monkeypatch, tmp_path):
    """the self-launch parent (which never touches the GPU) starts FRESH ranks with --no-graph when the first
    set left the capture-failure marker behind; otherwise it relays the exit code"""
    import types
    bench = _load_bench()
    calls = []

    def fake_call(cmd, env=None):
        calls.append(list(cmd))
        if "--no-graph" not in cmd:
            open(env["NMV_BENCH_CAPTURE_MARKER"], "w").close()   # what rank 0 does before leaving with code 75
            return 1                                             # torch.distributed.run reports a failed worker
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "2", "--steps", "4"])
    args = types.SimpleNamespace(gpus=2, no_graph=False)
    assert bench.launch_ranks(args) == 0
    assert len(calls) == 2 and "--no-graph" not in calls[0] and calls[1][-1] == "--no-graph"
    # an ordinary failure (no marker) is not retried
    calls.clear()
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.append(cmd) or 3)
    assert bench.launch_ranks(args) == 3 and len(calls) == 1


def test_llama3_70b_tp8_fits_an_mi355x():
    """BASELINE.json configs[4]: the weights + KV of Llama-3-70B w4a16 at TP = 8 per rank, and what does not fit"""
    bench = _load_bench()
    from neural_magic_vllm_amd.worker import decode_runner as dr
    fit = bench.weights_fit(dr.LLAMA3_70B, "w4a16", 8, 64, 1024, "auto")
    # 4.4 GB of int4 weights + scales per rank (one tensor: the MFMA-native one) + the bf16 embedding / lm_head shards
    assert fit["fits"] and 4.4 < fit["weights_gb"] < 6.5
    os.environ["NMV_W4_KEEP_MARLIN"] = "1"     # both tensors resident: twice the codes
    try:
        both = bench.weights_fit(dr.LLAMA3_70B, "w4a16", 8, 64, 1024, "auto")
    finally:
        del os.environ["NMV_W4_KEEP_MARLIN"]
    assert both["fits"] and 8.0 < both["weights_gb"] < 12.0
    assert bench.weights_fit(dr.LLAMA3_70B, "bf16", 1, 64, 1024, "auto")["fits"]          # 141 GB of bf16: one MI355X holds it
    assert not bench.weights_fit(dr.LLAMA3_70B, "bf16", 1, 4096, 8192, "auto")["fits"]    # ... but not with 1.3 TB of KV
    # the rehearsal architecture has the TP = 8 per-rank head geometry at TP = 2
    assert dr.TINY_70B.num_attention_heads // 2 == 8 and dr.TINY_70B.num_key_value_heads // 2 == 1
