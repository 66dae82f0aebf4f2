"""bench.py's launch contract, the part that needs no GPU: a world size that does not match --gpus is refused
before anything touches a device (driver contract: one rank per GPU, `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N`), and `--help` names the contract's flags."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
