"""CPU, world_size 2, gloo: the tensor-parallel path that bench.py --gpus N takes over RCCL.
Covers GroupCoordinator collectives (reference tests/distributed/test_comm_ops.py recipe: all_reduce
of arange*(r+1) vs the stacked sum, all_gather, broadcast_tensor_dict) and the Megatron sharding of
the linear / embedding layers against an unsharded single-process computation."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist
    from neural_magic_vllm_amd import distributed as nd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nd.initialize_model_parallel(world, backend="gloo", local_rank=rank)
    return dist, nd


def _worker_comm(rank, world, port, q):
    try:
        dist, nd = _init(rank, world, port)
        n = 8 * 64
        # all_reduce
        t = torch.arange(n, dtype=torch.float32) * (rank + 1)
        exp = sum(torch.arange(n, dtype=torch.float32) * (r + 1) for r in range(world))
        out = nd.tensor_model_parallel_all_reduce(t.clone())
        assert torch.equal(out, exp)
        # all_gather along last and first dim
        x = torch.full((2, 3), float(rank))
        g = nd.tensor_model_parallel_all_gather(x, dim=-1)
        assert g.shape == (2, 3 * world) and all(bool((g[:, 3 * r:3 * r + 3] == r).all()) for r in range(world))
        g0 = nd.tensor_model_parallel_all_gather(x, dim=0)
        assert g0.shape == (2 * world, 3) and bool((g0[2 * rank:2 * rank + 2] == rank).all())
        # gather to rank 0
        ga = nd.tensor_model_parallel_gather(x, dst=0, dim=-1)
        assert (ga is None) == (rank != 0)
        if rank == 0:
            assert ga.shape == (2, 3 * world)
        # broadcast_tensor_dict (worker_base.py:246-249)
        if rank == 0:
            d = {"a": torch.arange(5), "b": torch.ones(2, 2, dtype=torch.bfloat16), "c": "meta", "n": 7,
                 "e": torch.empty(0)}
            nd.broadcast_tensor_dict(d, src=0)
        else:
            d = nd.broadcast_tensor_dict(None, src=0)
            assert torch.equal(d["a"], torch.arange(5)) and d["c"] == "meta" and d["n"] == 7
            assert d["b"].dtype == torch.bfloat16 and d["e"].numel() == 0
        nd.destroy_model_parallel()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def _worker_layers(rank, world, port, q):
    try:
        dist, nd = _init(rank, world, port)
        from neural_magic_vllm_amd.model_executor.layers.linear import (
            ColumnParallelLinear, MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear)
        from neural_magic_vllm_amd.model_executor.layers.logits_processor import LogitsProcessor
        from neural_magic_vllm_amd.model_executor.layers.vocab_parallel_embedding import (
            ParallelLMHead, VocabParallelEmbedding)
        torch.manual_seed(0)  # identical full weights on every rank
        dt = torch.float32
        x = torch.randn(5, 64)
        # column -> row parallel MLP == dense MLP
        w1, w2 = torch.randn(96, 64), torch.randn(64, 96)
        col = ColumnParallelLinear(64, 96, bias=False, params_dtype=dt)
        row = RowParallelLinear(96, 64, bias=False, params_dtype=dt)
        col.weight.weight_loader(col.weight, w1)
        row.weight.weight_loader(row.weight, w2)
        y, _ = row(col(x)[0])
        assert torch.allclose(y, (x @ w1.t()) @ w2.t(), atol=1e-4, rtol=1e-4)
        # gather_output
        colg = ColumnParallelLinear(64, 96, bias=False, gather_output=True, params_dtype=dt)
        colg.weight.weight_loader(colg.weight, w1)
        assert torch.allclose(colg(x)[0], x @ w1.t(), atol=1e-4, rtol=1e-4)
        # merged column (gate_up): shard ids 0/1
        wg, wu = torch.randn(32, 64), torch.randn(32, 64)
        mc = MergedColumnParallelLinear(64, [32, 32], bias=False, params_dtype=dt)
        mc.weight.weight_loader(mc.weight, wg, 0)
        mc.weight.weight_loader(mc.weight, wu, 1)
        out = mc(x)[0]
        half = 32 // world
        assert torch.allclose(out[:, :half], (x @ wg.t())[:, rank * half:(rank + 1) * half], atol=1e-4)
        assert torch.allclose(out[:, half:], (x @ wu.t())[:, rank * half:(rank + 1) * half], atol=1e-4)
        # qkv with 4 q heads / 2 kv heads of size 8
        wq, wk, wv = torch.randn(32, 64), torch.randn(16, 64), torch.randn(16, 64)
        qkv = QKVParallelLinear(64, 8, 4, 2, bias=False, params_dtype=dt)
        for sid, w in (("q", wq), ("k", wk), ("v", wv)):
            qkv.weight.weight_loader(qkv.weight, w, sid)
        o = qkv(x)[0]
        nq, nkv = 4 // world, 2 // world
        assert torch.allclose(o[:, :nq * 8], (x @ wq.t())[:, rank * nq * 8:(rank + 1) * nq * 8], atol=1e-4)
        assert torch.allclose(o[:, nq * 8:nq * 8 + nkv * 8], (x @ wk.t())[:, rank * nkv * 8:(rank + 1) * nkv * 8], atol=1e-4)
        # vocab parallel embedding + logits gather
        emb_w = torch.randn(100, 64)
        emb = VocabParallelEmbedding(100, 64, params_dtype=dt)
        emb.weight.weight_loader(emb.weight, emb_w)
        ids = torch.tensor([0, 49, 50, 63, 64, 99])
        assert torch.allclose(emb(ids), emb_w[ids], atol=1e-6)
        head = ParallelLMHead(100, 64, params_dtype=dt)
        head.weight.weight_loader(head.weight, emb_w)
        logits = LogitsProcessor(100)(head.weight, x)
        if rank == 0:
            assert torch.allclose(logits, x @ emb_w.t(), atol=1e-4, rtol=1e-4)
        else:
            assert logits is None
        nd.destroy_model_parallel()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def _worker_capture_failure(rank, world, port, q):
    """rank 1's hipGraph capture 'fails': the agreement over the CPU group must raise on EVERY rank (no rank goes
    on eagerly in the same process; round 1's SIGSEGV)"""
    try:
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        from neural_magic_vllm_amd.worker.decode_runner import CaptureFailedError, agree_on_capture
        grp = dist.new_group(list(range(world)), backend="gloo")
        assert agree_on_capture(True, world, grp) is True            # everybody captured: fine
        try:
            agree_on_capture(rank != 1, world, grp)
            q.put((rank, "no exception"))
            return
        except CaptureFailedError as e:
            assert "rank(s) [1]" in str(e), str(e)
        assert agree_on_capture(False, 1, None) is False              # TP = 1 keeps its eager fallback
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("worker", [_worker_comm, _worker_layers, _worker_capture_failure],
                         ids=["collectives", "tp_layers", "capture_failure"])
def test_world_size_2_gloo(worker):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_single_process_defaults():
    """without an initialised group the wrappers are identities (TP=1)"""
    from neural_magic_vllm_amd import distributed as nd
    t = torch.arange(4.0)
    assert nd.get_tensor_model_parallel_world_size() == 1
    assert torch.equal(nd.tensor_model_parallel_all_reduce(t), t)
    assert torch.equal(nd.tensor_model_parallel_all_gather(t), t)
