"""GPU: the `_C_custom_ar` op surface (registered-buffer all-reduce protocol of the reference: init_custom_ar,
register_buffer, all_reduce_reg / _unreg, get_graph_buffer_ipc_meta, register_graph_buffers, dispose) through
the reference-shaped `CustomAllreduce` communicator, two processes sharing the test box's one GPU.  Recipe of
tests/distributed/test_custom_all_reduce.py:22-78: eager calls, then calls captured into a graph whose inputs
are registered after the capture, checked against the fp32 sum in rank order (bit-exact here)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inp(numel, dtype, it, r):
    g = torch.Generator().manual_seed(131 * it + r)
    return torch.randn(numel, generator=g).to(dtype)


def _want(world, numel, dtype, it):
    ref = torch.zeros(numel, dtype=torch.float32)
    for r in range(world):
        ref += _inp(numel, dtype, it, r).float()
    return ref.to(dtype)


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0",
                          # ranks time-sharing one GPU: a longer flag-wait bound than the 2 s of a deployment
                          NMV_CUSTOM_AR_TIMEOUT_MS="30000")
        import torch.distributed as dist
        import neural_magic_vllm_amd  # noqa: F401  (registers torch.ops._C_custom_ar)
        from neural_magic_vllm_amd import _custom_ops as ops
        from neural_magic_vllm_amd import _lib
        from neural_magic_vllm_amd.distributed.device_communicators.custom_all_reduce import CustomAllreduce
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        fa = CustomAllreduce(dist.group.WORLD, dev, max_size=2 << 20)
        assert not fa.disabled
        assert ops.meta_size() == _lib.load().nmv_car_meta_size() > 0
        it = 0
        for dtype in (torch.bfloat16, torch.float16, torch.float32):
            for numel in (8, 4096, 64 * 4096, 3 * 4096 + 8):
                x = _inp(numel, dtype, it, rank).to(dev)
                assert fa.should_custom_ar(x)
                y = fa.custom_all_reduce(x)                      # eager: copy into the registered buffer + reduce
                assert y is not None and torch.equal(y.cpu(), _want(world, numel, dtype, it)), (dtype, numel)
                it += 1
        # not taken: size not a multiple of 16 bytes, larger than max_size
        assert fa.custom_all_reduce(torch.ones(5, dtype=torch.bfloat16, device=dev)) is None
        assert fa.custom_all_reduce(torch.ones((2 << 20) + 16, dtype=torch.uint8, device=dev).view(torch.bfloat16)) is None
        # an unregistered tensor handed to all_reduce_reg outside a capture is refused (custom_all_reduce.cuh:424-430)
        with pytest.raises(RuntimeError, match="not registered"):
            fa.all_reduce_reg(torch.ones(64, dtype=torch.bfloat16, device=dev))
        # graph: two dependent all-reduces on graph-private tensors, registered after the capture
        xs = torch.stack([_inp(64 * 512, torch.bfloat16, 900, r) for r in range(world)])
        src = xs[rank].to(dev)
        with fa.capture():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fa.custom_all_reduce(src + 0)                    # warm-up: allocation pattern only
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            dist.barrier()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                a = src + 0
                y1 = fa.custom_all_reduce(a)
                y2 = fa.custom_all_reduce(y1 * 0.5)
        torch.cuda.synchronize()
        dist.barrier()
        want1 = xs.float().sum(0).to(torch.bfloat16)
        want2 = ((want1 * 0.5).float() * world).to(torch.bfloat16)
        for _ in range(3):
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(y1.cpu(), want1) and torch.equal(y2.cpu(), want2)
        assert _lib.load().nmv_car_error(fa._ptr) == 0
        dist.barrier()
        fa.close()
        dist.destroy_process_group()
        q.put(("ok", None))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", f"rank {rank}: {e!r}\n{traceback.format_exc()}"))


def test_custom_ar_ops_registered_buffers_and_graph(gpu_device):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    errs = [m for s, m in res if s != "ok"]
    assert not errs, "\n".join(errs)
