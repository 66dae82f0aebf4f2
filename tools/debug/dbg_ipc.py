import os, sys, socket, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.multiprocessing as mp

def worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from neural_magic_vllm_amd import _lib
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    L = _lib.load()
    hip = ctypes.CDLL("libamdhip64.so")
    for label, t in (("small", torch.zeros(4096, dtype=torch.uint8, device="cuda")),
                     ("2MB+", torch.zeros(2 * 1024 * 1024 + 1792, dtype=torch.uint8, device="cuda")),
                     ("32MB", torch.zeros(32 << 20, dtype=torch.uint8, device="cuda"))):
        torch.cuda.synchronize()
        data = t.untyped_storage()._share_cuda_()
        h, off = bytes(data[1]), int(data[3])
        # also the handle the runtime gives for the allocation's base, as get_graph_buffer_ipc_meta does
        base = ctypes.c_void_p(); size = ctypes.c_size_t()
        rc = hip.hipMemGetAddressRange(ctypes.byref(base), ctypes.byref(size), ctypes.c_void_p(t.data_ptr()))
        hb = ctypes.create_string_buffer(64)
        rc2 = hip.hipIpcGetMemHandle(hb, base)
        mine = dict(torch_handle=h, torch_off=off, len=len(h), rt_handle=hb.raw, rt_rc=(rc, rc2), rt_off=t.data_ptr() - (base.value or 0), size=size.value)
        allv = [None, None]
        dist.all_gather_object(allv, mine)
        peer = allv[1 - rank]
        for kind in ("torch_handle", "rt_handle"):
            ptr = ctypes.c_void_p()
            hbuf = ctypes.create_string_buffer(peer[kind], 64)
            e = hip.hipIpcOpenMemHandle(ctypes.byref(ptr), hbuf, 1)
            print(f"rank {rank} {label} {kind}: len {peer['len']} same_as_rt {peer['torch_handle'] == peer['rt_handle']} open rc={e} ptr={ptr.value} toff={peer['torch_off']} rtoff={peer['rt_off']} size={peer['size']} getrc={peer['rt_rc']}", flush=True)
            if e == 0: hip.hipIpcCloseMemHandle(ptr)
        dist.barrier()
    dist.destroy_process_group()
    q.put("ok")

if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, port, q)) for r in range(2)]
    [p.start() for p in ps]; [p.join(120) for p in ps]
