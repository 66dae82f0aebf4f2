"""Runs a sample of the registered ops on seeded inputs and saves the outputs: tests/test_gpu_bindings.py runs it
once per binding (NMV_BINDING=cpp / python, a process each -- a namespace can be registered once per process) and
compares the files bit for bit."""
import sys

import torch


def main(path: str) -> None:
    import neural_magic_vllm_amd  # noqa: F401
    from neural_magic_vllm_amd import _custom_ops as ops
    from neural_magic_vllm_amd import _torch_bindings as tb
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    import helpers
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    out = {"binding": tb.binding}
    x = torch.randn(37, 1024, generator=g).to(torch.bfloat16).to(dev)
    w = torch.randn(1024, generator=g).to(torch.bfloat16).to(dev)
    y = torch.empty_like(x)
    ops.rms_norm(y, x, w, 1e-5)
    out["rms_norm"] = y
    res = torch.randn(37, 1024, generator=g).to(torch.bfloat16).to(dev)
    x2 = x.clone()
    ops.fused_add_rms_norm(x2, res, w, 1e-5)
    out["fused_add_rms_norm"] = torch.stack([x2, res])
    gu = torch.randn(37, 2048, generator=g).to(torch.bfloat16).to(dev)
    act = torch.empty(37, 1024, dtype=torch.bfloat16, device=dev)
    ops.silu_and_mul(act, gu)
    out["silu_and_mul"] = act
    pos = torch.randint(0, 512, (37, ), generator=g).to(dev)
    q = torch.randn(37, 8 * 128, generator=g).to(torch.bfloat16).to(dev)
    k = torch.randn(37, 2 * 128, generator=g).to(torch.bfloat16).to(dev)
    cs = torch.randn(512, 128, generator=g).to(torch.bfloat16).to(dev)
    ops.rotary_embedding(pos, q, k, 128, cs, True)
    out["rotary_embedding"] = torch.cat([q, k], dim=1)
    inp = helpers.make_paged_attention_inputs(0, 4, (8, 2), 128, 16, torch.bfloat16, seq_lens=[1, 17, 300, 777],
                                              num_blocks=128)
    qq, kc, vc = inp["query"].to(dev), inp["key_cache"].to(dev), inp["value_cache"].to(dev)
    o = torch.empty_like(qq)
    ops.paged_attention_v1(o, qq, kc, vc, 2, inp["scale"], inp["block_tables"].to(dev), inp["seq_lens"].to(dev), 16,
                           inp["max_seq_len"], None, "auto", 1.0)
    out["paged_attention_v1"] = o
    kn = torch.randn(4, 2, 128, generator=g).to(torch.bfloat16).to(dev)
    vn = torch.randn(4, 2, 128, generator=g).to(torch.bfloat16).to(dev)
    slots = torch.tensor([3, 40, -1, 99], dtype=torch.int64, device=dev)
    ops.reshape_and_cache(kn, vn, kc, vc, slots, "auto", 1.0)
    out["reshape_and_cache"] = torch.stack([kc[:8].flatten(), vc[:8].flatten()])
    pr = helpers.make_w4a16_problem(0, 5, 256, 128, 4, 128, False, torch.bfloat16)
    ws = torch.zeros(128 // 64 * 16, dtype=torch.int32, device=dev)
    e = torch.empty(0, dtype=torch.int32, device=dev)
    qw = torch.randint(-2**31, 2**31 - 1, (256 // 8, 128), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
    out["gptq_marlin_repack"] = ops.gptq_marlin_repack(qw, e, 256, 128, 4)
    out["gptq_marlin_gemm"] = ops.gptq_marlin_gemm(pr["a"].to(dev), pr["marlin_q_w"].to(dev), pr["marlin_s"].to(dev), e, e,
                                                   ws, 4, 5, 128, 256, True)
    kcs, vcs = [kc.clone(), kc.clone() + 1], [vc.clone(), vc.clone() + 1]
    ops.copy_blocks(kcs, vcs, torch.tensor([[0, 5], [2, 7]], dtype=torch.int64, device=dev))
    out["copy_blocks"] = torch.stack([kcs[1][:8].flatten(), vcs[1][:8].flatten()])
    f8 = torch.empty(kc.shape, dtype=torch.uint8, device=dev)
    ops.convert_fp8(f8, kc, 0.5, "fp8")
    out["convert_fp8"] = f8[:8]
    out["meta_size"] = int(torch.ops._C_custom_ar.meta_size())
    xq, sc = ops.scaled_int8_quant(x)
    out["scaled_int8_quant"] = torch.cat([xq.float().flatten(), sc.flatten()])
    try:
        ops.rms_norm(y, x.cpu(), w, 1e-5)
        out["cpu_call"] = "no error"
    except (NotImplementedError, RuntimeError) as err:
        out["cpu_call"] = "RuntimeError" if isinstance(err, RuntimeError) else type(err).__name__
    try:
        ops.gptq_marlin_repack(qw, e, 256, 128, 5)
    except RuntimeError as err:
        out["bad_bits"] = "num_bits" in str(err)
    torch.cuda.synchronize()
    torch.save({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}, path)


if __name__ == "__main__":
    main(sys.argv[1])
