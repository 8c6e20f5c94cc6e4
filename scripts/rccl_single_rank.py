"""RCCL bring-up on a 1-GPU box: a process group of ONE rank over the nccl (= RCCL) backend, through this package's own wrappers.

    INC_MI355X_DIST_SINGLE_RANK=1 python scripts/rccl_single_rank.py

What runs for real: RCCL's communicator init with `device_id` (distributed.init_from_env), every collective CalibrationGroup wraps
(all_reduce, reduce, asynchronous broadcast + handle.wait, all_gather_into_tensor), a batch_isend_irecv round trip (self send / recv:
the call form of the layer mode's activation exchange), broadcast_object_list, and BOTH multi-GPU drivers end to end on the tiny
Llama of the tests with the group live (mode "layer": independent_setup / round / finish incl. the packed-block broadcasts; mode
"sample+rows": Hessian reduce, factor broadcast, row-sharded solve + all-gather) -- each compared bit for bit with the same mode
in a plain single process.  What it cannot show: more than one rank (RCCL refuses two ranks on one device).
"""
import os
import sys

os.environ.setdefault("INC_MI355X_DIST_SINGLE_RANK", "1")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from neural_compressor_amd import distributed as D  # noqa: E402
from tests.model_zoo import calib_ids, tiny_llama  # noqa: E402


def packed(model):
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear

    return {n: (m.qweight.cpu(), m.scales.cpu(), m.qzeros.cpu()) for n, m in model.named_modules() if isinstance(m, MI355XWeightOnlyLinear)}


def run_mode(mode, with_group):
    from neural_compressor_amd.torch.quantization import GPTQConfig, convert, prepare

    kw = {}
    if with_group:
        os.environ["INC_MI355X_GPTQ_MULTI_GPU"] = mode
    else:
        os.environ.pop("INC_MI355X_GPTQ_MULTI_GPU", None)
        if mode == "layer":
            kw["independent_blocks"] = True
    try:
        model = prepare(tiny_llama(layers=3), GPTQConfig(bits=4, group_size=32, block_size=128, use_sym=False), **kw)
        rq = model.quantizer.gptq_quantizer
        if with_group:
            assert (rq.layer_ctx if mode == "layer" else rq.dist_ctx) is not None, "the group is not live"
        for x in calib_ids():
            model(x)
        return packed(convert(model))
    finally:
        os.environ.pop("INC_MI355X_GPTQ_MULTI_GPU", None)


def main():
    assert torch.cuda.is_available()
    rank, world, local = D.init_from_env()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1 and D.live()
    dev = torch.device("cuda", local)
    print(f"process group: backend {dist.get_backend()}, world {dist.get_world_size()}, device {torch.cuda.get_device_name(dev)}")
    ctx = D.CalibrationGroup()
    g = torch.Generator().manual_seed(0)
    a = torch.randn(1 << 20, generator=g).to(dev)
    ref = a.clone()
    ctx.all_reduce(a)
    ctx.reduce(a, 0)
    h = ctx.broadcast(a, 0, async_op=True)
    if h is not None:
        h.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, ref), "one-rank all_reduce / reduce / broadcast must be the identity"
    rows = ctx.all_gather_rows(a.view(1024, 1024)[:1000], 1000, 1024)
    assert torch.equal(rows, ref.view(1024, 1024)[:1000])
    # the call form of the activation exchange: posted sends / receives, waited on later
    src = torch.arange(1 << 16, dtype=torch.float32, device=dev)
    dst = torch.zeros_like(src)
    works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, dst, 0), dist.P2POp(dist.isend, src, 0)])
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    assert torch.equal(src, dst), "self send / recv"
    meta = [dict(ok=True)]
    dist.broadcast_object_list(meta, src=0)
    print("collectives ok: all_reduce, reduce, async broadcast, all_gather_into_tensor, batch_isend_irecv, broadcast_object_list")
    for mode in ("layer", "sample+rows"):
        live = run_mode(mode, True)
        plain = run_mode(mode, False)
        assert live.keys() == plain.keys() and len(live) == 21
        if mode == "layer":
            for n in live:
                for x, y in zip(live[n], plain[n]):
                    assert torch.equal(x, y), (mode, n)
            print(f"mode {mode!r} over the RCCL group == the same mode in a plain process: {len(live)} packed modules bit-identical")
        else:
            # the owner's Hessian went through reduce(): H * n / n rounds once more than the plain process' H, so a rounding tie may flip
            same = [float((a[0] == b[0]).float().mean()) for a, b in zip(live.values(), plain.values())]
            assert min(same) >= 0.98, same
            print(f"mode {mode!r} over the RCCL group vs the plain process: packed words identical {min(same):.4f} .. {max(same):.4f} "
                  "(the reduced Hessian is H * n / n: one more rounding)")
    dist.barrier()
    dist.destroy_process_group()
    print("rccl_single_rank: OK")


if __name__ == "__main__":
    main()
