"""Worker of tests/test_gpu_model.py::test_rccl_allreduce_of_the_flat_gradient_buffer_after_a_graph_replay.

ONE rank, backend "nccl" (= RCCL on ROCm; it initialises with world_size 1): what the 1-GPU lease can show about the data-parallel
exchange that the gloo tests cannot -- the RCCL library loads under HSA_ENABLE_IPC_MODE_LEGACY=0, creates a communicator on this GPU,
and a SUM all-reduce of the optimizer's flat fp32 gradient buffer (coarse-small: 91.6 M floats = 366.6 MB, the buffer the product
exchanges) issued on the CURRENT stream right after a replay of the captured micro-step (a) is ordered behind the graph's last kernel
(the grouped weight-gradient launch) and (b) returns the buffer unchanged (sum over one rank), after which the fused optimizer step runs.
Writes a small report to argv[1]."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)
    import torch.distributed as dist
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.graph import GraphedForwardBackward
    from open_musiclm_amd.optimizer import get_optimizer
    from open_musiclm_amd.parallel import DataParallel
    from oracle import musiclm_oracle as O                      # test infrastructure: synthetic ids only
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0", "the package import must have exported dmabuf IPC mode"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    t0 = time.time()
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    dp = DataParallel(device=dev)                               # adopts the initialised group
    torch.manual_seed(0)
    precision = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, ff_dropout=0.1, precision=precision).to(dev)
    stage = M.CoarseStage(coarse_transformer=model, cross_entropy_loss_weights=[0., 0., 1.])
    stage.train()
    optim = get_optimizer(model.parameters(), lr=3e-4, wd=0.01)
    spec = O.coarse_spec(dim=1024, depth=6, heads=8)
    ids = [t.to(dev) for t in O.synthetic_ids(spec, 4, [1, 199, 300], seed=3)]
    keys = ("clap_token_ids", "semantic_token_ids", "coarse_token_ids")
    fb = GraphedForwardBackward(lambda **kw: stage(**kw, return_loss=True, return_logits=False)[0])
    optim.zero_grad()

    def discard():
        optim.mark_grads_dirty()
        optim.zero_grad()
    fb.prepare(dict(zip(keys, ids)), after_warmup=discard)
    assert fb.graph is not None, fb.capture_error
    flat = optim.flat_grad
    assert flat.numel() * 4 > 360e6, flat.numel()
    rep = dict(init_s=round(time.time() - t0, 2), flat_mb=round(flat.numel() * 4 / 1e6, 1), precision=precision, equal=[], ms=[])
    os.environ["OMLM_DP_BUCKET_MB"] = "64"
    for it in range(3):
        optim.zero_grad()
        loss = fb(**dict(zip(keys, ids)))                       # graph replay: the flat buffer is written by kernels still in flight
        optim.mark_grads_dirty()
        before = flat.clone()                                   # same stream, behind the replay
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if it == 2:                                             # the bucketed form: six asynchronous 64 MB all-reduces behind the same replay
            nb = dp.bucket_count(flat)
            assert nb >= 5, nb
            rep["buckets"] = nb
            dp.allreduce_buckets_(flat, nb)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)         # exactly DataParallel.allreduce_sum_'s single-collective call
        e1.record()
        after = flat.clone()
        optim.step(max_grad_norm=0.5, grad_scale=dp.grad_scale())
        torch.cuda.synchronize()
        assert float(before.abs().max()) > 0, "the replay left no gradient"
        rep["equal"].append(bool(torch.equal(before, after)))
        rep["ms"].append(round(e0.elapsed_time(e1), 3))
        rep["loss"] = float(loss)
    # the validation-side collectives of trainer.py:470-473 through the same communicator
    m = dp.reduce_mean(torch.tensor([2.5], device=dev))
    g = dp.all_gather_cat(torch.arange(6, device=dev).view(2, 3))
    rep["reduce_mean"], rep["gather_shape"] = float(m), list(g.shape)
    rep["grad_norm_sq"] = float(optim.last_grad_norm_sq)
    json.dump(rep, open(sys.argv[1], "w"))
    dist.barrier()
    dist.destroy_process_group()
    print("rccl worker ok", rep, flush=True)


if __name__ == "__main__":
    main()
