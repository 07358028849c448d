"""Worker of tests/test_gpu_model.py::test_data_parallel_step_equals_single_process_accumulation.

mode "dp":     launched twice (RANK 0 / 1, both on cuda:0, gloo rendezvous on 127.0.0.1): every rank runs ONE micro-batch
               per optimizer step through SingleStageTrainer.micro_step / optimizer_step (flat SUM all-reduce, 1/world folded
               into the fused AdamW, clip after the exchange -- trainer.py:428-447 of the reference).
mode "single": one process, the same two micro-batches per step accumulated with grad_accum_every = 2.
Rank 1 builds its model from a DIFFERENT seed: the trainer's start-up broadcast must make the replicas identical.
Writes {flat gradient of the last step (after exchange and scaling), parameters after the last step} to argv[2]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)          # a hung collective / kernel: say where, then die (the test kills the peer)
    mode, out_path = sys.argv[1], sys.argv[2]
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.data import SyntheticTokenDataset
    from open_musiclm_amd.trainer import SingleStageTrainer
    rank = int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda:0")
    torch.manual_seed(0 if rank == 0 else 12345)
    model = M.create_coarse_transformer(dim=128, depth=2, heads=2, num_coarse_quantizers=3, ff_dropout=0.0,
                                        precision="bf16x3").to(dev)
    ds = SyntheticTokenDataset("coarse", length=64, coarse_window_seconds=1, semantic_window_seconds=2)
    tr = SingleStageTrainer(model, "coarse", num_train_steps=10, batch_size=2, dataset=ds, lr=1e-3, lr_warmup=0,
                            grad_accum_every=2 if mode == "single" else 1, wd=0.01, max_grad_norm=0.5, valid_frac=0.0,
                            save_results_every=10 ** 6, save_model_every=10 ** 6, results_folder=out_path + ".res",
                            save_predicted_tokens=False, save_reconstructed_wave=False, use_hip_graph=False,
                            accelerate_kwargs={})
    tr.train_wrapper.transformer_wrapper.mask_prob = 0.0          # deterministic key mask
    tr.transformer.train(); tr.train_wrapper.train()

    def batch(i):          # micro-batch i of the run: samples 2i, 2i+1
        items = [ds[2 * i], ds[2 * i + 1]]
        return {k: torch.cat([it[f] for it in items], 0).to(dev) for f, k in enumerate(tr.ds_fields)}
    gsnap = None
    for step in range(2):
        tr.optim.zero_grad()
        if mode == "single":
            for a in range(2):
                tr.micro_step(batch(2 * step + a))
        else:
            tr.micro_step(batch(2 * step + rank))
        if step == 1:          # the gradient the optimizer is about to consume
            g = tr.optim.flat_grad.clone()
            tr.dp.allreduce_sum_(g)
            gsnap = (g * tr.dp.grad_scale()).cpu()
        tr.optimizer_step()
    torch.cuda.synchronize()
    torch.save(dict(grad=gsnap, param=tr.optim.flat_param.detach().cpu(), world=tr.dp.world_size), out_path)
    tr.dp.barrier()
    tr.dp.shutdown()


if __name__ == "__main__":
    main()
