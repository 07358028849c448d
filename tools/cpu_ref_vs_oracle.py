"""Calibration of bench.py's cpu_baseline (kind "port"): the REAL reference (/root/reference, imported through the test shim of
oracle/make_golden.py) and the oracle port timed side by side on THIS container's host cores -- the same micro-step bench.py's CPU leg
runs (coarse-small, B = 2, N = 1116, fp32: forward + backward + global-norm clip + AdamW; 1 warm-up + median of 3).  Build container only
(/root/reference does not exist on the GPU box); writes profiles/r05_cpu_reference_vs_oracle.md.  Test infrastructure."""
import os, statistics, sys, time
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import musiclm_oracle as O
from oracle.make_golden import import_reference

N_SEQ, CB = 1116, 2


def timed(step, n=4):
    ts = []
    for it in range(n):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
        print(f"   iteration {it}: {ts[-1]:.2f} s", flush=True)
    return ts


def main():
    ref = import_reference()
    spec = O.coarse_spec(dim=1024, depth=6, heads=8)
    ids = O.synthetic_ids(spec, CB, [1, 199, 300], seed=1234)
    # ---- the reference's own modules (open_musiclm.py:432 create_coarse_transformer, :219 wrapper, optimizer.py:10 get_optimizer, trainer.py:439-447)
    torch.manual_seed(0)
    model = ref.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, attn_dropout=0.0, ff_dropout=0.1)
    wrapper = ref.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False, cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.15)
    wrapper.train()
    import importlib
    ref_opt = importlib.import_module("open_musiclm.optimizer")
    opt = ref_opt.get_optimizer(model.parameters(), lr=3e-4, wd=0.01)

    def ref_step():
        opt.zero_grad(set_to_none=True)
        loss, _, _ = wrapper(all_token_ids=[t.clone() for t in ids], return_loss=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
    print(f"reference ({torch.get_num_threads()} torch threads):", flush=True)
    t_ref = timed(ref_step)
    del model, wrapper, opt
    # ---- the oracle, exactly as bench.cpu_baseline drives it
    sd = O.init_state_dict(spec, seed=0)
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith("beta")]
    oopt = torch.optim.AdamW(params, lr=3e-4, betas=(0.9, 0.99), weight_decay=0.01)
    noise = torch.randn(CB, N_SEQ, generator=torch.Generator().manual_seed(1))

    def oracle_step():
        oopt.zero_grad(set_to_none=True)
        l, _, _ = O.wrapper_forward_loss(sd, spec, ids, [0., 0., 1.], forget_noise=noise)
        l.backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 0.5)
        oopt.step()
    print("oracle:", flush=True)
    t_orc = timed(oracle_step)
    m_ref, m_orc = statistics.median(t_ref[1:]), statistics.median(t_orc[1:])
    out = os.path.join(ROOT, "profiles", "r05_cpu_reference_vs_oracle.md")
    with open(out, "w") as fh:
        fh.write("# cpu_baseline calibration: the real reference vs the oracle port, same host, same micro-step (tools/cpu_ref_vs_oracle.py)\n\n"
                 f"Build container, {torch.get_num_threads()} torch threads ({os.cpu_count()} hardware threads), torch {torch.__version__}; coarse-small, B = {CB}, N = {N_SEQ}, fp32, "
                 "forward + backward + clip_grad_norm_(0.5) + AdamW; 1 warm-up + median of 3.  The reference runs with its own dropout (0.1) and forgetful mask.\n\n"
                 "| | seconds per micro-step (4 runs) | median of the last 3 | samples/s |\n|---|---|---:|---:|\n"
                 f"| reference (`/root/reference/open_musiclm`, its wrapper + `get_optimizer`) | {', '.join(f'{t:.2f}' for t in t_ref)} | {m_ref:.2f} | {CB / m_ref:.3f} |\n"
                 f"| oracle (`oracle/musiclm_oracle.py`, what `bench.py` times as `cpu_baseline.kind = port`) | {', '.join(f'{t:.2f}' for t in t_orc)} | {m_orc:.2f} | {CB / m_orc:.3f} |\n\n"
                 f"oracle / reference time ratio: **{m_orc / m_ref:.3f}** -- multiply the GPU box's `cpu_baseline.value` (the port) by this to estimate the reference's own rate there.\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
