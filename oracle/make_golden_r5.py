"""Round-5 golden vectors, produced by running the REAL reference (/root/reference) on CPU (build container only):

    python oracle/make_golden_r5.py

  tests/golden/generate_eos.npz          reference TokenConditionedTransformerWrapper.generate (open_musiclm.py:253-326) with
                                         allow_eos_in_output=True and include_eos_in_output False / True on a tiny coarse model whose
                                         sampling is made near-uniform (filter_thres 0, temperature 8), so that eos ids ARE sampled on
                                         last-quantizer steps, embedded by the following steps (offset aliasing, :126-130) and masked by
                                         mask_out_after_eos_id (:321-322): every uniform draw recorded, oracle.generate checked bit-exact.
  tests/golden/musiclm_forward_prime.npz reference MusicLM.forward WITH prime_wave (audio continuation, open_musiclm.py:896-926) on
                                         tiny stages: stand-in wav2vec / codec whose inputs (the two prepare_audio results) and outputs
                                         (semantic / acoustic ids) are recorded, every stage.generate call and its draws recorded, final
                                         [coarse | fine] ids captured by the stand-in codec.  (Default strides: with a fine stride < 1 the
                                         reference's own final torch.cat of coarse and fine ids fails on unequal lengths, :1032.)
Test infrastructure: nothing here is imported by the product.
"""
from __future__ import annotations

import importlib
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import musiclm_oracle as O  # noqa: E402
from oracle.make_golden import OUT, import_reference, spec_of, to_np  # noqa: E402


def record_draws(ref_utils):
    """Replace utils.gumbel_noise (utils.py:73-75) by a recording twin; returns (list of draws, restore())."""
    uniforms = []
    orig = ref_utils.gumbel_noise

    def recording_noise(t):
        u = torch.zeros_like(t).uniform_(0, 1)
        uniforms.append(u.clone())
        return -ref_utils.log(-ref_utils.log(u))
    ref_utils.gumbel_noise = recording_noise

    def restore():
        ref_utils.gumbel_noise = orig
    return uniforms, restore


def make_generate_eos(ref):
    ref_utils = importlib.import_module("open_musiclm.utils")
    tiny = dict(dim=128, depth=2, heads=2, attn_dropout=0.0, ff_dropout=0.0)
    kwargs = dict(tiny, num_coarse_quantizers=3, clap_codebook_size=32, semantic_codebook_size=48, acoustic_codebook_size=6)
    B, steps, temperature, thres = 4, 6, 8.0, 0.0
    for seed in range(40, 80):
        torch.manual_seed(seed)
        model = ref.create_coarse_transformer(**kwargs)
        model.eval()
        spec = spec_of(model)
        ids = O.synthetic_ids(spec, B, [1, 7, 2], seed=seed)
        wrapper = ref.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
        outs, draws = {}, None
        for include in (False, True):
            uniforms, restore = record_draws(ref_utils)
            try:
                torch.manual_seed(1000 + seed)
                o = wrapper.generate(conditioning_token_ids=[t.clone() for t in ids[:-1]], max_time_steps=steps, filter_thres=thres,
                                     temperature=temperature, allow_eos_in_output=True, include_eos_in_output=include)
            finally:
                restore()
            outs[include] = o
            draws = torch.stack(uniforms)
        eos = spec.eos_ids[-1]
        kept = outs[True]
        rows_with_eos = (kept == eos).flatten(1).any(1)
        # wanted: some row samples an eos before the last time step (ids behind it are masked), another row never does
        first = [(int((r.flatten() == eos).nonzero()[0]) if e else -1) for r, e in zip(kept, rows_with_eos)]
        if rows_with_eos.any() and (~rows_with_eos).any() and any(0 <= f < 3 * (steps - 2) for f in first):
            break
    else:
        raise SystemExit("no seed produced the wanted eos pattern")
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    with torch.no_grad():
        for include in (False, True):
            o_ids = O.generate(sd, spec, ids[:-1], steps, draws, temperature=temperature, filter_thres=thres,
                               allow_eos_in_output=True, include_eos_in_output=include)
            assert torch.equal(o_ids, outs[include]), (include, o_ids, outs[include])
        # and the default flags on the same draws: eos never sampled
        torch.manual_seed(1000 + seed)
        uniforms, restore = record_draws(ref_utils)
        try:
            o_def = wrapper.generate(conditioning_token_ids=[t.clone() for t in ids[:-1]], max_time_steps=steps, filter_thres=thres,
                                     temperature=temperature)
        finally:
            restore()
        assert torch.equal(torch.stack(uniforms), draws)
        assert torch.equal(O.generate(sd, spec, ids[:-1], steps, draws, temperature=temperature, filter_thres=thres), o_def)
        assert not (o_def == eos).any()
    print(f"[generate_eos] seed {seed}: first eos per row {first}; kept-eos output\n{outs[True].flatten(1)}\nmasked output\n{outs[False].flatten(1)}")
    out = {"sd." + k: v for k, v in to_np(sd).items()}
    for i, t in enumerate(ids[:-1]):
        out[f"cond.{i}"] = t.numpy()
    out["uniforms"] = draws.numpy()
    out["generated_allow"] = outs[False].numpy()
    out["generated_allow_include"] = outs[True].numpy()
    out["generated_default"] = o_def.numpy()
    out["temperature"], out["filter_thres"], out["max_time_steps"] = np.float64(temperature), np.float64(thres), np.int64(steps)
    out["meta.kwargs"] = np.array(repr(kwargs))
    np.savez_compressed(os.path.join(OUT, "generate_eos.npz"), **out)


def make_musiclm_forward_prime(ref):
    ref_utils = importlib.import_module("open_musiclm.utils")
    torch.manual_seed(31)
    tiny = dict(dim=128, depth=2, heads=2, attn_dropout=0.0, ff_dropout=0.0)
    cb = dict(clap_codebook_size=32, semantic_codebook_size=48, acoustic_codebook_size=40)
    sem = ref.create_semantic_transformer(**tiny, clap_codebook_size=32, semantic_codebook_size=48)
    coarse = ref.create_coarse_transformer(**tiny, num_coarse_quantizers=3, **cb)
    fine = ref.create_fine_transformer(**tiny, num_coarse_quantizers=3, num_fine_quantizers=5,
                                       clap_codebook_size=32, acoustic_codebook_size=40)
    B, rate = 1, 24                                  # one clip (prepare_audio folds channels, utils.py:158-159); every sample rate 24 Hz
    clap_ids = torch.randint(0, 32, (B, 12, 1), generator=torch.Generator().manual_seed(4))
    g = torch.Generator().manual_seed(8)
    kw = dict(output_seconds=4, semantic_window_seconds=2, coarse_window_seconds=2, fine_window_seconds=1,
              semantic_steps_per_second=6, acoustic_steps_per_second=4)
    prime_wave = torch.randn(2, 3 * rate, generator=g) * 0.3          # stereo, 3 s: longer than the 2 s semantic window (truncated :162-163)
    sem_prime = torch.randint(0, 48, (B, 2 * 6 - 1), generator=g)     # what the stand-in wav2vec "hears" (MERT: 50 s - 1 ids)
    ac_prime = torch.randint(0, 40, (B, 2 * 4, 8), generator=g)       # stand-in codec: 8 codebooks = 3 coarse + 5 fine
    seen, captured = {}, {}

    class Clap:
        def __call__(self, *, text_input=None, audio_input=None, **k):
            return clap_ids.clone()

    class Wav2vec:
        target_sample_hz = rate
        codebook_size = 48                           # SemanticStage / CoarseStage assert it against the transformer (:534, :628)
        def __call__(self, wav, flatten=False, **k):
            seen["wav2vec_in"] = wav.clone()
            return sem_prime.clone()

    class Codec:
        sample_rate = rate
        def eval(self):
            return self
        def __call__(self, wav, return_encoded=True, **k):
            seen["codec_in"] = wav.clone()
            return None, ac_prime.clone(), None
        def decode_from_codebook_indices(self, ids):
            captured["acoustic"] = ids.clone()
            return torch.zeros(ids.shape[0], 1, 8)

    mlm = ref.MusicLM(wav2vec=Wav2vec(), clap=Clap(), neural_codec=Codec(), semantic_transformer=sem, coarse_transformer=coarse,
                      fine_transformer=fine)
    uniforms, restore = record_draws(ref_utils)
    calls = []
    for name in ("semantic", "coarse", "fine"):
        stage = getattr(mlm, name)
        def wrap(fn, name=name):
            def inner(*a, **k):
                first = len(uniforms)
                out = fn(*a, **k)
                calls.append((name, first, len(uniforms), out.clone()))
                return out
            return inner
        stage.generate = wrap(stage.generate)
    try:
        torch.manual_seed(77)
        mlm(text=["x"] * B, prime_wave=prime_wave.clone(), prime_wave_sample_hz=rate, **kw)
    finally:
        restore()
    out = {}
    for pfx, m in (("sem", sem), ("coarse", coarse), ("fine", fine)):
        out.update({f"sd.{pfx}." + k: v for k, v in to_np(m.state_dict()).items()})
    out["clap_ids"], out["prime_wave"], out["rate"] = clap_ids.numpy(), prime_wave.numpy(), np.int64(rate)
    out["sem_prime"], out["ac_prime"] = sem_prime.numpy(), ac_prime.numpy()
    out["wav2vec_in"], out["codec_in"] = seen["wav2vec_in"].numpy(), seen["codec_in"].numpy()
    out["acoustic"] = captured["acoustic"].numpy()
    out["n_calls"] = np.int64(len(calls))
    for i, (name, a, b, ids) in enumerate(calls):
        out[f"call.{i}.stage"] = np.array(name)
        out[f"call.{i}.ids"] = ids.numpy()
        out[f"call.{i}.uniforms"] = torch.stack(uniforms[a:b]).numpy() if b > a else np.zeros((0, B, 1), np.float32)
    out["meta.kwargs"], out["meta.tiny"] = np.array(repr(kw)), np.array(repr(tiny))
    np.savez_compressed(os.path.join(OUT, "musiclm_forward_prime.npz"), **out)
    print(f"[musiclm_forward_prime] {len(calls)} stage.generate calls ({[(c[0], tuple(c[3].shape)) for c in calls]}), {len(uniforms)} draws, "
          f"prepared audio {tuple(seen['wav2vec_in'].shape)} / {tuple(seen['codec_in'].shape)}, acoustic ids {tuple(captured['acoustic'].shape)}")


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    make_generate_eos(ref)
    make_musiclm_forward_prime(ref)


if __name__ == "__main__":
    main()
