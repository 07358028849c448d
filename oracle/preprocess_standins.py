"""TEST INFRASTRUCTURE (oracle/): deterministic integer stand-ins for the three pretrained tokenizers the DataPreprocessor drives
(ClapQuantized, HfHubertWithKmeans, EncodecWrapper), with the reference's call signatures (open_musiclm.py:476-510), and the seeded
"audio" batches they are fed.  Used by oracle/make_golden_r2.py (through the REFERENCE's DataPreprocessor.process) and by
tests/test_host_logic.py (through ours): both must write the same sqlite token store byte for byte."""
import numpy as np
import torch

CLAP_SR, SEM_SR, AC_SR = 8, 10, 12          # "sample rates" of the three resampled waves (tiny: the window arithmetic is rate-free)
SEM_HZ, AC_HZ, NQ = 5, 3, 8                  # token rates and Encodec quantizers
WINDOW_S = 4                                 # clap == semantic window (seconds)


class Clap:
    sample_rate = CLAP_SR

    def __init__(self):
        self.calls = []

    def __call__(self, *, text_input=None, audio_input=None, **kw):
        assert text_input is None and audio_input.dim() == 2 and audio_input.shape[1] == CLAP_SR * WINDOW_S
        self.calls.append(audio_input.shape[0])
        s = audio_input.long().sum(-1, keepdim=True)                                   # [b, 1]
        q = torch.arange(12)
        return ((s * (q + 3) + q * q) % 1024).unsqueeze(-1)                            # [b, 12, 1]


class Wav2Vec:
    def __call__(self, wave, flatten=True, **kw):
        assert not flatten and wave.dim() == 2 and wave.shape[0] == 1
        frames = wave.long().reshape(1, -1, SEM_SR // SEM_HZ).sum(-1)
        return ((frames * 7 + 1) % 1024)[:, :-1]                                       # [1, secs * SEM_HZ - 1]


class Codec(torch.nn.Module):
    def forward(self, wave, return_encoded=False, **kw):
        assert return_encoded and wave.dim() == 2 and wave.shape[0] == 1
        frames = wave.long().reshape(1, -1, AC_SR // AC_HZ).sum(-1)                    # [1, secs * AC_HZ]
        q = torch.arange(NQ)
        return None, (frames.unsqueeze(-1) * (2 * q + 1) + 31 * q) % 1024, None        # [1, T, NQ]


def batches(seed=11, seconds=(6, 9, 4, 12, 5)):
    """What SoundDatasetForPreprocessing + its collate yield (data.py:170-301): one file per batch, three resampled waves."""
    rng = np.random.RandomState(seed)
    out = []
    for i, s in enumerate(seconds):
        waves = tuple(torch.from_numpy(rng.randint(0, 200, (1, s * sr)).astype(np.float32)) for sr in (CLAP_SR, SEM_SR, AC_SR))
        out.append({"data": waves, "file_path": [f"/audio/{i:03d}.mp3"]})
    return out
