"""EncodecWrapper (reference open_musiclm/encodec_wrapper.py:12-71).  The 24 kHz Encodec network is a pretrained third-party
model (package `encodec`, weights fetched from the network) and is not part of this build; the wrapper itself -- the boundary the
stages and `MusicLM.forward` talk to -- is: it takes ANY object with Encodec's interface (`sample_rate`, `bandwidth`,
`quantizer.n_q`, `quantizer.bins`, `encode(x) -> [(codes [B, n_q, T], scale)]`, `decode(frames)`), turns waveforms into
`[B, T, n_q]` codebook indices and indices back into a waveform exactly as the reference does."""
import torch
from torch import nn

from .utils import exists


class EncodecWrapper(nn.Module):
    def __init__(self, *, encodec, output_hz: int = 75):
        super().__init__()
        self.encodec = encodec
        self.sample_rate = encodec.sample_rate
        self.output_hz = output_hz
        assert exists(encodec.bandwidth)
        total_quantizers = encodec.quantizer.n_q
        self.num_quantizers = int(encodec.bandwidth / 24 * total_quantizers)      # output quantizers per frame
        self.codebook_size = encodec.quantizer.bins

    def forward(self, x: torch.Tensor, return_encoded=True, **kwargs):
        assert return_encoded == True
        if x.dim() == 2:
            x = x.unsqueeze(1)                                                     # 'b t -> b 1 t': the mono dimension
        with torch.no_grad():
            if hasattr(self.encodec, "eval"):
                self.encodec.eval()
            encoded_frames = self.encodec.encode(x)
        codes = torch.cat([encoded[0] for encoded in encoded_frames], dim=-1)      # [B, n_q, T]
        return None, codes.transpose(1, 2), None                                   # [B, T, n_q]

    def decode_from_codebook_indices(self, quantized_indices):
        """quantized_indices [B, T, n_q] -> wave."""
        frames = [(quantized_indices.transpose(1, 2), None)]                       # one frame, as in the reference
        with torch.no_grad():
            if hasattr(self.encodec, "eval"):
                self.encodec.eval()
            return self.encodec.decode(frames)


def create_encodec_24khz(bandwidth: float = 6.0, codebook_size: int = 1024, **kwargs):
    assert bandwidth in [1.5, 3., 6., 12., 24.], "invalid bandwidth. must be one of [1.5, 3., 6., 12., 24.]"
    try:
        from encodec import EncodecModel
    except ImportError as e:
        raise ImportError("the `encodec` package (pretrained neural codec) is not part of the MI355X hot-path build: pass your own "
                          "model to EncodecWrapper(encodec=...), or stay on token ids (return_tokens / reconstruct_wave=False)") from e
    encodec = EncodecModel.encodec_model_24khz()
    encodec.set_target_bandwidth(bandwidth)
    wrapper = EncodecWrapper(encodec=encodec, **kwargs)
    assert wrapper.codebook_size == codebook_size, "encodec codebook size must be 1024 for now"
    return wrapper
