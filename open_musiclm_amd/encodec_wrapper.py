"""EncodecWrapper boundary stub (reference open_musiclm/encodec_wrapper.py): third-party pretrained codec, outside the hot path."""
from torch import nn


class EncodecWrapper(nn.Module):
    def __init__(self, *a, **k):
        raise ImportError("the `encodec` package / pretrained codec is not part of the MI355X hot-path build; the AR stack "
                          "works on token ids (use return_tokens=True / reconstruct_wave=False)")


def create_encodec_24khz(**kwargs):
    return EncodecWrapper(**kwargs)
