"""ClapQuantized (reference open_musiclm/clap_quantized.py).  The frozen CLAP towers are third-party pretrained
networks outside the hot path; what IS on the path is ``quantize``: the residual-VQ nearest-codeword chain,
which runs as one HIP kernel (omlm_rvq_encode) and is bit-exact against the stated definition in
oracle/musiclm_oracle.py::rvq_encode (RVQ parity is unpinned against the un-vendored vector-quantize-pytorch)."""
from __future__ import annotations

from typing import List, Optional, Union

import torch
from torch import nn

from . import ops
from .utils import exists


class ResidualVQCodebooks(nn.Module):
    """Inference-side stand-in for vector_quantize_pytorch.ResidualVQ: holds the codebooks under the
    library's checkpoint keys (layers.{s}._codebook.embed [1, C, D]) and encodes with the HIP kernel."""

    def __init__(self, *, dim, num_quantizers, codebook_size):
        super().__init__()
        self.dim, self.num_quantizers, self.codebook_size = dim, num_quantizers, codebook_size
        self.register_buffer("codebooks", torch.zeros(num_quantizers, codebook_size, dim))
        self._cbT = None

    def load_state_dict(self, sd, strict=True):
        if "codebooks" in sd:
            return super().load_state_dict(sd, strict=strict)
        for s in range(self.num_quantizers):                      # vector-quantize-pytorch layout
            self.codebooks[s].copy_(sd[f"layers.{s}._codebook.embed"].reshape(self.codebook_size, self.dim))
        self._cbT = None
        return None

    def _transposed(self):
        if self._cbT is None or self._cbT.device != self.codebooks.device:
            self._cbT = self.codebooks.transpose(1, 2).contiguous()        # [S, D, C] for coalesced reads
        return self._cbT

    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """x [n, D] fp32 (cuda) -> indices [n, S] int64."""
        x = x.contiguous().float()
        n = x.shape[0]
        idx = torch.empty(n, self.num_quantizers, dtype=torch.int32, device=x.device)
        ops.rvq_encode(x, self._transposed(), idx, None, n, self.dim, self.codebook_size, self.num_quantizers)
        return idx.long()


class ClapQuantized(nn.Module):
    def __init__(self, *, clap=None, codebook_size: int = 1024, rq_num_quantizers: int = 12, rq_ema_decay: float = 0.95,
                 learn_rvq: bool = False, threshold_ema_dead_code: float = 0.0, embed_dim: Optional[int] = None):
        super().__init__()
        if learn_rvq:
            raise NotImplementedError("RVQ fitting (EMA k-means) is outside the hot path; load a trained codebook")
        self.clap, self.codebook_size, self.learn_rvq = clap, codebook_size, learn_rvq
        self.sample_rate = clap.model_cfg['audio_cfg']['sample_rate'] if exists(clap) else 48000
        dim = embed_dim if exists(embed_dim) else (clap.model.joint_embed_shape if exists(clap) else 512)
        self.rq = ResidualVQCodebooks(dim=dim, num_quantizers=rq_num_quantizers, codebook_size=codebook_size)

    def forward(self, *, audio_input=None, text_input: Optional[List[str]] = None, return_embedding=False,
                return_rvq_loss=False):
        assert exists(audio_input) ^ exists(text_input), "either audio or text must be provided, but not both"
        if not exists(self.clap):
            raise RuntimeError("ClapQuantized was built without CLAP towers; call quantize(embedding) directly")
        with torch.no_grad():
            self.clap.eval()
            emb = self.clap.get_audio_embedding_from_data(audio_input) if exists(audio_input) else self.clap.get_text_embedding(text_input)
        return emb if return_embedding else self.quantize(emb, return_rvq_loss=return_rvq_loss)

    def quantize(self, embedding, return_rvq_loss=False):
        """clap_quantized.py:75-87: [n, D] -> indices [n, num_quantizers, 1]."""
        idx = self.rq.encode(embedding)
        if return_rvq_loss:
            q = torch.stack([self.rq.codebooks[s][idx[:, s]] for s in range(idx.shape[1])]).sum(0)
            return torch.nn.functional.mse_loss(q, embedding).item()
        return idx.unsqueeze(-1)


def create_clap_quantized(device=None, learn_rvq=False, enable_fusion=False, rvq_checkpoint_path=None,
                          checkpoint_path: Optional[str] = None, amodel_type: str = 'HTSAT-tiny', **kwargs):
    try:
        from .laion_clap import CLAP_Module          # vendored in the reference; not part of this hot-path build
    except ImportError as e:
        raise ImportError("the LAION-CLAP towers (pretrained, third-party) are not part of the MI355X hot-path build; "
                          "construct ClapQuantized(clap=None, ...) and feed embeddings to .quantize(), or pass clap_token_ids") from e
    clap = CLAP_Module(enable_fusion=enable_fusion, device=device, amodel=amodel_type)
    clap.load_ckpt(ckpt=checkpoint_path)
    cq = ClapQuantized(clap=clap, learn_rvq=learn_rvq, **kwargs)
    if exists(rvq_checkpoint_path):
        cq.rq.load_state_dict(torch.load(rvq_checkpoint_path, map_location=device))
    return cq
