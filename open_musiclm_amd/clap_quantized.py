"""ClapQuantized (reference open_musiclm/clap_quantized.py).  The frozen CLAP towers are third-party pretrained
networks outside the hot path; what IS on the path is ``quantize``: the residual-VQ nearest-codeword chain,
which runs as one HIP kernel (omlm_rvq_encode) in the distance form of vector-quantize-pytorch's EuclideanCodebook (-cdist,
first maximum), bit-exact against oracle/musiclm_oracle.py::rvq_encode and -- on exactly representable inputs, where the
summation order of a BLAS cannot matter -- against torch.cdist itself (the library is un-vendored; tests pin the form).
``learn_rvq=True`` fits the codebooks (what scripts/train_clap_rvq.py drives through ClapRVQTrainer) with the kernels of
csrc/vq_fit.hip, checked against oracle.rvq_fit_step."""
from __future__ import annotations

from typing import List, Optional, Union

import torch
from torch import nn

from . import ops
from .hip import require_gpu
from .utils import exists


class ResidualVQCodebooks(nn.Module):
    """Stand-in for vector_quantize_pytorch.ResidualVQ (un-vendored; distance form pinned against torch.cdist, fit step restated in
    oracle.rvq_fit_step): holds
    the codebooks and their EMA statistics, reads / writes the library's checkpoint keys
    (layers.{s}._codebook.{initted, cluster_size [1, K], embed [1, K, D], embed_avg [1, K, D]}), encodes with the HIP
    nearest-codeword kernel and -- in training mode -- runs the library's fit step (k-means init on the first batch, EMA
    codebook update, dead-code re-seeding) on the device (csrc/vq_fit.hip)."""

    def __init__(self, *, dim, num_quantizers, codebook_size, decay: float = 0.95, eps: float = 1e-5, kmeans_iters: int = 10,
                 threshold_ema_dead_code: float = 0.0):
        super().__init__()
        self.dim, self.num_quantizers, self.codebook_size = dim, num_quantizers, codebook_size
        self.decay, self.eps, self.kmeans_iters, self.threshold_ema_dead_code = decay, eps, kmeans_iters, threshold_ema_dead_code
        self.register_buffer("codebooks", torch.zeros(num_quantizers, codebook_size, dim))
        self.register_buffer("embed_avg", torch.zeros(num_quantizers, codebook_size, dim))
        self.register_buffer("cluster_size", torch.zeros(num_quantizers, codebook_size))
        self.register_buffer("initted", torch.zeros(num_quantizers, dtype=torch.bool))
        self._cbT = None
        # test hooks: (n, K) -> LongTensor [K] of row indices (initial k-means means / dead-code re-seeds); default: device RNG
        self.init_pick_source = None
        self.expire_pick_source = None

    # ---- checkpoints in the library's layout (what trainer.py:731 saves and clap_quantized.py:109 loads) -----------------
    # The key translation lives in the nn.Module hooks (_save_to_state_dict / _load_from_state_dict), so an enclosing module's
    # state_dict() / load_state_dict() round-trips too (ClapQuantized.state_dict()["rq.layers.0._codebook.embed"], strict load).
    def _lib_prefix(self, prefix, s):
        return f"{prefix}layers.{s}._codebook."

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for s in range(self.num_quantizers):
            p = self._lib_prefix(prefix, s)
            destination[p + "initted"] = self.initted[s:s + 1].detach().clone()
            destination[p + "cluster_size"] = self.cluster_size[s:s + 1].detach().clone()
            destination[p + "embed"] = self.codebooks[s:s + 1].detach().clone()
            destination[p + "embed_avg"] = self.embed_avg[s:s + 1].detach().clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        own = ("codebooks", "embed_avg", "cluster_size", "initted")
        with torch.no_grad():
            if prefix + "codebooks" in state_dict:                   # this module's own buffer names (round-1 checkpoints)
                for k in own:
                    if prefix + k in state_dict:
                        getattr(self, k).copy_(state_dict[prefix + k])
                if prefix + "initted" not in state_dict:
                    self.initted.fill_(True)
            else:                                                    # vector-quantize-pytorch layout
                for s in range(self.num_quantizers):
                    p = self._lib_prefix(prefix, s)
                    if p + "embed" not in state_dict:
                        missing_keys.append(p + "embed")
                        continue
                    self.codebooks[s].copy_(state_dict[p + "embed"].reshape(self.codebook_size, self.dim))
                    if p + "embed_avg" in state_dict:
                        self.embed_avg[s].copy_(state_dict[p + "embed_avg"].reshape(self.codebook_size, self.dim))
                    if p + "cluster_size" in state_dict:
                        self.cluster_size[s].copy_(state_dict[p + "cluster_size"].reshape(self.codebook_size))
                    self.initted[s] = bool(state_dict[p + "initted"].reshape(-1)[0]) if p + "initted" in state_dict else True
        known = {prefix + k for k in own}
        known |= {self._lib_prefix(prefix, s) + k for s in range(self.num_quantizers)
                  for k in ("initted", "cluster_size", "embed", "embed_avg")}
        if strict:
            unexpected_keys.extend(k for k in state_dict if k.startswith(prefix) and k not in known)
        self._cbT = None

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

    def _picks(self, source, n: int, device) -> torch.Tensor:
        K = self.codebook_size
        if source is not None:
            return source(n, K).to(device).long()
        if n >= K:
            return torch.randperm(n, device=device)[:K]
        return torch.randint(0, n, (K,), device=device)

    @torch.no_grad()
    def fit_step(self, x: torch.Tensor):
        """Training-mode forward of the library (ClapQuantized.quantize with learn_rvq, clap_quantized.py:75-84): returns
        (indices [n, S] int64, quantized sum [n, D]) and updates codebooks / EMA statistics in place."""
        x = x.contiguous().float()
        require_gpu(x, "RVQ embeddings")
        n, D, K, S = x.shape[0], self.dim, self.codebook_size, self.num_quantizers
        dev = x.device
        idx = torch.empty(n, S, dtype=torch.int32, device=dev)
        r = x.clone()
        r_next = torch.empty_like(r)
        counts = torch.empty(K, device=dev)
        sums = torch.empty(K, D, device=dev)
        total = torch.empty(1, device=dev)
        cbT = self._transposed().clone()                               # [S, D, K]; kept in step with the codebooks below
        one = torch.empty(n, dtype=torch.int32, device=dev)

        def assign_accumulate(codes_T, out_idx, stride, resid):
            ops.rvq_encode(r, codes_T, out_idx, resid, n, D, K, 1, idx_stride=stride)
            counts.zero_(); sums.zero_()
            ops.vq_accumulate(r, out_idx, stride, counts, sums, n, D, K)

        for s in range(S):
            if not bool(self.initted[s]):
                self.codebooks[s].copy_(r[self._picks(self.init_pick_source, n, dev)])
                cbT[s].copy_(self.codebooks[s].t())
                for _ in range(self.kmeans_iters):
                    assign_accumulate(cbT[s], one, 1, None)
                    ops.vq_kmeans_update(self.codebooks[s], cbT[s], counts, sums, K, D)
                self.cluster_size[s].copy_(counts)                     # the library's `bins`: bucket sizes of the last iteration
                self.embed_avg[s].copy_(self.codebooks[s] * counts[:, None])
                self.initted[s] = True
            assign_accumulate(cbT[s], idx[:, s], S, r_next)            # r_next = r - e_idx with the codes BEFORE the update
            ops.vq_ema_update(self.cluster_size[s], self.embed_avg[s], self.codebooks[s], cbT[s], counts, sums, total, K, D,
                              self.decay, self.eps)
            if self.threshold_ema_dead_code > 0:
                dead = self.cluster_size[s] < self.threshold_ema_dead_code
                if bool(dead.any()):
                    samp = r[self._picks(self.expire_pick_source, n, dev)]
                    self.codebooks[s][dead] = samp[dead]
                    self.cluster_size[s][dead] = self.threshold_ema_dead_code
                    self.embed_avg[s][dead] = samp[dead] * self.threshold_ema_dead_code
                    cbT[s].copy_(self.codebooks[s].t())
            r, r_next = r_next, r
        self._cbT = None
        return idx.long(), x - r


class ClapQuantized(nn.Module):
    def __init__(self, *, clap=None, codebook_size: int = 1024, rq_num_quantizers: int = 12, rq_ema_decay: float = 0.95,
                 learn_rvq: bool = False, threshold_ema_dead_code: float = 0.0, embed_dim: Optional[int] = None):
        super().__init__()
        self.clap, self.codebook_size, self.learn_rvq = clap, codebook_size, learn_rvq
        self.sample_rate = clap.model_cfg['audio_cfg']['sample_rate'] if exists(clap) else 48000
        dim = embed_dim if exists(embed_dim) else (clap.model.joint_embed_shape if exists(clap) else 512)
        self.rq = ResidualVQCodebooks(dim=dim, num_quantizers=rq_num_quantizers, codebook_size=codebook_size, decay=rq_ema_decay,
                                      threshold_ema_dead_code=threshold_ema_dead_code)

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
        """clap_quantized.py:75-87: [n, D] -> indices [n, num_quantizers, 1].  With learn_rvq the residual VQ runs in training
        mode (rq.train(True), :79-81): k-means init on the first batch, then one EMA codebook update per call."""
        if self.learn_rvq:
            idx, q = self.rq.fit_step(embedding)
        else:
            idx = self.rq.encode(embedding)
            q = None
        if return_rvq_loss:
            if q is None:
                q = torch.stack([self.rq.codebooks[s][idx[:, s]] for s in range(idx.shape[1])]).sum(0)
            return torch.nn.functional.mse_loss(q, embedding.float()).item()
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
