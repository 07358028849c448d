"""HfHubertWithKmeans (reference open_musiclm/hf_hubert_kmeans.py).  The MERT/HuBERT feature extractor is a
pretrained third-party network outside the hot path; the k-means ASSIGN step (:87, sklearn predict) runs as a
HIP nearest-centroid kernel, bit-exact against oracle.kmeans_assign.  The FIT (:98-149, what scripts/train_hubert_kmeans.py drives
through HfHubertKmeansTrainer) is sklearn's MiniBatchKMeans in the reference; it is the same call here, so a fit on the same
features and seed yields the reference's centroids bit for bit (tests/golden/kmeans_fit.npz, produced by the reference itself)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import ops
from .utils import exists


class KmeansAssigner(nn.Module):
    def __init__(self, centroids: torch.Tensor):
        super().__init__()
        self.register_buffer("centroids", centroids.float().contiguous())          # [C, D]
        self._cT = None

    @property
    def codebook_size(self):
        return self.centroids.shape[0]

    @torch.no_grad()
    def predict(self, feats: torch.Tensor) -> torch.Tensor:
        """feats [n, D] fp32 (cuda) -> cluster ids [n] int64."""
        if self._cT is None or self._cT.device != self.centroids.device:
            self._cT = self.centroids.t().contiguous()                              # [D, C]
        feats = feats.contiguous().float()
        idx = torch.empty(feats.shape[0], 1, dtype=torch.int32, device=feats.device)
        ops.nearest_centroid(feats, self._cT, idx, feats.shape[0], feats.shape[1], self.centroids.shape[0])
        return idx[:, 0].long()


class HfHubertWithKmeans(nn.Module):
    def __init__(self, *, hubert=None, kmeans=None, embed_layer=7, target_sample_hz=16000, seq_len_multiple_of=None,
                 normalize_embeds=True, codebook_size: int = 1024, output_hz: int = 50):
        super().__init__()
        self.target_sample_hz, self.output_hz = target_sample_hz, output_hz
        self.embed_layer, self.normalize_embeds = embed_layer, normalize_embeds
        self.seq_len_multiple_of = seq_len_multiple_of
        self.codebook_size = codebook_size
        self.hubert = hubert
        self.kmeans = None
        if exists(kmeans):
            centers = torch.as_tensor(getattr(kmeans, "cluster_centers_", kmeans))
            self.kmeans = KmeansAssigner(centers)
            self.codebook_size = self.kmeans.codebook_size

    @torch.no_grad()
    def assign(self, embed: torch.Tensor) -> torch.Tensor:
        """embed [B, T, D] -> ids [B, T] (hf_hubert_kmeans.py:78-94 minus the feature extractor)."""
        from .utils import zero_mean_unit_var_norm
        if self.normalize_embeds:
            embed = zero_mean_unit_var_norm(embed)
        b, t, d = embed.shape
        return self.kmeans.predict(embed.reshape(b * t, d)).reshape(b, t)

    @torch.no_grad()
    def forward(self, wav_input, flatten=True, return_embed=False, input_sample_hz=None):
        """hf_hubert_kmeans.py:54-93 with a supplied extractor: any module with HF HuBERT's call
        (`hubert(input_values=, attention_mask=, output_hidden_states=True).hidden_states[layer]`).  The assignment to centroids
        runs on the HIP nearest-centroid kernel (sklearn predict on the host in the reference); ids come back on the wave's device."""
        if not exists(self.hubert):
            raise RuntimeError("HfHubertWithKmeans was built without the MERT/HuBERT feature extractor (pretrained weights are "
                               "outside the MI355X hot path); use .assign(features) or supply semantic_token_ids")
        assert return_embed or exists(self.kmeans), "kmeans model must be provided if return_embed==False"
        if exists(input_sample_hz) and input_sample_hz != self.target_sample_hz:
            raise ImportError("resampling needs torchaudio, which is not part of this build: feed audio at target_sample_hz")
        if exists(self.seq_len_multiple_of):
            from .utils import curtail_to_multiple
            wav_input = curtail_to_multiple(wav_input, self.seq_len_multiple_of)
        outputs = self.hubert(input_values=wav_input, attention_mask=torch.ones_like(wav_input), output_hidden_states=True)
        embed = outputs.hidden_states[self.embed_layer]
        if return_embed:
            from .utils import zero_mean_unit_var_norm
            return zero_mean_unit_var_norm(embed) if self.normalize_embeds else embed
        ids = self.assign(embed)                                   # [B, T]; normalises inside
        return ids.reshape(-1) if flatten else ids


def get_kmeans_model(n_clusters, init, max_iter, batch_size, tol, max_no_improvement, n_init, reassignment_ratio, verbose=1):
    """hf_hubert_kmeans.py:95-118."""
    from sklearn.cluster import MiniBatchKMeans
    return MiniBatchKMeans(n_clusters=n_clusters, init=init, max_iter=max_iter, batch_size=batch_size, verbose=verbose,
                           compute_labels=False, tol=tol, max_no_improvement=max_no_improvement, init_size=None, n_init=n_init,
                           reassignment_ratio=reassignment_ratio)


def learn_kmeans(feat, seed, km_path='./results/kmeans.joblib', n_clusters=1024, init="k-means++", max_iter=100, batch_size=10000,
                 tol=0.0, n_init=20, reassignment_ratio=0.0, max_no_improvement=100, verbose=1):
    """hf_hubert_kmeans.py:121-149: seeds numpy's global RNG (the estimator has random_state=None), fits, dumps with joblib."""
    import joblib
    import numpy as np
    np.random.seed(seed)
    km_model = get_kmeans_model(n_clusters, init, max_iter, batch_size, tol, max_no_improvement, n_init, reassignment_ratio, verbose)
    km_model.fit(feat)
    joblib.dump(km_model, km_path)
    inertia = -km_model.score(feat) / len(feat)
    print("total intertia: %.5f", inertia)
    print("finished successfully")
    return km_model


def get_hubert_kmeans(model_name: str = "m-a-p/MERT-v0", kmeans_path: Optional[str] = None, **kwargs):
    kmeans = None
    if exists(kmeans_path):
        import joblib
        kmeans = joblib.load(kmeans_path)
    # the reference downloads the checkpoint (hf_hubert_kmeans.py:152-160); here only a copy already in the local Hugging Face cache
    # is used -- without one the wrapper still serves .assign(features) and every token-id path
    hubert = None
    try:
        from transformers import HubertModel
        hubert = HubertModel.from_pretrained(model_name, local_files_only=True)
    except Exception:
        hubert = None
    return HfHubertWithKmeans(hubert=hubert, kmeans=kmeans, **kwargs)
