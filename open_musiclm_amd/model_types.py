"""Role aliases that the reference's scripts and type hints import (reference open_musiclm/model_types.py): the semantic
tokenizer role and the neural-codec role, bound to this package's wrappers around the out-of-scope extractors."""
from . import encodec_wrapper as _codec
from . import hf_hubert_kmeans as _semantic

Wav2Vec, NeuralCodec = _semantic.HfHubertWithKmeans, _codec.EncodecWrapper

__all__ = ["Wav2Vec", "NeuralCodec"]
