"""reference open_musiclm/model_types.py"""
from .encodec_wrapper import EncodecWrapper
from .hf_hubert_kmeans import HfHubertWithKmeans

Wav2Vec = HfHubertWithKmeans
NeuralCodec = EncodecWrapper
