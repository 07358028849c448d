"""DataPreprocessor boundary stub (reference open_musiclm/preprocess.py): offline audio -> token ETL, outside the hot path.
The on-disk format it defines is read by data.PreprocessedDataset."""
from torch import nn


class DataPreprocessor(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("audio preprocessing needs the pretrained tokenizers (CLAP / MERT / Encodec); outside the hot path")
