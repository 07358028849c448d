"""JSON config dataclasses and factory functions with the reference's names and signatures
(reference open_musiclm/config.py:21-454), wired to the MI355X implementation."""
from __future__ import annotations

import json
import os
import sys
from dataclasses import asdict, dataclass, fields
from pathlib import Path
from typing import List, Optional

import torch

from .clap_quantized import ClapQuantized, create_clap_quantized
from .encodec_wrapper import EncodecWrapper, create_encodec_24khz
from .hf_hubert_kmeans import HfHubertWithKmeans, get_hubert_kmeans
from .open_musiclm import (MusicLM, TokenConditionedTransformer, create_coarse_transformer, create_fine_transformer,
                           create_semantic_transformer)
from .preprocess import DataPreprocessor
from .trainer import ClapRVQTrainer, HfHubertKmeansTrainer, SingleStageTrainer
from .utils import exists


@dataclass
class ClapRVQConfig:
    rq_num_quantizers: int
    codebook_size: int
    enable_fusion: bool = False
    rq_ema_decay: float = 0.95
    threshold_ema_dead_code: float = 0.0
    checkpoint_path: Optional[str] = None
    amodel_type: str = 'HTSAT-tiny'


@dataclass
class HubertKmeansConfig:
    model_name: str
    normalize_embeds: bool
    embed_layer: int = 7
    target_sample_hz: int = 16000
    seq_len_multiple_of: int = 320
    codebook_size: int = 1024
    output_hz: int = 50


@dataclass
class EncodecConfig:
    bandwidth: float
    codebook_size: int
    output_hz: int = 75


@dataclass
class _StageConfig:
    dim: int = 1024
    depth: int = 6
    heads: int = 8
    attn_dropout: float = 0.0
    ff_dropout: float = 0.1
    use_conv_ff: bool = True
    grad_shrink_alpha: float = 0.1
    non_causal_prefix_size: int = 0
    relative_position_bias_type: str = 'continuous'
    use_memory_efficient_attention: bool = False
    use_absolute_position_embeddings: bool = False
    max_absolute_position_embeddings: int = 262


@dataclass
class SemanticConfig(_StageConfig):
    max_absolute_position_embeddings: int = 12 + 250


@dataclass
class CoarseConfig(_StageConfig):
    max_absolute_position_embeddings: int = 12 + 100 + 600


@dataclass
class FineConfig(_StageConfig):
    max_absolute_position_embeddings: int = 12 + 300 + 900


@dataclass
class GlobalConfig:
    semantic_audio_length_seconds: float = 10.0
    coarse_audio_length_seconds: float = 4.0
    fine_audio_length_seconds: float = 2.0
    clap_audio_length_seconds: float = 10.0
    num_coarse_quantizers: int = 3
    num_fine_quantizers: int = 5


@dataclass
class MusicLMModelConfig:
    clap_rvq_cfg: ClapRVQConfig
    hubert_kmeans_cfg: HubertKmeansConfig
    encodec_cfg: EncodecConfig
    semantic_cfg: SemanticConfig
    coarse_cfg: CoarseConfig
    fine_cfg: FineConfig
    global_cfg: GlobalConfig


@dataclass
class ClapRVQTrainerConfig:
    folder: str
    num_train_steps: int
    batch_size: int
    accumulate_batches: int
    save_model_every: int
    save_results_every: int


@dataclass
class HubertKmeansTrainerConfig:
    folder: str
    feature_extraction_num_steps: int
    feature_extraction_batch_size: int


@dataclass
class SingleStageTrainerConfig:
    stage: str
    folder: str
    valid_frac: float
    lr: float
    lr_warmup: int
    batch_size: int
    grad_accum_every: int
    wd: float
    max_grad_norm: float
    cross_entropy_loss_weights: List[float]
    num_train_steps: int
    save_results_every: int
    save_model_every: int
    save_predicted_tokens: bool
    save_reconstructed_wave: bool
    use_preprocessed_data: bool


@dataclass
class DataPreprocessorConfig:
    folder: str = './data/fma_large'
    metadata_folder: str = './data/fma_metadata'
    results_folder: str = './fma_preprocessed'
    max_audio_length_seconds: int = 30
    random_crop: bool = True
    num_crops: int = 1
    clap_batch_size: int = 32


@dataclass
class MusicLMTrainingConfig:
    clap_rvq_trainer_cfg: ClapRVQTrainerConfig
    hubert_kmeans_trainer_cfg: HubertKmeansTrainerConfig
    semantic_trainer_cfg: SingleStageTrainerConfig
    coarse_trainer_cfg: SingleStageTrainerConfig
    fine_trainer_cfg: SingleStageTrainerConfig
    data_preprocessor_cfg: DataPreprocessorConfig


def _read(path):
    with open(path, 'r') as f:
        return json.load(f)


def load_model_config(config_path: str) -> MusicLMModelConfig:
    c = _read(config_path)
    return MusicLMModelConfig(
        clap_rvq_cfg=ClapRVQConfig(**c['clap_rvq_cfg']), hubert_kmeans_cfg=HubertKmeansConfig(**c['hubert_kmeans_cfg']),
        encodec_cfg=EncodecConfig(**c['encodec_cfg']), semantic_cfg=SemanticConfig(**c['semantic_cfg']),
        coarse_cfg=CoarseConfig(**c['coarse_cfg']), fine_cfg=FineConfig(**c['fine_cfg']),
        global_cfg=GlobalConfig(**c['global_cfg']))


def load_training_config(config_path: str) -> MusicLMTrainingConfig:
    c = _read(config_path)
    return MusicLMTrainingConfig(
        clap_rvq_trainer_cfg=ClapRVQTrainerConfig(**c['clap_rvq_trainer_cfg']),
        hubert_kmeans_trainer_cfg=HubertKmeansTrainerConfig(**c['hubert_kmeans_trainer_cfg']),
        semantic_trainer_cfg=SingleStageTrainerConfig(**c['semantic_trainer_cfg']),
        coarse_trainer_cfg=SingleStageTrainerConfig(**c['coarse_trainer_cfg']),
        fine_trainer_cfg=SingleStageTrainerConfig(**c['fine_trainer_cfg']),
        data_preprocessor_cfg=DataPreprocessorConfig(**c['data_preprocessor_cfg']))


def load_model(model, path):
    """strict state_dict load of a reference-format checkpoint (config.py:199-204)."""
    path = Path(path)
    assert path.exists(), f'checkpoint does not exist at {str(path)}'
    model.load_state_dict(torch.load(str(path), map_location=next(model.parameters()).device))


class disable_print:
    def __enter__(self):
        self._stdout = sys.stdout
        sys.stdout = open(os.devnull, 'w')

    def __exit__(self, *exc):
        sys.stdout.close()
        sys.stdout = self._stdout


def create_clap_quantized_from_config(model_config: MusicLMModelConfig, rvq_path: Optional[str], device, **kwargs) -> ClapQuantized:
    with disable_print():
        return create_clap_quantized(**asdict(model_config.clap_rvq_cfg), device=device, learn_rvq=False,
                                     rvq_checkpoint_path=rvq_path, **kwargs).to(device)


def create_hubert_kmeans_from_config(model_config: MusicLMModelConfig, kmeans_path: Optional[str], device, **kwargs) -> HfHubertWithKmeans:
    return get_hubert_kmeans(**asdict(model_config.hubert_kmeans_cfg), kmeans_path=kmeans_path, **kwargs).to(device)


def create_encodec_from_config(model_config: MusicLMModelConfig, device, **kwargs) -> EncodecWrapper:
    return create_encodec_24khz(**asdict(model_config.encodec_cfg), **kwargs).to(device)


def _finish(transformer, checkpoint_path, device):
    transformer = transformer.to(device)
    if exists(checkpoint_path):
        load_model(transformer, checkpoint_path)
    return transformer


def create_semantic_transformer_from_config(model_config, checkpoint_path: Optional[str], device, **kwargs) -> TokenConditionedTransformer:
    return _finish(create_semantic_transformer(
        **asdict(model_config.semantic_cfg), clap_codebook_size=model_config.clap_rvq_cfg.codebook_size,
        semantic_codebook_size=model_config.hubert_kmeans_cfg.codebook_size,
        num_clap_quantizers=model_config.clap_rvq_cfg.rq_num_quantizers, **kwargs), checkpoint_path, device)


def create_coarse_transformer_from_config(model_config, checkpoint_path: Optional[str], device, **kwargs) -> TokenConditionedTransformer:
    return _finish(create_coarse_transformer(
        **asdict(model_config.coarse_cfg), clap_codebook_size=model_config.clap_rvq_cfg.codebook_size,
        semantic_codebook_size=model_config.hubert_kmeans_cfg.codebook_size,
        acoustic_codebook_size=model_config.encodec_cfg.codebook_size,
        num_clap_quantizers=model_config.clap_rvq_cfg.rq_num_quantizers,
        num_coarse_quantizers=model_config.global_cfg.num_coarse_quantizers, **kwargs), checkpoint_path, device)


def create_fine_transformer_from_config(model_config, checkpoint_path: Optional[str], device, **kwargs) -> TokenConditionedTransformer:
    return _finish(create_fine_transformer(
        **asdict(model_config.fine_cfg), clap_codebook_size=model_config.clap_rvq_cfg.codebook_size,
        acoustic_codebook_size=model_config.encodec_cfg.codebook_size,
        num_clap_quantizers=model_config.clap_rvq_cfg.rq_num_quantizers,
        num_coarse_quantizers=model_config.global_cfg.num_coarse_quantizers,
        num_fine_quantizers=model_config.global_cfg.num_fine_quantizers, **kwargs), checkpoint_path, device)


def create_clap_rvq_trainer_from_config(model_config, training_config, clap, results_folder: str, device,
                                        accelerate_kwargs: dict = {}, config_paths=None, **kwargs):
    return ClapRVQTrainer(audio_conditioner=clap, results_folder=results_folder,
                          data_max_length_seconds=model_config.global_cfg.semantic_audio_length_seconds,
                          accelerate_kwargs=accelerate_kwargs, config_paths=config_paths,
                          **asdict(training_config.clap_rvq_trainer_cfg), **kwargs).to(device)


def create_hubert_kmeans_trainer_from_config(model_config, training_config, hubert_kmeans, results_folder: str, device,
                                             config_paths=None, **kwargs):
    return HfHubertKmeansTrainer(hubert_kmeans=hubert_kmeans, results_folder=results_folder,
                                 data_max_length_seconds=model_config.global_cfg.semantic_audio_length_seconds,
                                 config_paths=config_paths, **asdict(training_config.hubert_kmeans_trainer_cfg), **kwargs).to(device)


def create_single_stage_trainer_from_config(model_config, training_config, stage, results_folder: str,
                                            transformer: TokenConditionedTransformer, clap=None, wav2vec=None,
                                            encodec_wrapper=None, device='cpu', accelerate_kwargs: dict = {},
                                            config_paths: Optional[List[str]] = None, **kwargs) -> SingleStageTrainer:
    g = model_config.global_cfg
    if stage == 'semantic':
        trainer_cfg = training_config.semantic_trainer_cfg
        lengths = (g.semantic_audio_length_seconds, g.semantic_audio_length_seconds)
    elif stage == 'coarse':
        trainer_cfg = training_config.coarse_trainer_cfg
        lengths = (g.semantic_audio_length_seconds, g.coarse_audio_length_seconds, g.coarse_audio_length_seconds)
    elif stage == 'fine':
        trainer_cfg = training_config.fine_trainer_cfg
        lengths = (g.semantic_audio_length_seconds, g.fine_audio_length_seconds)
    else:
        raise ValueError(f'invalid stage: {stage}')
    return SingleStageTrainer(model_config=model_config, training_config=training_config, transformer=transformer,
                              audio_conditioner=clap, wav2vec=wav2vec, neural_codec=encodec_wrapper,
                              results_folder=results_folder, data_max_length_seconds=lengths,
                              accelerate_kwargs=accelerate_kwargs, config_paths=config_paths,
                              **asdict(trainer_cfg), **kwargs).to(device)


def create_data_preprocessor_from_config(model_config, training_config, clap, wav2vec, encodec_wrapper, device='cpu',
                                         config_paths=None, **kwargs):
    g = model_config.global_cfg
    return DataPreprocessor(audio_conditioner=clap, wav2vec=wav2vec, neural_codec=encodec_wrapper,
                            num_coarse_quantizers=g.num_coarse_quantizers,
                            semantic_audio_length_seconds=g.semantic_audio_length_seconds,
                            coarse_audio_length_seconds=g.coarse_audio_length_seconds,
                            fine_audio_length_seconds=g.fine_audio_length_seconds,
                            clap_audio_length_seconds=g.clap_audio_length_seconds, config_paths=config_paths,
                            **asdict(training_config.data_preprocessor_cfg), **kwargs).to(device)


def create_musiclm_from_config(model_config, semantic_path: str, coarse_path: str, fine_path: str, rvq_path: str,
                               kmeans_path: str, device, **kwargs):
    clap = create_clap_quantized_from_config(model_config, rvq_path, device)
    wav2vec = create_hubert_kmeans_from_config(model_config, kmeans_path, device)
    encodec_wrapper = create_encodec_from_config(model_config, device)
    return MusicLM(wav2vec=wav2vec, clap=clap, neural_codec=encodec_wrapper,
                   semantic_transformer=create_semantic_transformer_from_config(model_config, semantic_path, device),
                   coarse_transformer=create_coarse_transformer_from_config(model_config, coarse_path, device),
                   fine_transformer=create_fine_transformer_from_config(model_config, fine_path, device), **kwargs).to(device)
