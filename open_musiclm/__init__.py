"""Drop-in alias: `import open_musiclm.<module>` resolves to `open_musiclm_amd.<module>`, so the reference's
scripts (scripts/train_*_stage.py, scripts/infer*.py: `from open_musiclm.config import ...`) run unchanged
against the MI355X implementation."""
import importlib
import sys

_MODULES = ("utils", "transformer", "open_musiclm", "optimizer", "parallel", "data", "trainer", "config", "model_types",
            "clap_quantized", "hf_hubert_kmeans", "encodec_wrapper", "preprocess", "engine", "ops", "hip")
for _m in _MODULES:
    sys.modules[f"{__name__}.{_m}"] = importlib.import_module(f"open_musiclm_amd.{_m}")
    globals()[_m] = sys.modules[f"{__name__}.{_m}"]
