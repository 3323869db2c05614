"""
hf_auto.py -- HuggingFace Auto-class registration: the reference's ONLY plug-in point.

The reference makes its classes reachable through `AutoModelForVision2Seq.from_pretrained(...)` / `AutoProcessor` by
registering them under the `"openvla"` model type (experiments/robot/openvla_utils.py:38-41; hub export:
vla-scripts/extern/convert_openvla_weights_to_hf.py:259-262):

    AutoConfig.register("openvla", OpenVLAConfig)
    AutoImageProcessor.register(OpenVLAConfig, PrismaticImageProcessor)
    AutoProcessor.register(OpenVLAConfig, PrismaticProcessor)
    AutoModelForVision2Seq.register(OpenVLAConfig, OpenVLAForActionPrediction)

`register_auto_classes()` is the same four lines for the MI355X classes, so the README call sequence
(`AutoModelForVision2Seq.from_pretrained(path, torch_dtype=torch.bfloat16, ...).to("cuda:0")`, `AutoProcessor.from_pretrained(path)`)
resolves to `EmmaXForActionPrediction` / `EmmaXProcessor`.  `transformers` is imported lazily (host side only).  Releases that
dropped `AutoModelForVision2Seq` (transformers >= 5 renamed it `AutoModelForImageTextToText`) get the successor class
registered instead; an Auto class whose optional backend is missing (AutoImageProcessor needs torchvision) is skipped and
reported in the returned dict.
"""

from __future__ import annotations

from typing import Dict

_HF_CONFIG_CLASSES: Dict[str, type] = {}


def hf_config_classes():
    """`PretrainedConfig` stand-ins for the reference's `PrismaticConfig` / `OpenVLAConfig`
    (prismatic/extern/hf/configuration_prismatic.py:72-140): they only carry config.json's fields to `from_pretrained`, which
    re-reads the directory through `EmmaXConfig.from_pretrained` -- the kernels never see an HF object."""
    if _HF_CONFIG_CLASSES:
        return _HF_CONFIG_CLASSES["prismatic"], _HF_CONFIG_CLASSES["openvla"]
    try:
        from transformers import PretrainedConfig
    except ImportError:   # transformers >= 5 spelling
        from transformers import PreTrainedConfig as PretrainedConfig

    class PrismaticConfig(PretrainedConfig):
        model_type = "prismatic"
        is_composition = False

        def __init__(self, vision_backbone_id="dinosiglip-vit-so-224px", llm_backbone_id="llama2-7b-pure",
                     arch_specifier="no-align+fused-gelu-mlp", image_resize_strategy="resize-naive", text_config=None,
                     llm_max_length=2048, pad_token_id=32000, pad_to_multiple_of=64, **kwargs):
            self.vision_backbone_id, self.llm_backbone_id = vision_backbone_id, llm_backbone_id
            self.arch_specifier, self.image_resize_strategy = arch_specifier, image_resize_strategy
            self.text_config = text_config or {}
            self.llm_max_length, self.pad_to_multiple_of = llm_max_length, pad_to_multiple_of
            super().__init__(pad_token_id=pad_token_id, **kwargs)

    class OpenVLAConfig(PrismaticConfig):
        model_type = "openvla"

        def __init__(self, norm_stats=None, n_action_bins=256, **kwargs):
            self.norm_stats, self.n_action_bins = norm_stats, n_action_bins
            super().__init__(**kwargs)

    _HF_CONFIG_CLASSES.update(prismatic=PrismaticConfig, openvla=OpenVLAConfig)
    return PrismaticConfig, OpenVLAConfig


def register_auto_classes(exist_ok: bool = True) -> Dict[str, str]:
    """Mirror of experiments/robot/openvla_utils.py:38-41 for the MI355X classes.  Returns {auto class name: "ok" | reason it
    was skipped}.  Raises ImportError when `transformers` itself is missing."""
    import transformers

    from .modeling import EmmaXForActionPrediction
    from .processing import EmmaXImageProcessor, EmmaXProcessor

    _, OpenVLAConfig = hf_config_classes()
    EmmaXForActionPrediction.config_class = OpenVLAConfig     # Auto `register` checks the names agree
    report: Dict[str, str] = {}

    def _try(name, fn):
        try:
            auto = getattr(transformers, name)
            fn(auto)
            report[name] = "ok"
        except (ImportError, AttributeError) as e:   # class absent in this release / optional backend missing
            report[name] = f"skipped: {type(e).__name__}: {str(e).strip().splitlines()[0] if str(e).strip() else ''}"

    def _once(fn):   # re-registering the same pair is fine (tests, notebooks); anything else propagates
        def run(auto):
            try:
                fn(auto, False)
            except ValueError:
                if not exist_ok:
                    raise
                fn(auto, True)
        return run

    _try("AutoConfig", _once(lambda a, ok: a.register("openvla", OpenVLAConfig, exist_ok=ok)))
    _try("AutoImageProcessor", _once(lambda a, ok: a.register(OpenVLAConfig, EmmaXImageProcessor, exist_ok=ok)))
    _try("AutoProcessor", _once(lambda a, ok: a.register(OpenVLAConfig, EmmaXProcessor, exist_ok=ok)))
    _try("AutoModelForVision2Seq", _once(lambda a, ok: a.register(OpenVLAConfig, EmmaXForActionPrediction, exist_ok=ok)))
    _try("AutoModelForImageTextToText", _once(lambda a, ok: a.register(OpenVLAConfig, EmmaXForActionPrediction, exist_ok=ok)))
    if report.get("AutoConfig") != "ok" or not any(report.get(k) == "ok" for k in ("AutoModelForVision2Seq", "AutoModelForImageTextToText")):
        raise RuntimeError(f"could not register the Emma-X classes with transformers {transformers.__version__}: {report}")
    return report


def require_emmax(model):
    """Fail loudly when an Auto class handed back anything but the MI355X model.  Real OpenVLA / Emma-X checkpoints carry an
    `auto_map` plus bundled `modeling_prismatic.py`; on transformers 4.x `trust_remote_code=True` makes that remote code win
    over locally registered classes, i.e. the README call would silently load the PyTorch reference."""
    from .modeling import EmmaXForActionPrediction

    if not isinstance(model, EmmaXForActionPrediction):
        raise RuntimeError(f"AutoModel resolved to {type(model).__module__}.{type(model).__qualname__}, not emmax's "
                           "EmmaXForActionPrediction: the checkpoint's auto_map / bundled code took precedence. Load with "
                           "trust_remote_code=False (emmax.hf_auto.load_vision2seq does) or strip `auto_map` from config.json")
    return model


def load_vision2seq(path: str, **kwargs):
    """`AutoModelForVision2Seq.from_pretrained(path, ...)` that is guaranteed to come back as the MI355X class: registers the
    Auto classes, forces `trust_remote_code=False` (so a checkpoint's auto_map cannot route to its bundled PyTorch code) and
    checks the resolved type.  Accepts the README's keyword arguments (torch_dtype / dtype, low_cpu_mem_usage, ...)."""
    import transformers

    report = register_auto_classes()
    kwargs.pop("trust_remote_code", None)
    auto = next(getattr(transformers, n) for n in ("AutoModelForVision2Seq", "AutoModelForImageTextToText") if report.get(n) == "ok")
    if int(transformers.__version__.split(".")[0]) >= 5 and "torch_dtype" in kwargs:
        kwargs["dtype"] = kwargs.pop("torch_dtype")
    return require_emmax(auto.from_pretrained(path, trust_remote_code=False, **kwargs))
