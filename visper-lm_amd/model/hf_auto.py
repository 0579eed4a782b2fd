"""AutoConfig / AutoModelForCausalLM registration of the mirror classes, the counterpart of the reference's
    AutoConfig.register("ola_llama", OlaLlavaLlamaConfig); AutoModelForCausalLM.register(OlaLlavaLlamaConfig, OlaLlavaLlamaForCausalLM)
(ola_llama.py:246-247, ola_phi3.py:243-244, llava_llama.py:174-175, llava_phi3.py), so `AutoConfig.from_pretrained(ckpt_dir)` and
`AutoModelForCausalLM.from_pretrained(ckpt_dir)` resolve "ola_llama" / "ola_phi3" / "llava_llama" / "llava_phi3" checkpoints to
the MI355X classes.  transformers is imported only when this is called (the kernels and the engine never need it)."""
from __future__ import annotations

_REGISTERED = {}


def register_auto_classes():
    """Idempotent.  Returns {model_type: (hf_config_class, model_class)}."""
    if _REGISTERED:
        return _REGISTERED
    from transformers import AutoConfig, AutoModelForCausalLM, PretrainedConfig
    from . import language_model as lm, llava

    def make(model_type, visper_cfg_cls, model_cls):
        class HFConfig(PretrainedConfig):
            """PretrainedConfig view of a VisperConfig (same attribute names); `to_visper()` rebuilds the engine-side config."""
            def __init__(self, **kw):
                base = visper_cfg_cls().to_dict()
                own = {k: kw.pop(k) for k in list(kw) if k in base}
                super().__init__(**kw)
                for k, v in {**base, **own}.items():
                    setattr(self, k, v)

            def to_visper(self):
                base = visper_cfg_cls().to_dict()
                drop = getattr(visper_cfg_cls, "STORED_KEYS_IGNORED", ())
                c = visper_cfg_cls(**{k: getattr(self, k) for k in base if hasattr(self, k) and k not in drop})
                c.model_type = model_type
                return c
        HFConfig.model_type = model_type
        HFConfig.__name__ = visper_cfg_cls.__name__

        class Auto(model_cls):
            """`model_cls` constructible from the HF config object (AutoModelForCausalLM.from_config / from_pretrained)."""
            config_class = HFConfig

            def __init__(self, config, **kw):
                super().__init__(config.to_visper() if isinstance(config, PretrainedConfig) else config, **kw)

            @classmethod
            def _from_config(cls, config, **kw):
                kw = {k: v for k, v in kw.items() if k in ("device", "dtype", "init", "seed")}
                return cls(config, **kw)

            @classmethod
            def from_pretrained(cls, directory, *a, **kw):
                kw = {k: v for k, v in kw.items() if k in ("device", "dtype", "strict")}
                m = model_cls.from_pretrained(directory, **kw)
                m.__class__ = cls
                return m
        Auto.__name__ = model_cls.__name__
        try:
            AutoConfig.register(model_type, HFConfig)
            AutoModelForCausalLM.register(HFConfig, Auto)
        except ValueError:                                        # already registered in this process (e.g. by the reference package)
            pass
        _REGISTERED[model_type] = (HFConfig, Auto)

    make("ola_llama", lm.OlaLlavaLlamaConfig, lm.OlaLlavaLlamaForCausalLM)
    make("ola_phi3", lm.OlaLlavaPhi3Config, lm.OlaLlavaPhi3ForCausalLM)
    make("llava_llama", llava.LlavaConfig, llava.LlavaLlamaForCausalLM)
    make("llava_phi3", llava.LlavaPhi3Config, llava.LlavaPhi3ForCausalLM)
    return _REGISTERED
