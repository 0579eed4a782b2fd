"""Host-side mirror of the reference's `ola_vlm.model` API surface for the PT-stage hot path:
same class names, constructor/forward signatures, attribute names, output fields and state-dict keys, so a
trainer written against the reference drives the MI355X kernels unchanged.  Compute is NOT here: every forward
goes through visper_lm_amd.engine.Engine (C ABI -> HIP kernels)."""
from .ola_arch import OlaLlavaMetaModel, OlaLlavaMetaForCausalLM                      # noqa: F401
from .language_model import (OlaLlavaLlamaConfig, OlaLlavaLlamaModel, OlaLlavaLlamaForCausalLM,      # noqa: F401
                             OlaLlavaPhi3Config, OlaLlavaPhi3Model, OlaLlavaPhi3ForCausalLM,
                             OlaCausalLLMOutputWithPast, BaseOLA_VLM)
from .builders import build_vision_tower, build_vision_projector, CLIPVisionTower    # noqa: F401
from .llava import (LlavaConfig, LlavaPhi3Config, LlavaMetaModel, LlavaMetaForCausalLM, LlavaLlamaModel, LlavaPhi3Model,    # noqa: F401
                    LlavaLlamaForCausalLM, LlavaPhi3ForCausalLM, CausalLMOutputWithPast)
from .hf_auto import register_auto_classes                                            # noqa: F401
