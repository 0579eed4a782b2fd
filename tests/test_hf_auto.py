"""AutoConfig / AutoModelForCausalLM registration (ola_llama.py:246-247, ola_phi3.py, llava_llama.py:174-175, llava_phi3.py): the four
model types resolve to the mirror classes; a config.json written by save_pretrained's recipe loads back through AutoConfig."""
import json
import os

from visper_lm_amd.model import register_auto_classes, OlaLlavaLlamaConfig, LlavaConfig


def test_auto_classes_resolve_all_four_model_types(tmp_path):
    from transformers import AutoConfig, AutoModelForCausalLM
    reg = register_auto_classes()
    assert sorted(reg) == ["llava_llama", "llava_phi3", "ola_llama", "ola_phi3"]
    assert register_auto_classes() is reg                                     # idempotent
    c = AutoConfig.for_model("ola_phi3")
    assert c.model_type == "ola_phi3" and c.hidden_size == 3072 and c.sliding_window == 2047 and c.to_visper().num_sys_tokens == 13
    c = AutoConfig.for_model("llava_llama", num_hidden_layers=2)
    v = c.to_visper()
    assert v.train_llm and not v.aux_heads and v.num_hidden_layers == 2 and v.num_task_tokens == 0 and isinstance(v, LlavaConfig)
    cfg = OlaLlavaLlamaConfig(num_hidden_layers=3)
    cd = {k: (list(x) if isinstance(x, tuple) else x) for k, x in cfg.to_dict().items()}
    cd["model_type"] = "ola_llama"
    json.dump(cd, open(os.path.join(tmp_path, "config.json"), "w"))
    c2 = AutoConfig.from_pretrained(str(tmp_path))
    assert type(c2) is reg["ola_llama"][0] and c2.num_hidden_layers == 3 and c2.image_seg["seg_layer_indices"] == "18"
    assert AutoModelForCausalLM._model_mapping[type(c2)] is reg["ola_llama"][1]
    assert issubclass(reg["llava_llama"][1], __import__("visper_lm_amd.model", fromlist=["x"]).LlavaMetaForCausalLM)


def test_ift_config_ignores_a_stored_pt_trainability_and_model_type():
    """a PT-stage config.json carries train_llm=False / aux_heads=True / model_type "ola_*" (save_pretrained dumps
    config.to_dict()); the IFT classes are trainable whatever the stored config says (train.py:1045-1068), and stay llava_* models."""
    from visper_lm_amd.model import LlavaPhi3Config, OlaLlavaPhi3Config
    pt = OlaLlavaLlamaConfig(num_hidden_layers=2)
    assert pt.train_llm is False and pt.aux_heads is True
    c = LlavaConfig(**pt.to_dict())
    assert c.train_llm is True and c.aux_heads is False and c.model_type == "llava_llama"
    assert c.num_task_tokens == 8 and c.aux_mode == "gen-depth-seg" and c.task_token_format == "emb"      # the PT recipe's tokens are kept
    assert LlavaConfig(**{**pt.to_dict(), "freeze_llm": True}).train_llm is False                     # the only way to freeze: explicit
    p = LlavaPhi3Config()
    assert p.model_type == "llava_phi3" and "model_type" not in p.to_dict() and p.arch == "phi3" and p.train_llm
    p2 = LlavaPhi3Config(**OlaLlavaPhi3Config().to_dict())
    assert p2.model_type == "llava_phi3" and p2.train_llm and p2.hidden_size == 3072
    reg = register_auto_classes()
    assert reg["llava_phi3"][0]().to_visper().model_type == "llava_phi3"
