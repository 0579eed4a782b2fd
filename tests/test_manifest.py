"""State-dict compatibility: the manifest the host side builds (visper_lm_amd.params) equals the reference
model's own state_dict keys/shapes recorded in the golden fixture (minus the out-of-scope frozen DPT decoder)."""
import json

from oracle import cases
from visper_lm_amd.config import VisperConfig
from visper_lm_amd.params import param_shapes
from visper_lm_amd.engine import is_trainable


def test_manifest_matches_reference_state_dict():
    ocfg, W, batch, g = cases.tiny_llama_case()
    ref = {k: tuple(v) for k, v in json.loads(str(g["manifest"])).items()}
    mine = param_shapes(VisperConfig(**vars(ocfg)), vit_nested=False)
    assert set(mine) == set(ref), (sorted(set(mine) ^ set(ref))[:10])
    for k in ref:
        assert tuple(mine[k]) == ref[k], (k, mine[k], ref[k])
    # reference pin (transformers 4.41.1) nests the CLIP weights one level deeper
    nested = param_shapes(VisperConfig(**vars(ocfg)), vit_nested=True)
    assert "model.vision_tower.vision_tower.vision_model.embeddings.class_embedding" in nested


def test_trainable_set_matches_reference_pt_stage():
    ocfg, W, batch, g = cases.tiny_llama_case()
    ref_tr = set(json.loads(str(g["trainable"])))
    mine = {k for k in param_shapes(VisperConfig(**vars(ocfg)), vit_nested=False) if is_trainable(k)}
    assert mine == ref_tr


def test_api_mirror_state_dict_keys_cpu():
    """The nn.Module mirror (constructed on CPU, no compute) exposes exactly the reference's state-dict keys/shapes
    under the pinned transformers==4.41.1 naming, and the PT-stage requires_grad pattern."""
    import torch
    from visper_lm_amd.model import OlaLlavaLlamaForCausalLM, OlaLlavaLlamaConfig
    ocfg, W, batch, g = cases.tiny_llama_case()
    model = OlaLlavaLlamaForCausalLM(OlaLlavaLlamaConfig(**vars(ocfg)), device="cpu", init="empty")
    ref = {k: tuple(v) for k, v in json.loads(str(g["manifest"])).items()}
    ref = {(k.replace("model.vision_tower.vision_tower.", "model.vision_tower.vision_tower.vision_model.")
            if k.startswith("model.vision_tower.vision_tower.") else k): v for k, v in ref.items()}
    sd = model.state_dict()
    assert set(sd) == set(ref), sorted(set(sd) ^ set(ref))[:8]
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k], k
    tr = {n for n, p in model.named_parameters() if p.requires_grad}
    assert tr == set(json.loads(str(g["trainable"])))
    assert model.get_model().mm_projector[2].weight.shape == (ocfg.hidden_size, ocfg.hidden_size)
    assert model.image_seg_heads[1].projector.proj_in.weight.shape == (1536, ocfg.hidden_size)
