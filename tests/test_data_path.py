"""Host data path: batch format helpers against fixtures produced by the reference's own functions
(tests/golden/data_path.json <- oracle/gen_golden.py data) and against torch's pad_sequence semantics."""
import json
import os
import types

import numpy as np
import torch

from visper_lm_amd import data
from visper_lm_amd.config import IGNORE_INDEX

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "data_path.json")))


class Tok:
    bos_token_id = 1

    def __init__(self, bos):
        self.bos = bos

    def __call__(self, text):
        ids = [3 + (sum(map(ord, w)) % 997) for w in text.split()]
        return types.SimpleNamespace(input_ids=([1] if self.bos else []) + ids)


def test_tokenizer_image_token_matches_reference():
    for p, a, b in zip(G["prompts"], G["with_bos"], G["no_bos"]):
        assert data.tokenizer_image_token(p, Tok(True)) == a, p
        assert data.tokenizer_image_token(p, Tok(False)) == b, p
    t = data.tokenizer_image_token(G["prompts"][1], Tok(True), return_tensors="pt")
    assert t.dtype == torch.long and t.tolist() == G["with_bos"][1]


def test_expand2square_matches_reference():
    from PIL import Image
    for case in G["squares"]:
        w, h = case["size"]
        img = Image.fromarray((np.arange(w * h * 3).reshape(h, w, 3) % 251).astype(np.uint8), "RGB")
        assert np.array_equal(np.asarray(data.expand2square(img, (122, 116, 104))), np.array(case["out"], dtype=np.uint8))


def test_collator_pads_truncates_and_masks_like_the_reference():
    """ola_vlm_train.py:882-925: pad_sequence(batch_first, pad id / IGNORE_INDEX), [:, :model_max_length], mask = ids != pad."""
    pad, mx = 0, 9
    inst = []
    for i, L in enumerate((5, 12, 9)):
        ids = torch.arange(1, L + 1) + 10 * i
        inst.append(dict(input_ids=ids, labels=ids.clone().masked_fill(ids % 3 == 0, IGNORE_INDEX), image=torch.full((3, 4, 4), float(i)),
                         pil_image=None, seg_mask=1, depth_mask=0, gen_mask=1))
    b = data.Collator(pad, mx, pin_memory=False)(inst)
    ref_ids = torch.nn.utils.rnn.pad_sequence([x["input_ids"] for x in inst], batch_first=True, padding_value=pad)[:, :mx]
    ref_lab = torch.nn.utils.rnn.pad_sequence([x["labels"] for x in inst], batch_first=True, padding_value=IGNORE_INDEX)[:, :mx]
    assert torch.equal(b["input_ids"], ref_ids) and torch.equal(b["labels"], ref_lab) and torch.equal(b["attention_mask"], ref_ids.ne(pad))
    assert b["images"].shape == (3, 3, 4, 4) and b["pil_images"] == [None] * 3
    assert b["seg_mask"].tolist() == [1, 1, 1] and b["depth_mask"].tolist() == [0, 0, 0]
    inst[1]["image"] = torch.zeros(3, 5, 5)                                   # ragged image shapes stay a list (reference :905-909)
    assert isinstance(data.Collator(pad, mx, pin_memory=False)(inst)["images"], list)


def test_collator_matches_the_reference_class():
    """Fixture produced by the reference's own DataCollatorForSupervisedDataset (class body compiled from ola_vlm_train.py:881-925 at
    generation time, oracle/gen_golden.py data): same keys, padding, truncation, mask, image stacking rule and task-mask tensors."""
    for case in G["collator"]:
        inst = []
        for i, L in enumerate((5, 12, 9)):
            ids = torch.arange(1, L + 1) + 10 * i
            side = 5 if (case["ragged"] and i == 1) else 4
            inst.append(dict(input_ids=ids, labels=ids.clone().masked_fill(ids % 3 == 0, IGNORE_INDEX), image=torch.full((3, side, side), float(i)),
                             pil_image=None, seg_mask=int(i != 1), depth_mask=int(i == 1), gen_mask=1))
        b = data.Collator(0, 9, pin_memory=False)(inst)
        assert sorted(b.keys()) == case["keys"]
        assert b["input_ids"].tolist() == case["input_ids"] and b["labels"].tolist() == case["labels"]
        assert b["attention_mask"].long().tolist() == case["attention_mask"] and b["attention_mask"].dtype == torch.bool
        assert isinstance(b["images"], list) == case["images_is_list"]
        if not case["images_is_list"]:
            assert list(b["images"].shape) == case["images_shape"]
        for k in ("seg_mask", "depth_mask", "gen_mask"):
            assert b[k].tolist() == case[k] and str(b[k].dtype) == case["seg_mask_dtype"]


def test_adapter_state_selects_the_projector_only():
    named = [("model.mm_projector.0.weight", torch.zeros(2, 2)), ("model.embed_tokens.weight", torch.zeros(3, 2)),
             ("image_gen_heads.0.projector.proj_in.weight", torch.zeros(1))]
    assert list(data.adapter_state(named)) == ["model.mm_projector.0.weight"]
    assert list(data.adapter_state(named, use_im_start_end=True)) == ["model.mm_projector.0.weight", "model.embed_tokens.weight"]


def test_process_images_matches_reference_with_hf_clip_processor():
    """ola_vlm/mm_utils.py:309-333 driven with HF's CLIPImageProcessor in the generator (tests/golden/data_path.json "process_images"):
    `data.process_images` + `data.ClipImageProcessor` (PIL + numpy restatement of the openai/clip-vit-large-patch14-336 preprocessing)
    reproduce the reference's pixel tensors for the "pad" mode of the training scripts and the default mode."""
    from PIL import Image
    imgs = [Image.fromarray(((np.arange(w * h * 3).reshape(h, w, 3) * 7 + 13 * (np.arange(h)[:, None, None] % 5)) % 253).astype(np.uint8), "RGB")
            for (w, h) in ((90, 41), (37, 120), (24, 24))]
    ip = data.ClipImageProcessor()
    for mode, ref in G["process_images"].items():
        px = data.process_images(imgs, ip, types.SimpleNamespace(image_aspect_ratio=mode))
        assert list(px.shape) == ref["shape"] and px.dtype == torch.float32
        sub = px[:, :, ::17, ::13].numpy()
        assert np.abs(sub - np.array(ref["sub"])).max() < 2e-6, (mode, np.abs(sub - np.array(ref["sub"])).max())
        assert abs(float(px.double().mean()) - ref["mean"]) < 1e-6
    import pytest
    with pytest.raises(NotImplementedError):
        data.process_images(imgs, ip, types.SimpleNamespace(image_aspect_ratio="anyres"))
