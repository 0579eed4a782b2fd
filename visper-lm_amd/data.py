"""Host data path feeding the step: the pieces of the reference's input pipeline that define the *format* of a batch.

* `tokenizer_image_token`  — ola_vlm/mm_utils.py:333-352: split the prompt at "<image>", tokenize the chunks, splice IMAGE_TOKEN_INDEX.
* `expand2square`          — ola_vlm/mm_utils.py:289-306: pad a PIL image to a square with the processor's mean colour
                             (`image_aspect_ratio == "pad"`, the setting of scripts/train/*.sh).
* `Collator`               — ola_vlm/train/ola_vlm_train.py:882-925 `DataCollatorForSupervisedDataset`: right-pad ids with the pad token and
                             labels with IGNORE_INDEX, truncate to `model_max_length`, attention_mask = ids != pad, stack images, pass
                             `pil_images` and the per-sample {seg,depth,gen}_mask flags through.  Output buffers are pinned so the engine's
                             H2D copies of `images` overlap the host-side index plan.
Pinned: the first two against the reference's own functions (oracle/gen_golden.py `data` -> tests/golden/data_path.json); the collator is
a restatement checked against torch's pad_sequence semantics (the reference module does not import under the installed transformers)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .config import IGNORE_INDEX, IMAGE_TOKEN_INDEX


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX, return_tensors: Optional[str] = None):
    chunks = [tokenizer(chunk).input_ids for chunk in prompt.split("<image>")]
    ids: List[int] = []
    offset = 0
    if len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1                                               # keep one BOS, drop the BOS every later chunk starts with
        ids.append(chunks[0][0])
    for i, chunk in enumerate(chunks):
        if i > 0:
            ids.append(image_token_index)
        ids.extend(chunk[offset:])
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return ids


def expand2square(pil_img, background_color):
    from PIL import Image
    width, height = pil_img.size
    if width == height:
        return pil_img
    side = max(width, height)
    result = Image.new(pil_img.mode, (side, side), background_color)
    result.paste(pil_img, (0, (width - height) // 2) if width > height else ((height - width) // 2, 0))
    return result


OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class ClipImageProcessor:
    """The CLIP image pre-processing the reference gets from HF `CLIPImageProcessor.from_pretrained(vision_tower)` (clip_encoder.py:31;
    openai/clip-vit-large-patch14-336: resize shortest edge -> 336 bicubic, center crop 336, rescale 1/255, normalise with the OpenAI
    mean / std, RGB), restated with PIL + numpy so the host data path needs neither transformers nor the hub.  Same interface as the
    HF object for what process_images touches: `.preprocess(images, return_tensors="pt")["pixel_values"]`, `.image_mean`, `.crop_size`."""

    def __init__(self, size: int = 336, crop_size: int = 336, image_mean=OPENAI_CLIP_MEAN, image_std=OPENAI_CLIP_STD):
        self.size = {"shortest_edge": size}
        self.crop_size = {"height": crop_size, "width": crop_size}
        self.image_mean, self.image_std = list(image_mean), list(image_std)

    def _one(self, img):
        import numpy as np
        from PIL import Image
        img = img.convert("RGB")
        w, h = img.size
        short, long = (w, h) if w <= h else (h, w)
        ns = self.size["shortest_edge"]
        nl = int(ns * long / short)                                      # HF get_resize_output_image_size(default_to_square=False)
        nw, nh = (ns, nl) if w <= h else (nl, ns)
        img = img.resize((nw, nh), resample=Image.BICUBIC)
        a = np.asarray(img)                                              # [H, W, 3] uint8
        ch, cw = self.crop_size["height"], self.crop_size["width"]
        top, left = (nh - ch) // 2, (nw - cw) // 2
        if top < 0 or left < 0:                                          # HF center_crop pads with zeros when the image is smaller
            pad = np.zeros((max(nh, ch), max(nw, cw), 3), a.dtype)
            pt, pl = (pad.shape[0] - nh) // 2, (pad.shape[1] - nw) // 2
            pad[pt:pt + nh, pl:pl + nw] = a
            a, nh, nw = pad, pad.shape[0], pad.shape[1]
            top, left = (nh - ch) // 2, (nw - cw) // 2
        a = a[top:top + ch, left:left + cw].astype(np.float32) * np.float32(1.0 / 255.0)
        a = (a - np.asarray(self.image_mean, np.float32)) / np.asarray(self.image_std, np.float32)
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))

    def preprocess(self, images, return_tensors="pt"):
        if not isinstance(images, (list, tuple)):
            images = [images]
        return {"pixel_values": torch.stack([self._one(im) for im in images])}

    __call__ = preprocess


def process_images(images, image_processor, model_cfg):
    """ola_vlm/mm_utils.py:309-333 for the aspect-ratio modes the training scripts use: "pad" (scripts/train/*.sh: expand to a square
    filled with the processor's mean colour, then pre-process each image) and the default (pre-process the list as is).  The anyres /
    highres / crop_split tilings are inference-time options of other checkpoints and are refused."""
    mode = getattr(model_cfg, "image_aspect_ratio", None)
    if mode in ("highres", "anyres", "crop_split") or (isinstance(mode, str) and "anyres_max" in mode):
        raise NotImplementedError(f"image_aspect_ratio={mode!r}: only 'pad' and the default are part of the training data path")
    if mode == "pad":
        bg = tuple(int(x * 255) for x in image_processor.image_mean)
        out = [image_processor.preprocess(expand2square(im, bg), return_tensors="pt")["pixel_values"][0] for im in images]
        if all(x.shape == out[0].shape for x in out):
            return torch.stack(out, dim=0)
        return out
    return image_processor.preprocess(list(images), return_tensors="pt")["pixel_values"]


class Collator:
    def __init__(self, pad_token_id: int, model_max_length: int, pin_memory: bool = True):
        self.pad, self.max_len, self.pin = int(pad_token_id), int(model_max_length), pin_memory

    def _pin(self, t: torch.Tensor) -> torch.Tensor:
        return t.pin_memory() if (self.pin and torch.cuda.is_available()) else t

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, object]:
        n = len(instances)
        T = max(int(x["input_ids"].shape[0]) for x in instances)
        ids = torch.full((n, T), self.pad, dtype=torch.long)
        labels = torch.full((n, T), IGNORE_INDEX, dtype=torch.long)
        for i, x in enumerate(instances):
            L = int(x["input_ids"].shape[0])
            ids[i, :L] = x["input_ids"]
            labels[i, :L] = x["labels"]
        ids, labels = ids[:, :self.max_len], labels[:, :self.max_len]
        batch: Dict[str, object] = dict(input_ids=ids, labels=labels, attention_mask=ids.ne(self.pad))
        if "image" in instances[0]:
            images = [x["image"] for x in instances]
            if all(im is not None and im.shape == images[0].shape for im in images):
                batch["images"] = self._pin(torch.stack(images))
            else:
                batch["images"] = images
        if "pil_image" in instances[0]:
            batch["pil_images"] = [x["pil_image"] for x in instances]
            for k in ("seg_mask", "depth_mask", "gen_mask"):
                batch[k] = torch.tensor([x[k] for x in instances])
        return batch


# ---------------------------------------------------------------------------------------------- checkpoints
ADAPTER_KEYS = ("mm_projector", "vision_resampler")                     # llava_trainer.py:1006


def adapter_state(named_tensors, use_im_start_end: bool = False) -> Dict[str, torch.Tensor]:
    """`get_mm_adapter_state_maybe_zero_3` (llava_trainer.py:116-119): what the PT stage writes to `mm_projector.bin`."""
    keys = ADAPTER_KEYS + (("embed_tokens", "embed_in") if use_im_start_end else ())
    return {k: v.detach().cpu() for k, v in named_tensors if any(m in k for m in keys)}


def save_mm_projector(engine, path: str) -> None:
    """PT-stage checkpoint in the reference's format (torch.save of {state-dict name: tensor}, llava_trainer.py:1014)."""
    ps = engine.ps
    torch.save(adapter_state((k, ps.p(k).to(torch.bfloat16)) for k in ps.index), path)


def load_mm_projector(engine, path: str) -> List[str]:
    """Load a reference `mm_projector.bin` (keys may carry the `base_model.model.` / `model.` prefixes of builder.py:131-137)."""
    sd = torch.load(path, map_location="cpu")
    loaded = []
    for k, v in sd.items():
        name = k
        for pre in ("base_model.model.", ):
            if name.startswith(pre):
                name = name[len(pre):]
        if name not in engine.ps.index and ("model." + name) in engine.ps.index:
            name = "model." + name
        if name in engine.ps.index:
            engine.ps.p(name).copy_(v.to(torch.float32).reshape(engine.ps.p(name).shape))
            loaded.append(name)
    engine.ps.refresh_shadow()
    return loaded


def save_checkpoint(engine, out_dir: str) -> None:
    """What the reference's PT trainer leaves in a checkpoint directory that matters for the path (llava_trainer.py:997-1016 +
    HF Trainer._save_checkpoint): `mm_projector.bin` (the adapter, loaded by the IFT stage through --pretrain_mm_mlp_adapter),
    the trained modules under their reference names (heads, special_*_tokens, *_logit_scale, projector; the IFT stage reads them
    through model_name_or_path: ola_vlm_train.py:1015-1021) as `trainable.safetensors`, and the optimizer state for resume
    (`trainer.train(resume_from_checkpoint=...)`, ola_vlm_train.py:1306-1309) as `optimizer.pt`."""
    import os
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    save_mm_projector(engine, os.path.join(out_dir, "mm_projector.bin"))
    save_file({k: v.detach().cpu().contiguous() for k, v in engine.state_dict().items()}, os.path.join(out_dir, "trainable.safetensors"),
              metadata={"format": "pt"})
    torch.save(engine.optimizer_state_dict(), os.path.join(out_dir, "optimizer.pt"))


def load_checkpoint(engine, ckpt_dir: str, resume_optimizer: bool = True) -> List[str]:
    """Inverse of save_checkpoint.  With `resume_optimizer` the fp32 master and AdamW moments are restored, so continuing is bitwise
    identical to never having stopped; without it only the (bf16) weights are loaded, like starting the next stage from this one."""
    import os
    from safetensors.torch import load_file
    loaded = engine.load_state_dict(load_file(os.path.join(ckpt_dir, "trainable.safetensors")), strict=False)
    opt = os.path.join(ckpt_dir, "optimizer.pt")
    if resume_optimizer and os.path.exists(opt):
        engine.load_optimizer_state_dict(torch.load(opt, map_location="cpu"))
    return loaded
