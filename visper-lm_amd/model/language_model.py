"""OlaLlavaLlamaForCausalLM / OlaLlavaPhi3ForCausalLM / BaseOLA_VLM mirrors
(ola_vlm/model/language_model/ola_llama.py:39-247, ola_phi3.py, base_ola_vlm.py:38-168,289-320,413-534).

The modules hold nn.Parameters under the reference's state-dict names (so reference checkpoints load with
`load_state_dict`) and expose the reference's forward signature.  The whole forward+backward of a step runs on the
HIP engine inside ONE autograd node: `out.loss.backward()` delivers `.grad` for the PT-stage trainable set
(projector, heads, task tokens, logit scales) exactly like the reference's autograd would."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ..config import VisperConfig, phi3_mini
from ..engine import Engine, is_trainable
from ..params import param_shapes, init_value
from .builders import ParamTree, CLIPVisionTower, CLIPConvNextVisionTower
from .ola_arch import OlaLlavaMetaModel, OlaLlavaMetaForCausalLM


class _Lazy:
    """A field value that is computed on first read (see _ModelOutput)."""

    def __init__(self, fn):
        self.fn = fn


class _ModelOutput:
    """The slice of transformers.utils.ModelOutput that trainers rely on: `out["loss"]`, `out[0]`, `"loss" in out`, `.to_tuple()`
    (HF Trainer.compute_loss reads `outputs["loss"] if isinstance(outputs, dict) else outputs[0]`).
    A field may hold a `_Lazy`: it is materialised the first time it is READ (attribute, key, index, to_tuple) and is a plain tensor from
    then on.  The training forward uses it for `logits`: the reference returns fp32 logits of all B x S rows from every call
    (ola_llama.py:121-122), 8.4 GB and a 5.6 TFLOP lm_head pass over the label-less rows per step at configs[1], which a trainer that reads
    `.loss` never touches — the tensor it WOULD read is bit-identical either way (frozen lm_head, final hidden state captured by the closure)."""

    def __getattribute__(self, name):
        v = object.__getattribute__(self, name)
        if isinstance(v, _Lazy):
            v = v.fn()
            object.__setattr__(self, name, v)
        return v

    def _present(self):
        return [f.name for f in dataclasses.fields(self) if object.__getattribute__(self, f.name) is not None]

    def to_tuple(self):
        return tuple(getattr(self, n) for n in self._present())

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        names = self._present()[k]                     # an index / slice picks its fields first: out[0] (the loss) materialises nothing else
        return getattr(self, names) if isinstance(names, str) else tuple(getattr(self, n) for n in names)

    def __contains__(self, k):
        return isinstance(k, str) and k in self._present()

    def keys(self):
        return self._present()


@dataclass
class OlaCausalLLMOutputWithPast(_ModelOutput):
    """ola_llama.py:39-44 (CausalLMOutputWithPast + the four embedding fields)."""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Optional[Tuple] = None
    hidden_states: Optional[Tuple] = None
    attentions: Optional[Tuple] = None
    image_embs: Optional[List[torch.Tensor]] = None
    seg_embs: Optional[List[torch.Tensor]] = None
    depth_embs: Optional[List[torch.Tensor]] = None
    depth_preds: Optional[List[torch.Tensor]] = None


class OlaLlavaLlamaConfig(VisperConfig):
    model_type = "ola_llama"


class OlaLlavaPhi3Config(VisperConfig):
    model_type = "ola_phi3"

    def __init__(self, **kw):
        super().__init__(**{**phi3_mini().to_dict(), **kw})


class OlaLlavaLlamaModel(OlaLlavaMetaModel, ParamTree):
    config_class = OlaLlavaLlamaConfig


class OlaLlavaPhi3Model(OlaLlavaMetaModel, ParamTree):
    config_class = OlaLlavaPhi3Config


def _run_engine(module, eng, batch, labels, output_hidden_states, kwargs, force_states=True):
    """One engine step behind the reference's `forward` contract (shared by the PT and IFT mirrors): returns (loss, engine outputs,
    logits, hidden_states).  Reference: logits fp32 [B, S, V] always, every decoder-layer state (ola_llama.py:113-122, 181).
    Training calls (labels given) of a model whose lm_head is frozen get the logits as a `_Lazy` (computed from the captured final hidden state
    when first read: same bits); label-less calls, `output_logits=True` and the IFT classes (lm_head trains: a later read would see updated
    weights) compute them inside the step."""
    ref_out = bool(getattr(module.config, "reference_outputs", True))
    eager = labels is None or bool(kwargs.get("output_logits", False)) or (ref_out and bool(getattr(eng, "train_llm", False)))
    eng.keep_logits = True if eager else ("lazy" if ref_out else False)
    eng.keep_states = (ref_out and force_states) or bool(output_hidden_states)
    try:
        if labels is not None:
            loss = _VisperStep.apply(module, batch, *module._trainable_params)
        else:
            module._last = eng.train_step(batch, compute_grads=False)
            loss = None
    finally:
        eng.keep_logits, eng.keep_states = False, False
    out = module._last
    logits = out.pop("logits", None)                                 # fp32 [B, S, V] (ola_llama.py:122), written once by Engine._ntp
    fn = out.pop("logits_fn", None)
    if logits is None and fn is not None:
        logits = _Lazy(fn)
    # popped: `module._last` must not keep the 8.4 GB of logits / the L + 1 layer states alive until the next call
    hidden_states = out.pop("hidden_states", None) or (out["inputs_embeds"], out["hidden"])
    return loss, out, logits, hidden_states


class _VisperStep(torch.autograd.Function):
    """One fused forward+backward on the engine; backward hands the already-computed gradients to autograd."""

    @staticmethod
    def forward(ctx, owner, batch, *params):
        eng = owner._get_engine()
        out = eng.train_step(batch, compute_grads=any(ctx.needs_input_grad[2:]))   # (grad mode is off inside Function.forward)
        owner._last = out
        ctx.owner = owner
        ctx.names = owner._trainable_names
        return out["loss"].reshape(()).clone()          # a fresh tensor: trainers scale the loss in place (HF Trainer: `loss *= ...`)

    @staticmethod
    def backward(ctx, gout):
        """d loss / d parameter = the engine's flat fp32 gradient buffer x the incoming scalar.  ONE scale over the flat buffer and ONE cast to
        the bf16 the Parameters are stored in (two launches; round 5 issued a multiply + a cast per parameter, ~260 launches); each
        Parameter's gradient is a view of the result."""
        eng = ctx.owner._get_engine()
        eng.finish_grads()
        ps = eng.ps
        g32 = torch.mul(ps.grad, gout.to(torch.float32))
        g16 = None
        grads = []
        for n, p in zip(ctx.names, ctx.owner._trainable_params):
            if not p.requires_grad:
                grads.append(None)
                continue
            off, cnt, _ = ps.index[n]
            if p.dtype == torch.float32:
                grads.append(g32[off:off + cnt].view(p.shape))
            else:
                if g16 is None:
                    from .. import ops
                    g16 = ops.cast_to_bf16(g32) if p.dtype == torch.bfloat16 else g32.to(p.dtype)
                grads.append(g16[off:off + cnt].view(p.shape))
        return (None, None, *grads)


class BaseOLA_VLM:
    """base_ola_vlm.py:38-168 attribute surface (mode, layer indices, loss weights, logit scales) and the frozen-teacher hooks."""

    def init_heads(self, config):                      # base_ola_vlm.py:104-168 (parameters come from the manifest)
        from ..config import layer_indices
        self.mode = getattr(config, "aux_mode", "gen-depth-seg")
        self.pass_text_to_aux_head = getattr(config, "pass_text_to_aux", True)
        self.use_ce = getattr(config, "use_ce", False)
        self.contrastive_loss_weight = config.contrastive_loss_weight
        if "gen" in self.mode and hasattr(config, "image_gen"):
            self.img_layer_indices = layer_indices(config.image_gen["img_layer_indices"])
            self.img_gen_loss_weight = config.image_gen["img_loss_weight"]
        if "depth" in self.mode and hasattr(config, "image_depth"):
            self.depth_layer_indices = layer_indices(config.image_depth["depth_layer_indices"])
            self.img_depth_loss_weight = config.image_depth["depth_loss_weight"]
            self.use_intermediate_depth = config.image_depth.get("use_intermediate_depth", True)
        if "seg" in self.mode and hasattr(config, "image_seg"):
            self.seg_layer_indices = layer_indices(config.image_seg["seg_layer_indices"])
            self.img_seg_loss_weight = config.image_seg["seg_loss_weight"]

    def init_target_models(self, config):              # base_ola_vlm.py:56-95
        """The frozen teachers' features are inputs of the step: pass *_target tensors, override
        _get_gen_feats/_get_dav2_feats/_get_seg_targets, or attach the batched GPU teachers (attach_teachers) and pass
        pre-processed pixel tensors (gen_pixels / depth_pixels / seg_pixels)."""
        return None

    def attach_teachers(self, depth=None, gen=None, seg=None):
        """depth: teachers.DinoV2DepthTeacher, gen: teachers.ClipImageEmbedTeacher, seg: teachers.SwinSegTeacher (weights loaded).  With a
        teacher attached, forward(..., <task>_pixels=tensor) computes that task's target on the GPU (the reference does it per PIL image
        inside every step: base_ola_vlm.py:323-397; the PIL / cv2 pre-processing itself stays with the caller)."""
        self._teachers = dict(depth=depth, gen=gen, seg=seg)

    def _get_gen_feats(self, pil_images, device):
        raise NotImplementedError("frozen unCLIP teacher is out of scope: pass gen_target= or override _get_gen_feats")

    def _get_dav2_feats(self, pil_images, device):
        raise NotImplementedError("frozen DINOv2 teacher is out of scope: pass depth_target= or override _get_dav2_feats")

    def _get_seg_targets(self, pil_images, seg_preds):
        raise NotImplementedError("frozen OneFormer teacher is out of scope: pass seg_target= or override _get_seg_targets")


_LIVE_MODULES = None          # weak set of EngineModules whose Parameters alias an engine store (see _install_optimizer_hook)


def _install_optimizer_hook():
    """Any torch optimizer step (HF Trainer / accelerate wrap torch.optim.AdamW; some code paths update `param.data`, which does not move
    the autograd version counter) marks every live EngineModule's parameters as externally modified."""
    global _LIVE_MODULES
    if _LIVE_MODULES is not None:
        return
    import weakref
    _LIVE_MODULES = weakref.WeakSet()
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook

        def _mark(optimizer, args, kwargs):
            # only modules whose OWN Parameters this optimizer steps: an unrelated optimizer (another model in the process) must not make
            # _sync_trainable copy the bf16 shadow over the fp32 master the fused AdamW maintains (it would truncate the master to bf16)
            stepped = {id(p) for g in optimizer.param_groups for p in g["params"]}
            for m in list(_LIVE_MODULES):
                if any(id(p) in stepped for p in m.__dict__.get("_trainable_params", ())):
                    m.__dict__["_force_dirty"] = True
        register_optimizer_step_post_hook(_mark)
    except ImportError:                                           # older torch: the version counters alone
        pass


class EngineModule(nn.Module):
    """nn.Module whose parameters carry the reference's state-dict names and whose compute runs on the HIP engine: parameter
    creation from the manifest (params.param_shapes), engine construction, the Parameter <-> flat-store aliasing, optimizer step,
    HF-style save_pretrained / from_pretrained (safetensors shards + config.json)."""
    model_cls = None

    def __init__(self, config, device="cuda", dtype=torch.bfloat16, init="random", seed=0):
        nn.Module.__init__(self)
        self.config = config
        self.vocab_size = config.vocab_size
        self.steps = 0
        self._engine = None
        self._last = None
        self.model = self.model_cls()
        self.model.config = config
        train_llm = bool(getattr(config, "train_llm", False))
        shapes = param_shapes(config, vit_nested=True)
        gen = torch.Generator(device=device).manual_seed(seed) if init == "random" else None
        top = ParamTree()
        for name, shp in shapes.items():
            tgt, rel = (self.model, name[len("model."):]) if name.startswith("model.") else (top, name)
            if rel.startswith("vision_tower.") and "vision_tower" not in self.model._modules:
                tower = (CLIPConvNextVisionTower if config.is_convnext else CLIPVisionTower)(config.mm_vision_tower, args=config)
                tower.__dict__["_owner"] = self          # plain attribute: NOT a registered child (would create a module cycle)
                self.model.add_module("vision_tower", tower)
            tgt.add(rel, shp, device, dtype if len(shp) else torch.float32, requires_grad=is_trainable(name, train_llm))
            if gen is not None:
                p = dict(tgt.named_parameters())[rel]
                p.data.copy_(init_value(name, shp, gen, device, p.dtype))
        for k, m in list(top._modules.items()):                      # image_*_heads, lm_head
            self.add_module(k, m)
        for k, p in list(top._parameters.items()):                   # *_logit_scale
            self.register_parameter(k, p)
        self.model.initialize_special_tokens(config)

    def get_model(self):
        return self.model

    def state_dict(self, *args, **kwargs):
        """nn.Module.state_dict after folding in in-place edits of the Parameters (see _sync_trainable)."""
        if self._engine is not None:
            self._sync_trainable()
        return super().state_dict(*args, **kwargs)

    def _get_engine(self) -> Engine:
        """Builds the engine from this module's state_dict, then makes every TRAINABLE nn.Parameter a VIEW of the engine's flat
        parameter store (bf16 parameters alias the bf16 shadow the kernels read, fp32 ones alias the fp32 master): one copy of the
        weights, `state_dict()` / `save_pretrained` always see what `Engine.optimizer_step` trained, and an external torch optimizer
        stepping the Parameters writes straight into the kernels' weights.  (The 8 B-parameter IFT model therefore costs no second
        copy of the LLM.)"""
        if self._engine is None:
            dev = next(self.parameters()).device
            eng = Engine(self.config, device=dev)
            eng.load_weights({k: v for k, v in nn.Module.state_dict(self).items()})
            self._engine = eng
            self._trainable_names = [n for n in eng.ps.index]
            named = dict(self.named_parameters())
            self._trainable_params = [named[n] for n in self._trainable_names]
            self._alias = []
            for n, p in zip(self._trainable_names, self._trainable_params):
                if p.dtype == torch.bfloat16:
                    p.data = eng.ps.w(n).view(p.shape)
                    self._alias.append("shadow")
                elif p.dtype == torch.float32:
                    p.data = eng.ps.p(n).view(p.shape)
                    self._alias.append("master")
                else:
                    self._alias.append(None)
            self.__dict__["_seen"] = {"step": eng.ps.step, "ver": [p._version for p in self._trainable_params]}
            _install_optimizer_hook()
            _LIVE_MODULES.add(self)
        return self._engine

    def _sync_trainable(self):
        """Parameters that were modified in place since the last sync (their autograd version counter moved: an external
        optimizer.step(), load_state_dict, manual edits) are propagated to the other half of the store: a bf16 Parameter IS the
        shadow -> copy it into the fp32 master; an fp32 Parameter IS the master -> refresh its bf16 shadow.  Kernel-side updates
        (Engine.optimizer_step: fused AdamW writes master and shadow together) need no action.  No-op in steady state."""
        eng = self._get_engine()
        ps = eng.ps
        seen = self._seen
        vers = [p._version for p in self._trainable_params]
        if self.__dict__.pop("_force_dirty", False):              # a torch optimizer stepped since the last sync
            dirty = [i for i, p in enumerate(self._trainable_params) if p.requires_grad]
        else:
            dirty = [i for i, (a, b) in enumerate(zip(vers, seen["ver"])) if a != b]
        if dirty:
            for i in dirty:
                n, p, al = self._trainable_names[i], self._trainable_params[i], self._alias[i]
                if al == "shadow":
                    ps.p(n).copy_(p.detach().reshape(ps.p(n).shape))
                elif al == "master":
                    ps.w(n).copy_(p.detach().reshape(ps.w(n).shape))
                else:
                    ps.p(n).copy_(p.detach().reshape(ps.p(n).shape))
                    ps.w(n).copy_(p.detach().reshape(ps.w(n).shape))
            ps.version += 1                                      # derived caches (transposed dgrad copies) follow the shadow
            if getattr(eng, "train_llm", False):
                eng.refresh_transposes()
            seen["ver"] = vers
        seen["step"] = ps.step

    def optimizer_step(self, lr, **kw):
        """Engine.optimizer_step (fused AdamW on the flat fp32 master + bf16 shadow, DP mean folded in).  The nn.Parameters alias
        that store, so `state_dict()` / `save_pretrained` see the trained weights at once.  The reference leaves this to HF Trainer +
        DeepSpeed (ola_vlm_train.py:1297-1309); an external torch optimizer over `model.parameters()` works too (_sync_trainable)."""
        self._sync_trainable()                                   # fold in external edits first (they would be overwritten otherwise)
        self._get_engine().optimizer_step(lr, **kw)

    def reload_frozen(self):
        """Call after load_state_dict(): rebuilds the engine's fused / pre-transposed frozen weights."""
        self._engine = None
        self.__dict__.pop("_seen", None)


    # ---- HF-style persistence (f-4): what `trainer.save_model` / `from_pretrained(model_name_or_path)` exchange between the stages
    # (ola_vlm_train.py:228-249, 1015-1021; train.py loads the PT output directory the same way)
    def save_pretrained(self, save_directory, max_shard_size=5 * 2 ** 30):
        """config.json + model.safetensors (or model-0000i-of-0000n.safetensors + model.safetensors.index.json above
        `max_shard_size` bytes, the HF sharded layout), every parameter under its reference name in its own dtype."""
        import json
        import os
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        sd = {k: v.detach().contiguous() for k, v in self.state_dict().items()}
        shards, cur, size = [], {}, 0
        for k, v in sd.items():
            nb = v.numel() * v.element_size()
            if cur and size + nb > max_shard_size:
                shards.append(cur)
                cur, size = {}, 0
            cur[k] = v
            size += nb
        shards.append(cur)
        cfgd = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.to_dict().items()}
        cfgd.pop("task_token_layout", None)                            # derived by the model class (PT: pooled; IFT: from task_token_format)
        cfgd["model_type"] = getattr(self.config, "model_type", "ola_llama")
        cfgd["architectures"] = [type(self).__name__]
        with open(os.path.join(save_directory, "config.json"), "w") as fh:
            json.dump(cfgd, fh, indent=1, sort_keys=True)
        if len(shards) == 1:
            save_file({k: v.cpu() for k, v in shards[0].items()}, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
            return
        index = {"metadata": {"total_size": sum(v.numel() * v.element_size() for v in sd.values())}, "weight_map": {}}
        for i, sh in enumerate(shards):
            fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file({k: v.cpu() for k, v in sh.items()}, os.path.join(save_directory, fn), metadata={"format": "pt"})
            for k in sh:
                index["weight_map"][k] = fn
        with open(os.path.join(save_directory, "model.safetensors.index.json"), "w") as fh:
            json.dump(index, fh, indent=1)

    @classmethod
    def from_pretrained(cls, directory, device="cuda", dtype=torch.bfloat16, strict=True, **config_overrides):
        """Inverse of save_pretrained; also reads a directory written by HF `save_pretrained` of the reference classes (same key names;
        keys this configuration does not have — e.g. the PT stage's heads when loading into the IFT class — are skipped unless strict)."""
        import json
        import os
        from safetensors import safe_open
        with open(os.path.join(directory, "config.json")) as fh:
            cfgd = json.load(fh)
        cfgd.pop("architectures", None)
        cfgd.pop("model_type", None)                                 # the loading CLASS names the model type (a PT directory loaded into the
        for k in getattr(cls.config_class, "STORED_KEYS_IGNORED", ()):   # (IFT classes: trainability belongs to the class, not to the checkpoint)
            cfgd.pop(k, None)
        cfgd.update(config_overrides)                                # IFT class is a llava_* model from then on), not the stored string
        config = cls.config_class(**cfgd)
        model = cls(config, device=device, dtype=dtype, init="empty")
        idx = os.path.join(directory, "model.safetensors.index.json")
        files = sorted(set(json.load(open(idx))["weight_map"].values())) if os.path.exists(idx) else ["model.safetensors"]
        own = dict(nn.Module.state_dict(model))
        seen = set()
        with torch.no_grad():
            for fn in files:
                with safe_open(os.path.join(directory, fn), framework="pt", device="cpu") as f:
                    for k in f.keys():
                        if k in own:
                            own[k].copy_(f.get_tensor(k).to(own[k].dtype))
                            seen.add(k)
                        elif strict:
                            raise KeyError(f"unexpected key {k!r} in {fn}")
        missing = [k for k in own if k not in seen]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        model._missing_keys = missing
        return model


class _OlaCausalLMBase(OlaLlavaMetaForCausalLM, BaseOLA_VLM, EngineModule):
    model_cls = OlaLlavaLlamaModel

    def __init__(self, config, device="cuda", dtype=torch.bfloat16, init="random", seed=0):
        # the PT stage always pools the depth / seg task tokens (ola_arch.py:224-254), whatever an IFT-stage class left in a reused config
        config.task_token_layout = "pooled"
        EngineModule.__init__(self, config, device=device, dtype=dtype, init=init, seed=seed)
        self.NUM_SYS_TOKENS = config.num_sys_tokens                   # ola_llama.py:65-69 / ola_phi3.py:68
        self.init_heads(config)

    def _collect_targets(self, pil_images, kw, B, dev):
        t = {}
        for task, getter in (("gen", "_get_gen_feats"), ("depth", "_get_dav2_feats"), ("seg", "_get_seg_targets")):
            if task not in self.config.token_order:
                continue
            if kw.get(f"{task}_target") is not None:
                t[task] = kw[f"{task}_target"]
            elif kw.get(f"{task}_pixels") is not None and getattr(self, "_teachers", {}).get(task) is not None:
                t[task] = self._teachers[task].forward(kw[f"{task}_pixels"])
            elif pil_images is not None:
                if task == "gen":
                    t[task] = self._get_gen_feats(pil_images, dev)
                elif task == "depth":
                    t[task] = self._get_dav2_feats(pil_images, dev)[0][0][0]      # mean-of-4 DINOv2 feature (base_ola_vlm.py:355)
                else:
                    t[task] = self._get_seg_targets(pil_images, None)
        return t

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                image_sizes=None, return_dict=None, pil_images=None, gen_mask=None, seg_mask=None, depth_mask=None, **kwargs):
        """ola_llama.py:190-244 signature, ola_llama.py:170-188 outputs.  Extra kwargs: gen_target / depth_target / seg_target (precomputed
        frozen-teacher features), images_resident (device images already written: the frozen tower may run on the side stream).  With `config.reference_outputs` (default True) the call returns what the reference returns: fp32
        `logits` [B, S, V] always (ola_llama.py:121-122) and all L + 1 `hidden_states` (`output_hidden_states=True` is forced at :113);
        `return_dict=False` gives the reference's tuple.  `config.reference_outputs = False` is the lean mode for training loops that only read
        `loss`: lm_head + CE run on labelled rows only, `logits` is None unless `output_logits=True` / `labels is None`, and `hidden_states`
        = (inputs_embeds, final state) unless `output_hidden_states=True`."""
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("the MI355X path covers the training forward (input_ids + images); generation is out of scope")
        self._sync_trainable()
        eng = self._get_engine()
        dev = eng.dev
        B = input_ids.shape[0]
        images, gsz = eng.image_groups(images)                       # list / 5-D `images` (ola_arch.py:262-275, "flat" merge) -> one 4-D tensor + group sizes
        if images.device != dev and labels is not None:
            # host images (a collator that leaves them on the CPU): copy AND encode them on the engine's side stream — neither depends on the
            # work still queued on the current stream, so the frozen tower of this batch runs beside the previous step's backward (engine._embed)
            with torch.cuda.stream(eng._side_stream()):
                images = images.to(dev, non_blocking=True)
            batch = dict(input_ids=input_ids, attention_mask=attention_mask, labels=labels, images=images, images_resident=True)
        else:
            # images_resident=True (extension kwarg): the caller states the device images were written before anything still pending on the
            # current stream (a dataloader copying on its own stream; bench.py's resident pool) -> the frozen tower runs on the side stream
            batch = dict(input_ids=input_ids, attention_mask=attention_mask, labels=labels, images=images.to(dev),
                         images_resident=bool(kwargs.get("images_resident", False)) and images.device == dev)
        if gsz is not None:
            batch["image_group_sizes"] = gsz
        for task, tg in self._collect_targets(pil_images, kwargs, B, dev).items():
            batch[f"{task}_target"] = tg
            m = {"gen": gen_mask, "seg": seg_mask, "depth": depth_mask}[task]
            batch[f"{task}_mask"] = torch.ones(B, device=dev) if m is None else m
        loss, out, logits, hidden_states = _run_engine(self, eng, batch, labels, output_hidden_states, kwargs)
        embs = out.get("embs", {})
        res = OlaCausalLLMOutputWithPast(loss=loss, logits=logits, hidden_states=hidden_states,
                                         image_embs=embs.get("gen", []), seg_embs=embs.get("seg", []),
                                         depth_embs=out.get("depth_feats") or embs.get("depth", []),
                                         depth_preds=out.get("depth_preds", []))     # filled when config.depth_decoder
        if return_dict is None:                                      # ola_llama.py:102
            return_dict = bool(getattr(self.config, "use_return_dict", True))
        if not return_dict:                                          # ola_llama.py:170-172: (loss,) + (logits,) + outputs[1:]
            return tuple(v for v in (loss, res.logits, hidden_states) if v is not None)
        return res

    _forward = forward


class OlaLlavaLlamaForCausalLM(_OlaCausalLMBase):
    config_class = OlaLlavaLlamaConfig
    model_cls = OlaLlavaLlamaModel


class OlaLlavaPhi3ForCausalLM(_OlaCausalLMBase):
    config_class = OlaLlavaPhi3Config
    model_cls = OlaLlavaPhi3Model
