"""One tiny PT train step on cuda:0 through the C ABI, checked against the CPU oracle (used by
__graft_entry__.smoke()).  The oracle is only the checker here."""
import torch


def run(verbose=True):
    from oracle import cases, visper_oracle as O          # test infrastructure: checker only
    from .config import VisperConfig
    from .engine import Engine

    ocfg, W, batch, g = cases.tiny_llama_case()
    cfg = VisperConfig(**{k: v for k, v in vars(ocfg).items()})
    eng = Engine(cfg)
    eng.load_weights(W)
    gb = {k: (v.cuda() if k in ("images",) or k.endswith("_target") or k.endswith("_mask") else v) for k, v in batch.items()}
    out = eng.train_step(gb)
    torch.cuda.synchronize()
    loss = float(out["loss"])
    # oracle in the same dtype policy (bf16 params/activations, fp32 softmax/norm/loss)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    bb = {k: (v.to(torch.bfloat16) if v.is_floating_point() and not k.endswith("_mask") else v) for k, v in batch.items()}
    ref = O.forward(Wb, bb, ocfg, need_logits=False)
    ref_loss, gold = float(ref["loss"]), float(g["keep_loss"])
    if verbose:
        print(f"smoke: HIP loss {loss:.5f} | oracle bf16 {ref_loss:.5f} | reference fp32 golden {gold:.5f}")
    assert abs(loss - ref_loss) <= 2e-2 * abs(ref_loss), (loss, ref_loss)
    assert abs(loss - gold) <= 2e-2 * abs(gold), (loss, gold)
    gn = float(eng.ps.grad.norm())
    assert gn > 0 and gn == gn, gn
    return loss, ref_loss, gold
