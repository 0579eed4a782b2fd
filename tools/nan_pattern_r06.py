"""Dev tool (gpurun; VP_LIB_PATH = a variant library whose general one-wave-per-SIMD GEMM does NOT claim the whole register file): WHICH elements of
a side-stream GEMM go wrong when the IFT step overlaps the tower with the previous step, and what do they look like?  Every side-stream GEMM of a few
steps is followed (same stream) by a recomputation on the 8-phase kernel (force code 7: bit-identical by the tests); both outputs are kept and compared
after the run.  Prints, for the first bad launches: shape / epilogue, number of differing and NaN elements, and their row / column structure
(256-row tiles, the 128 x 128 wave quarters, 16-row blocks, columns mod 128)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from visper_lm_amd.config import llama3_8b
from visper_lm_amd.engine import Engine
from visper_lm_amd import ops as _ops

dev = torch.device("cuda:0")
L = int(os.environ.get("LAYERS", "4"))
cfg = llama3_8b(aux_mode="", num_task_tokens=0, train_llm=True)
cfg.num_hidden_layers = L
cfg.depth_decoder = False
eng = Engine(cfg, device=dev)
eng.set_distributed(0, 1, transport="torch")
eng.init_random(seed=0)
pool = [bench.make_batch(cfg, 8, 1473, 1000 * j, dev) for j in range(4)]
torch.cuda.synchronize()
kept = []
_gemm = _ops.gemm


def gemm(a, w, bias=None, residual=None, epi=0, out=None, out_f32=False, force_generic=False):
    y = _gemm(a, w, bias=bias, residual=residual, epi=epi, out=out, out_f32=out_f32, force_generic=force_generic)
    if torch.cuda.current_stream() != torch.cuda.default_stream() and not out_f32 and not force_generic and len(kept) < 400:
        ref = _gemm(a, w, bias=bias, residual=residual, epi=epi, force_generic=7)
        kept.append((len(kept), tuple(a.shape), tuple(w.shape), int(epi), bias, residual is not None, y.clone(), ref,
                     torch.isfinite(a.float()).all(), None if residual is None else torch.isfinite(residual.float()).all()))
    return y


_ops.gemm = gemm
gi = torch.Generator().manual_seed(4321)
losses = []
for it in range(int(os.environ.get("STEPS", "5"))):
    b = dict(pool[it % 4])
    ids = torch.randint(0, 1000, (8, 1473), generator=gi); ids[:, cfg.num_sys_tokens] = -200
    lab = ids.clone(); lab[:, :cfg.num_sys_tokens + 7] = -100
    b["input_ids"], b["labels"], b["images_resident"] = ids, lab, True
    out = eng.train_step(b)
    losses.append(out["loss"].clone())
    eng.optimizer_step(lr=1e-3, lr_mult=1.0)
torch.cuda.synchronize()
print("library:", os.environ.get("VP_LIB_PATH", "product"), "| losses:", [round(float(x), 4) for x in losses], "| side-stream GEMMs kept:", len(kept))
nbad = 0
for (j, ash, wsh, epi, hb, hr, y, ref, afin, rfin) in kept:
    d = (y != ref) & ~(torch.isnan(y.float()) & torch.isnan(ref.float()))
    n = int(d.sum())
    if n == 0:
        continue
    nbad += 1
    if nbad > 4:
        continue
    yf = y.float()
    M, N = y.shape[-2], y.shape[-1]
    d2 = d.view(M, N)
    rows, cols = d2.any(1).nonzero().flatten(), d2.any(0).nonzero().flatten()
    bias_t, hb = hb, hb is not None
    print(f"\nGEMM #{j}: a {ash} w {wsh} epi {epi} bias {hb} residual {hr} | inputs finite: a {bool(afin)} res {None if rfin is None else bool(rfin)}")
    print(f"  differing elements {n} of {M * N} ({int(torch.isnan(yf).sum())} NaN, {int(torch.isinf(yf).sum())} inf in the w4 output; ref finite: {bool(torch.isfinite(ref.float()).all())})")
    print("  bad rows:", rows.numel(), "range", int(rows.min()), "..", int(rows.max()), "| per 256-row tile:", sorted(collections.Counter((rows // 256).tolist()).items())[:24])
    print("  bad rows mod 256 per 128-row wave half:", sorted(collections.Counter(((rows % 256) // 128).tolist()).items()), "| per 16-row block:", sorted(collections.Counter(((rows % 128) // 16).tolist()).items()))
    print("  bad cols:", cols.numel(), "range", int(cols.min()), "..", int(cols.max()), "| per 256-col tile:", sorted(collections.Counter((cols // 256).tolist()).items())[:24])
    print("  bad cols mod 256 per 128-col wave half:", sorted(collections.Counter(((cols % 256) // 128).tolist()).items()), "| (col mod 128) // 8:", sorted(collections.Counter(((cols % 128) // 8).tolist()).items()))
    yy, rr_ = y.view(M, N), ref.view(M, N)
    idx = d2.nonzero()
    print("  bad (row % 16) histogram:", sorted(collections.Counter((idx[:, 0] % 16).tolist()).items()), "| bad (col % 8):", sorted(collections.Counter((idx[:, 1] % 8).tolist()).items()),
          "| bad (col % 64) // 8:", sorted(collections.Counter(((idx[:, 1] % 64) // 8).tolist()).items()))
    # hypothesis: a lane of rows 0..7 of a 16-row block took the OTHER source of w4_rows8_swap's v_cndmask_b32_dpp (y of lane fr + 8, i.e. the value that
    # belongs at (row + 8, col + 32)) because it read a stale VCC bit; rows 8..15: the value of (row - 8, col - 32 + 64 ...) likewise
    hit = tot = 0
    for r, c in idx[:4000].tolist():
        tot += 1
        if r % 16 < 8 and c % 64 < 32 and r + 8 < M:
            hit += int(yy[r, c] == rr_[r + 8, c + 32])
        elif r % 16 >= 8 and c % 64 >= 32:
            hit += int(yy[r, c] == rr_[r - 8, c - 32])
    print(f"  wrong value == the reference value of the swap partner (row +- 8, col +- 32): {hit} of {tot}")
    # is the wrong dword a RAW fp32 ACCUMULATOR (an unpacked v_accvgpr_read result)?  Reassemble (col c | col c + 1) as one fp32, add the column's bias and
    # look for a reference output of the same lane (rows r + 16 k of the wave's 128-row quarter, any column of its 128) that it rounds to
    import struct
    offs = collections.Counter()
    bits = lambda t: int(t.view(torch.int16)) & 0xffff
    nchk = 0
    if epi == 0 and not hr:
        for r, c in idx.tolist():
            if c % 2 or nchk >= 60:
                continue
            nchk += 1
            f = struct.unpack("<f", struct.pack("<I", (bits(yy[r, c + 1]) << 16) | bits(yy[r, c])))[0]
            r_lo, c_lo = r - r % 128, c - c % 128
            cand = rr_[r_lo + (r % 16): r_lo + 128: 16, c_lo: c_lo + 128].float()                          # [8 row blocks, 128 cols]
            b_ = bias_t[c_lo: c_lo + 128].float() if bias_t is not None else torch.zeros(128, device=cand.device)
            guess = (torch.tensor(f, device=cand.device) + b_).to(torch.bfloat16).float()[None, :].expand_as(cand)
            hit_ = (guess == cand).nonzero()
            if hit_.numel() and f == f and abs(f) < 1e4:
                for k_, cc_ in hit_[:3].tolist():
                    offs[(k_ - (r % 128) // 16, cc_ - (c - c_lo))] += 1
            else:
                offs["no match"] += 1
        print("  wrong dword read as ONE fp32 (+ bias) == reference output at (row block offset, col offset) of the same lane:", offs.most_common(8))
    r, c = idx[0].tolist()
    print(f"  first bad element ({r}, {c}): w4 {float(yy[r, c]):.4f} ref {float(rr_[r, c]):.4f} | ref at (row+8, col+32): {float(rr_[min(r + 8, M - 1), min(c + 32, N - 1)]):.4f}")
print(f"\n{nbad} of {len(kept)} side-stream GEMM launches differ from their 8-phase recomputation")
