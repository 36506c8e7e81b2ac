"""A score on MI355X — drop-in for A_score/compute.py.

Same module-level names (base_folder, subfolders, normalize_feat, load_tensors, results) and the same printed lines
(compute.py:84-85).  The reference is a script; importing this module does nothing until `main()` / `compute()` runs, and
`python -m law_of_vision_representation_in_mllms_amd.A_score.compute --base-folder DIR` reproduces the script behaviour.

Arithmetic: per image mean_t max_s cos(other[t], ref[s]) for ref in {clip336, clip224} (compute.py:54-72) in ONE fused
HIP kernel per (encoder, reference) over all images (csrc/ascore.hip) — the [Nt, Nr, D] broadcast product of the
reference is never formed; then python-float means exactly as compute.py:75-81.
With torch.distributed initialised, images are sharded over ranks and the per-encoder sums are all-reduced (RCCL).
"""
import argparse
import os

import torch

# The benchmark of which A score you want to compute (compute.py:7)
base_folder = '/any/path/mmbench'
# The subfolders for features (compute.py:10)
subfolders = ['clip336', 'clip224', 'dino', 'dit', 'imsd', 'openclip', 'sd1.5', 'sd2.1', 'sd3', 'sdxl']
results = {}
# "exact" (default): exact products + fp32 accumulation on fp32-upcast semantics - the parity definition of SURVEY F4.
# "reference": the reference script's IN-DTYPE arithmetic when the dumped tensors are bf16 (every op of compute.py:12-15,54-72 rounds to
# bf16 - what policy/ablations_t.csv was produced with); fp32 tensors score identically in both modes.  VISREP_ASCORE_ARITHMETIC overrides.
arithmetic = os.environ.get("VISREP_ASCORE_ARITHMETIC", "exact")


def normalize_feat(feat, epsilon=1e-10):
    # compute.py:12-15 (kept for API parity; the HIP kernel folds this normalisation into its row scale)
    norms = torch.linalg.norm(feat, dim=-1, keepdim=True)
    return feat / (norms + epsilon)


def load_tensors(subfolder, n=100):
    """compute.py:18-28: tensor_1.pt .. tensor_100.pt; any error -> message + empty list (encoder skipped by the caller).
    The files are read a few at a time; results are consumed in index order, so the FIRST missing / broken file is the one reported,
    as in the reference's sequential loop."""
    from concurrent.futures import ThreadPoolExecutor
    paths = [os.path.join(base_folder, subfolder, f"tensor_{i}.pt") for i in range(1, n + 1)]
    tensors = []
    with ThreadPoolExecutor(max_workers=8) as pool:
        futures = [pool.submit(torch.load, p, map_location="cpu") for p in paths]
        for tensor_path, fut in zip(paths, futures):
            try:
                tensors.append(fut.result())
            except Exception as e:
                print(f"Error loading {tensor_path}: {e}")
                for f in futures:
                    f.cancel()
                return []
    return tensors


def _shard(n, rank, world):
    return list(range(rank, n, world))


def _score_batch(other, ref, other_scale=None, ref_scale=None, arithmetic="exact"):
    """[n, Nt, D] x [n, Nr, D] -> [n] on the device: the HIP kernel (no CPU fallback; tests may monkeypatch this hook)."""
    from .. import ascore_ops
    if other.device.type == "cpu":                # only reachable through an EXPLICIT device="cpu" (compute(device=), --device cpu): the host twin
        if arithmetic == "reference":             # the host twin computes in exact fp32: never print exact scores where the bf16 op chain was asked for
            raise ValueError("arithmetic='reference' (torch's bf16 op chain, rounding by rounding) runs on the GPU kernel only: "
                             "device='cpu' computes the exact-arithmetic score (use arithmetic='exact' there)")
        return ascore_ops.max_cos_mean_cpu(other, ref)
    if arithmetic == "reference" and other.dtype == torch.bfloat16 and ref.dtype == torch.bfloat16:
        return ascore_ops.max_cos_mean(other, ref, arithmetic="reference")        # torch's bf16 op chain, rounding by rounding
    return ascore_ops.max_cos_mean(other, ref, other_scale, ref_scale)


def _row_scales(x):
    """Per-row normalisation factors of a token stack, computed once and handed to every _score_batch call that uses the stack
    (hook: a stand-in that returns None makes _score_batch compute them itself)."""
    from .. import ascore_ops
    if x.device.type == "cpu":                    # host twin (explicit device="cpu"): it normalises inside the call
        return None
    return ascore_ops.row_scales(x)


def _stack(tensors, ids, device):
    return torch.stack([tensors[i].reshape(-1, tensors[i].shape[-1]) for i in ids]).to(device)


def per_image_scores(other_tensors, ref_tensors, idx, device="cuda"):
    """{image index: mean_t max_s cos} for the images in idx (compute.py:54-72 for one reference)."""
    return per_image_scores_multi(other_tensors, [ref_tensors], idx, device)[0]


def per_image_scores_multi(other_tensors, ref_sets, idx, device="cuda", ref_cache=None, arithmetic=None):
    """One {image index: score} per reference set.  Every stack of tokens goes to the device once and is normalised once: the
    encoder's tokens serve all references, and with `ref_cache` (a dict kept by the caller) a reference stack serves all encoders
    - the reference script re-normalises both inside its innermost loop (compute.py:54-56)."""
    mode = globals()["arithmetic"] if arithmetic is None else arithmetic
    if mode not in ("exact", "reference"):
        raise ValueError(f"arithmetic must be 'exact' or 'reference', got {mode!r}")
    outs = [{} for _ in ref_sets]
    groups = {}
    for i in idx:
        key = (tuple(other_tensors[i].shape),) + tuple(tuple(rs[i].shape) for rs in ref_sets)
        groups.setdefault(key, []).append(i)
    for ids in groups.values():
        o = _stack(other_tensors, ids, device)
        ref_arith = mode == "reference" and o.dtype == torch.bfloat16 and o.device.type != "cpu"      # the in-dtype kernel normalises in its own op chain
        o_scale = None if ref_arith else _row_scales(o)
        for k, rs in enumerate(ref_sets):
            ck = (k, id(rs), tuple(ids))
            if ref_cache is not None and ck in ref_cache:
                r, r_scale = ref_cache[ck]
            else:
                r = _stack(rs, ids, device)
                r_scale = _row_scales(r) if (r.dtype == o.dtype and not ref_arith) else None        # mixed dtypes take the fp32 path inside the op
                if ref_cache is not None:
                    ref_cache[ck] = (r, r_scale)
            same = r.dtype == o.dtype
            if mode == "reference":
                s = _score_batch(o, r, o_scale if same else None, r_scale if same else None, arithmetic="reference").double().cpu()
            else:
                s = _score_batch(o, r, o_scale if same else None, r_scale if same else None).double().cpu()
            for j, i in enumerate(ids):
                outs[k][i] = float(s[j])
    return outs


def compute(base=None, subs=None, n_images=100, device="cuda", verbose=True, arithmetic=None):
    """Returns {subfolder: A score}; mirrors the main loop compute.py:30-85.  arithmetic: None = the module-level `arithmetic`."""
    global base_folder, results
    if base is not None:
        base_folder = base
    subs = subfolders if subs is None else subs
    dist = torch.distributed if torch.distributed.is_available() and torch.distributed.is_initialized() else None
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)

    clip336_tensors = load_tensors('clip336', n_images)
    clip224_tensors = load_tensors('clip224', n_images)
    if not clip336_tensors or not clip224_tensors:
        raise ValueError("Failed to load tensors from 'clip336' or 'clip224' subfolder")

    results = {}
    ref_cache = {}                                                  # reference stacks + their row factors, shared by all encoders
    for subfolder in subs:
        other_tensors = load_tensors(subfolder, n_images)
        if not other_tensors:
            print(f"Skipping {subfolder} due to loading error.")
            continue
        n = min(len(clip336_tensors), len(clip224_tensors), len(other_tensors))       # zip() semantics, compute.py:51
        idx = _shard(n, rank, world)
        s336, s224 = per_image_scores_multi(other_tensors, [clip336_tensors, clip224_tensors], idx, device, ref_cache, arithmetic)
        if dist:
            acc = torch.tensor([sum(s336.values()), sum(s224.values()), float(len(idx))], dtype=torch.float64, device=device)
            dist.all_reduce(acc)
            t336, t224, cnt = acc.tolist()
        else:
            t336 = sum(s336[i] for i in range(n))
            t224 = sum(s224[i] for i in range(n))
            cnt = n
        if cnt:
            results[subfolder] = (t336 / cnt + t224 / cnt) / 2
    if verbose and rank == 0:
        for subfolder, avg_similarity in results.items():
            print(f'Average cosine similarity between clip224+clip336 and {subfolder}: {avg_similarity}')
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description="A score (MI355X)")
    ap.add_argument("--base-folder", default=base_folder)
    ap.add_argument("--subfolders", nargs="*", default=None)
    ap.add_argument("--n-images", type=int, default=100)
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cuda = the HIP kernels (fails without a GPU: there is no fallback); cpu = the host twin (plain C++ fp32, explicit opt-in)")
    ap.add_argument("--arithmetic", default=None, choices=["exact", "reference"],
                    help="reference = the script's in-dtype arithmetic on bf16 tensors (per-op bf16 rounding: the published table's numbers)")
    a = ap.parse_args(argv)
    from .. import dist_env
    owned = dist_env.init_from_env()                                  # under torchrun: one process per GPU, images sharded over ranks
    try:
        return compute(a.base_folder, a.subfolders, a.n_images, device=a.device, arithmetic=a.arithmetic)
    finally:
        dist_env.finalize(owned)


if __name__ == "__main__":
    main()
