"""Lyric <-> audio alignment from one attention head of the top-level prior, with the reference's signature and
output (jukebox/align.py:15-83): for every hop window the tokens are teacher-forced through the prior and the
softmax of `prior.alignment_layer` / `prior.alignment_head` (music queries x lyric keys) is recorded; hops are stitched
into one (total_length, n_lyric_characters) matrix per item.

The reference runs a training-style forward per item (`z_forward(get_attn_weights=...)`); here the window is one
chunked prefill of the HIP engine with the probability kernel (`jb_attn_probs`) attached to the alignment layer --
the whole batch at once, no (N, H, T, T) tensor is ever materialised."""
import numpy as np
import torch as t

from .engine import PriorEngine
from .utils.sample_utils import get_starts
from .utils.torch_utils import empty_cache


def get_alignment(x, zs, labels, prior, fp16, hps, device="cuda"):
    level = hps.levels - 1
    n_ctx, n_tokens = prior.n_ctx, prior.n_tokens
    assert prior.single_enc_dec, "alignment is implemented for single_enc_dec priors (1b_lyrics)"
    z = zs[level]
    bs, total_length = z.shape[0], z.shape[1]
    padding_length = 0
    if total_length < n_ctx:
        padding_length = n_ctx - total_length
        z = t.cat([z, t.zeros(bs, padding_length, dtype=z.dtype, device=z.device)], dim=1)
        total_length = z.shape[1]
    hop_length = int(hps.hop_fraction[level] * prior.n_ctx)
    alignment_head, alignment_layer = prior.alignment_head, prior.alignment_layer
    ar = prior.prior
    prior.to(device)
    empty_cache()
    # the prior's packed weights are shared with its sampling engines; only caches / work buffers / the recording are new
    eng = PriorEngine(packed=ar.packed(fp16), n_batch=bs, chunk_cap=512, record=(alignment_layer, alignment_head, n_tokens))
    alignment_hops, indices_hops = {}, {}
    with t.no_grad():
        for start in get_starts(total_length, n_ctx, hop_length):
            end = start + n_ctx
            y, indices_hop = prior.get_y(labels, start, get_indices=True)
            assert len(indices_hop) == bs and all(len(ind) == n_tokens for ind in indices_hop)
            x_cond, y_cond, prime = prior.get_cond(None, y.to(device))
            zz, xc = prior.prior_preprocess([prime, z[:, start:end].to(device)], [None, x_cond])
            eng.set_cond(xc, y_cond)
            eng.tokens[:, :zz.shape[1]] = zz
            eng.prefill(0, zz.shape[1])                         # teacher-forced window, probabilities recorded
            w = eng.rec_out[:, n_tokens:n_tokens + n_ctx, :n_tokens]
            alignment_hops[start] = w.float().cpu().numpy()
            indices_hops[start] = indices_hop
    eng.close()
    del eng
    empty_cache()

    alignments = []
    for item in range(bs):
        full_tokens = labels["info"][item]["full_tokens"]
        alignment = np.zeros((total_length, len(full_tokens) + 1))
        for start in reversed(get_starts(total_length, n_ctx, hop_length)):
            indices = indices_hops[start][item]
            assert alignment_hops[start][item].shape == (n_ctx, n_tokens)
            alignment[start:start + n_ctx, indices] = alignment_hops[start][item]
        alignments.append(alignment[:total_length - padding_length, :-1])
    return alignments
