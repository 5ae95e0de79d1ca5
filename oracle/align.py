"""Oracle restatement of jukebox/align.py:15-83 (get_alignment) over the oracle prior.  Test infrastructure only."""
import numpy as np

from .sample import get_starts


def get_alignment(zs, labels_y, full_tokens, prior, get_y, fp16, hop_fraction, alignment_layer, alignment_head):
    """zs: per-level codes; labels_y / full_tokens: the top level's label matrix and lyric token lists;
    get_y(start) -> (y, indices) as prior.get_y(labels, start, get_indices=True) (prior.py:140-156)."""
    level = len(zs) - 1
    n_ctx, n_tokens = prior.n_ctx, prior.n_tokens
    z = np.asarray(zs[level])
    bs, total_length = z.shape
    padding_length = 0
    if total_length < n_ctx:
        padding_length = n_ctx - total_length
        z = np.concatenate([z, np.zeros((bs, padding_length), z.dtype)], axis=1)
        total_length = z.shape[1]
    hop_length = int(hop_fraction[level] * n_ctx)
    tr = prior.prior.transformer
    alignment_hops, indices_hops = {}, {}
    for start in get_starts(total_length, n_ctx, hop_length):
        y, indices_hop = get_y(start)
        x_cond, y_cond, prime = prior.get_cond(None, y)
        zz = np.concatenate([np.asarray(prime) + prior.prior_bins_shift[0], z[:, start:start + n_ctx] + prior.prior_bins_shift[1]], axis=1)
        xc = np.concatenate([np.zeros((bs, n_tokens, x_cond.shape[-1]), np.float32), x_cond], axis=1)
        tr.set_record_attn({alignment_layer})
        prior.prior.forward_logits(zz, xc, y_cond, fp16=fp16)          # z_forward(get_attn_weights=...) prior.py:319-352
        w = tr.ws[alignment_layer][:, alignment_head]
        tr.set_record_attn(None)
        assert w.shape == (bs, n_ctx, n_tokens)
        alignment_hops[start], indices_hops[start] = w, indices_hop
    alignments = []
    for item in range(bs):
        ft = full_tokens[item]
        alignment = np.zeros((total_length, len(ft) + 1))
        for start in reversed(get_starts(total_length, n_ctx, hop_length)):
            alignment[start:start + n_ctx, indices_hops[start][item]] = alignment_hops[start][item]
        alignments.append(alignment[:total_length - padding_length, :-1])
    return alignments
