"""Seeded random-init state dicts for full-size parity cases, generated with numpy's PCG64 (bit-identical on every
machine): the generator scripts in this directory load them into the UNMODIFIED reference, the `-m gpu` tests rebuild
the same arrays on the GPU box, where /root/reference does not exist.  Names follow the reference's state_dict
(SURVEY.md Appendix C)."""
import numpy as np


def prior_state(seed, width, depth, bins, seq, y_cond=False, m_attn=0.25, m_mlp=1.0, scale=0.02):
    """ConditionalAutoregressive2D (jukebox/prior/autoregressive.py:49-93) with tied x_emb / x_out."""
    rng = np.random.Generator(np.random.PCG64(seed))
    S, M = int(m_attn * width), int(m_mlp * width)
    r = lambda *shape, sc=scale: (rng.standard_normal(shape, dtype=np.float32) * np.float32(sc))
    sd = {"x_emb.weight": r(bins, width, sc=0.05), "pos_emb.pos_emb": r(seq, width, sc=0.01)}
    if not y_cond:
        sd["start_token"] = r(1, width, sc=0.01)
    for d in range(depth):
        p = f"transformer._attn_mods.{d}."
        sd[p + "attn.c_attn.w"], sd[p + "attn.c_attn.b"] = r(width, 3 * S), r(3 * S, sc=0.01)
        sd[p + "attn.c_proj.w"], sd[p + "attn.c_proj.b"] = r(S, width), r(width, sc=0.01)
        sd[p + "mlp.c_fc.w"], sd[p + "mlp.c_fc.b"] = r(width, M), r(M, sc=0.01)
        sd[p + "mlp.c_proj.w"], sd[p + "mlp.c_proj.b"] = r(M, width), r(width, sc=0.01)
        for ln in ("ln_0", "ln_1"):
            sd[p + ln + ".weight"] = 1 + r(width, sc=0.05)
            sd[p + ln + ".bias"] = r(width, sc=0.02)
    sd["x_out.weight"] = sd["x_emb.weight"]
    return sd
