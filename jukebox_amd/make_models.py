"""Model factory and checkpoint I/O with the reference's signatures (jukebox/make_models.py:17-196):
MODELS, load_checkpoint, restore_model, make_vqvae, make_prior, make_model.  Checkpoints are the reference's
`t.save({'hps','model','opt','step',...})` files and load with strict key matching."""
import os

import numpy as np
import torch as t

from .hparams import REMOTE_PREFIX, setup_hparams
from .utils import dist_adapter as dist
from .utils.dist_utils import print_all
from .utils.torch_utils import freeze_model
from .vqvae.vqvae import calculate_strides

MODELS = {
    "5b": ("vqvae", "upsampler_level_0", "upsampler_level_1", "prior_5b"),
    "5b_lyrics": ("vqvae", "upsampler_level_0", "upsampler_level_1", "prior_5b_lyrics"),
    "1b_lyrics": ("vqvae", "upsampler_level_0", "upsampler_level_1", "prior_1b_lyrics"),
}


def load_checkpoint(path):
    """make_models.py:24-39.  Remote paths resolve to ~/.cache (no download here: there is no network; place the
    file there).  One barrier so that a rank-0 copy step elsewhere has finished."""
    restore = path
    if restore.startswith(REMOTE_PREFIX):
        restore = os.path.join(os.path.expanduser("~/.cache"), restore[len(REMOTE_PREFIX):])
        if not os.path.exists(restore):
            raise FileNotFoundError(f"{restore} not found (checkpoint of {path}); downloads are not performed")
    dist.barrier()
    checkpoint = t.load(restore, map_location=t.device("cpu"), weights_only=False)
    print(f"Restored from {restore}")
    return checkpoint


def restore_model(hps, model, checkpoint_path):
    """make_models.py:52-62."""
    model.step = 0
    if checkpoint_path != "":
        checkpoint = load_checkpoint(checkpoint_path)
        checkpoint["model"] = {k[7:] if k[:7] == "module." else k: v for k, v in checkpoint["model"].items()}
        model.load_state_dict(checkpoint["model"])
        if "step" in checkpoint:
            model.step = checkpoint["step"]


def make_vqvae(hps, device="cuda"):
    """make_models.py:73-110 (eval path)."""
    from .vqvae.vqvae import VQVAE
    block_kwargs = dict(width=hps.width, depth=hps.depth, m_conv=hps.m_conv, dilation_growth_rate=hps.dilation_growth_rate,
                        dilation_cycle=hps.dilation_cycle, reverse_decoder_dilation=hps.vqvae_reverse_decoder_dilation)
    if not hps.sample_length:
        assert hps.sample_length_in_seconds != 0
        top_raw_to_tokens = int(np.prod(calculate_strides(hps.strides_t, hps.downs_t)))
        hps.sample_length = (hps.sample_length_in_seconds * hps.sr // top_raw_to_tokens) * top_raw_to_tokens
        print(f"Setting sample length to {hps.sample_length} (i.e. {hps.sample_length / hps.sr} seconds) to be multiple of {top_raw_to_tokens}")
    vqvae = VQVAE(input_shape=(hps.sample_length, 1), levels=hps.levels, downs_t=hps.downs_t, strides_t=hps.strides_t,
                  emb_width=hps.emb_width, l_bins=hps.l_bins, mu=hps.l_mu, commit=hps.commit, spectral=hps.spectral,
                  multispectral=hps.multispectral, multipliers=hps.hvqvae_multipliers, use_bottleneck=hps.use_bottleneck,
                  **block_kwargs)
    vqvae = vqvae.to(device)
    restore_model(hps, vqvae, hps.restore_vqvae)
    assert not (hps.train and not hps.prior), "training is out of scope"
    print_all("Loading vqvae in eval mode")
    vqvae.eval()
    freeze_model(vqvae)
    return vqvae


def make_prior(hps, vqvae, device="cuda"):
    """make_models.py:112-187."""
    from .prior.prior import SimplePrior
    from .transformer.ops import _convert_conv_weights_to_fp16
    prior_kwargs = dict(input_shape=(hps.n_ctx,), bins=vqvae.l_bins, width=hps.prior_width, depth=hps.prior_depth,
                        heads=hps.heads, attn_order=hps.attn_order, blocks=hps.blocks, spread=hps.spread,
                        attn_dropout=hps.attn_dropout, resid_dropout=hps.resid_dropout, emb_dropout=hps.emb_dropout,
                        zero_out=hps.zero_out, res_scale=hps.res_scale, pos_init=hps.pos_init, init_scale=hps.init_scale,
                        m_attn=hps.m_attn, m_mlp=hps.m_mlp, checkpoint_res=0, checkpoint_attn=0, checkpoint_mlp=0)
    x_cond_kwargs = dict(out_width=hps.prior_width, init_scale=hps.init_scale, width=hps.cond_width, depth=hps.cond_depth,
                         m_conv=hps.cond_m_conv, dilation_growth_rate=hps.cond_dilation_growth_rate,
                         dilation_cycle=hps.cond_dilation_cycle, zero_out=hps.cond_zero_out, res_scale=hps.cond_res_scale,
                         checkpoint_res=hps.cond_c_res)          # keeps the reference's `blocks.*` names
    y_cond_kwargs = dict(out_width=hps.prior_width, init_scale=hps.init_scale, y_bins=hps.y_bins, t_bins=hps.t_bins,
                         sr=hps.sr, min_duration=hps.min_duration, max_duration=hps.max_duration,
                         max_bow_genre_size=hps.max_bow_genre_size)
    if hps.use_tokens and not hps.single_enc_dec:
        prime_kwargs = dict(use_tokens=hps.use_tokens, prime_loss_fraction=hps.prime_loss_fraction, n_tokens=hps.n_tokens,
                            bins=hps.n_vocab, width=hps.prime_width, depth=hps.prime_depth, heads=hps.prime_heads,
                            attn_order=hps.prime_attn_order, blocks=hps.prime_blocks, spread=hps.prime_spread,
                            attn_dropout=hps.prime_attn_dropout, resid_dropout=hps.prime_resid_dropout,
                            emb_dropout=hps.prime_emb_dropout, zero_out=hps.prime_zero_out, res_scale=hps.prime_res_scale,
                            pos_init=hps.prime_pos_init, init_scale=hps.prime_init_scale, m_attn=hps.prime_m_attn,
                            m_mlp=hps.prime_m_mlp, checkpoint_res=0, checkpoint_attn=0, checkpoint_mlp=0)
    else:
        prime_kwargs = dict(use_tokens=hps.use_tokens, prime_loss_fraction=hps.prime_loss_fraction, n_tokens=hps.n_tokens,
                            bins=hps.n_vocab)
    rescale = lambda z_shape: (z_shape[0] * hps.n_ctx // vqvae.z_shapes[hps.level][0],)
    z_shapes = [rescale(z_shape) for z_shape in vqvae.z_shapes]
    prior = SimplePrior(z_shapes=z_shapes, l_bins=vqvae.l_bins, encoder=vqvae.encode, decoder=vqvae.decode, level=hps.level,
                        downs_t=vqvae.downs_t, strides_t=vqvae.strides_t, labels=hps.labels, prior_kwargs=prior_kwargs,
                        x_cond_kwargs=x_cond_kwargs, y_cond_kwargs=y_cond_kwargs, prime_kwargs=prime_kwargs,
                        copy_input=hps.copy_input, labels_v3=hps.labels_v3, merged_decoder=hps.merged_decoder,
                        single_enc_dec=hps.single_enc_dec)
    prior.alignment_head = hps.get("alignment_head", None)
    prior.alignment_layer = hps.get("alignment_layer", None)
    if hps.fp16_params:
        print_all("Converting to fp16 params")
        prior.apply(_convert_conv_weights_to_fp16)
    prior = prior.to(device)
    restore_model(hps, prior, hps.restore_prior)
    assert not hps.train, "training is out of scope"
    print_all("Loading prior in eval mode")
    prior.eval()
    freeze_model(prior)
    return prior


def make_model(model, device, hps, levels=None):
    """make_models.py:189-196: the VQ-VAE on `device`, the priors on CPU (moved per level by the sampler)."""
    vqvae, *priors = MODELS[model]
    vqvae = make_vqvae(setup_hparams(vqvae, dict(sample_length=hps.get("sample_length", 0),
                                                  sample_length_in_seconds=hps.get("sample_length_in_seconds", 0))), device)
    hps.sample_length = vqvae.sample_length
    if levels is None:
        levels = range(len(priors))
    priors = [make_prior(setup_hparams(priors[level], dict()), vqvae, "cpu") for level in levels]
    return vqvae, priors
