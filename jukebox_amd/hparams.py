"""Hyper-parameter registry for the sampling path -- same names, defaults and merge rule as the
reference's jukebox/hparams.py (setup_hparams :11-24; model presets :39-304; DEFAULTS :307-567), so that
`setup_hparams("prior_1b_lyrics", {})` yields the dimensions the released checkpoints were trained with.
Only groups that the sampling path reads are carried; training-only groups keep the few keys
make_models.py consults (train, labels, labels_v3, ...).
"""

REMOTE_PREFIX = "https://openaipublic.azureedge.net/"


class Hyperparams(dict):
    def __getattr__(self, attr):
        try:
            return self[attr]
        except KeyError:
            raise AttributeError(attr)

    def __setattr__(self, attr, value):
        self[attr] = value


HPARAMS_REGISTRY = {}
DEFAULTS = {}


def setup_hparams(hparam_set_names, kwargs):
    """Defaults <- named sets (comma separated or tuple) <- kwargs; unknown keys are rejected."""
    H = Hyperparams()
    if not isinstance(hparam_set_names, tuple):
        hparam_set_names = hparam_set_names.split(",")
    sets = [HPARAMS_REGISTRY[x.strip()] for x in hparam_set_names if x] + [kwargs]
    for group in DEFAULTS.values():
        H.update(group)
    for hps in sets:
        for k in hps:
            if k not in H:
                raise ValueError(f"{k} not in default args")
        H.update(**hps)
    H.update(**kwargs)
    return H


def _reg(name, *bases, **kw):
    h = Hyperparams(**kw)
    for b in bases:
        h.update(b)
    HPARAMS_REGISTRY[name] = h
    return h


_reg("teeny")
_reg("easy", sr=22050)

# ---- VQ-VAE (5b / 1b_lyrics share it) ---------------------------------------------------------------
_reg("vqvae", levels=3, downs_t=(3, 2, 2), strides_t=(2, 2, 2), emb_width=64, l_bins=2048, l_mu=0.99, commit=0.02,
     spectral=0.0, multispectral=1.0, hvqvae_multipliers=(2, 1, 1), loss_fn="lmix", lmix_l2=1.0, lmix_linf=0.02,
     width=32, depth=4, m_conv=1.0, dilation_growth_rate=3,
     restore_vqvae=REMOTE_PREFIX + "jukebox/models/5b/vqvae.pth.tar")

_labels = Hyperparams(y_bins=(120, 4111), t_bins=128, max_bow_genre_size=5, n_vocab=80)
_labels_v3 = Hyperparams(y_bins=(604, 7898), t_bins=64, max_bow_genre_size=1, n_vocab=79)

_upsamplers = Hyperparams(n_ctx=8192, prior_width=1920, prior_depth=72, heads=1, attn_order=2, blocks=128,
                          init_scale=0.4, c_res=1, cond_width=1024, cond_depth=16, cond_dilation_growth_rate=3,
                          cond_dilation_cycle=8, cond_c_res=1, use_tokens=False, prime_loss_fraction=0.0,
                          fp16_params=False)
_upsamplers.update(_labels)
_reg("upsampler_level_0", _upsamplers, level=0, restore_prior=REMOTE_PREFIX + "jukebox/models/5b/prior_level_0.pth.tar")
_reg("upsampler_level_1", _upsamplers, level=1, cond_res_scale=True,
     restore_prior=REMOTE_PREFIX + "jukebox/models/5b/prior_level_1.pth.tar")

_reg("prior_5b", _labels, level=2, n_ctx=8192, prior_width=4800, prior_depth=72, heads=8, attn_order=2, blocks=128,
     init_scale=0.1, c_res=1, beta2=0.925, min_duration=60.0, max_duration=600.0, use_tokens=False, n_tokens=0,
     prime_loss_fraction=0.0, merged_decoder=True, fp16_params=True,
     restore_prior=REMOTE_PREFIX + "jukebox/models/5b/prior_level_2.pth.tar")

_reg("prior_5b_lyrics", _labels, level=2, n_ctx=8192, prior_width=4800, prior_depth=79, heads=8, attn_order=10,
     blocks=128, init_scale=0.1, c_res=1, prime_width=1280, prime_depth=18, prime_heads=4, prime_attn_order=2,
     prime_blocks=32, prime_init_scale=0.7, prime_c_res=1, min_duration=23.8, max_duration=600.0, use_tokens=True,
     n_tokens=512, prime_loss_fraction=0.4, merged_decoder=True, fp16_params=True, alignment_layer=68,
     alignment_head=2, restore_prior=REMOTE_PREFIX + "jukebox/models/5b_lyrics/prior_level_2.pth.tar")

_reg("prior_1b_lyrics", _labels_v3, level=2, n_ctx=6144, prior_width=2048, prior_depth=72, heads=2, attn_order=12,
     blocks=64, init_scale=0.2, c_res=1, labels_v3=True, min_duration=17.84, max_duration=600.0, use_tokens=True,
     n_tokens=384, prime_loss_fraction=0.4, single_enc_dec=True, fp16_params=False, alignment_layer=63,
     alignment_head=0, restore_prior=REMOTE_PREFIX + "jukebox/models/1b_lyrics/prior_level_2.pth.tar")

# ---- small models (README recipe) ---------------------------------------------------------------------
_reg("small_vqvae", sr=22050, levels=2, downs_t=(5, 3), strides_t=(2, 2), emb_width=64, l_bins=1024, l_mu=0.99,
     commit=0.02, spectral=0.0, multispectral=1.0, loss_fn="l2", width=32, depth=4, m_conv=1.0,
     dilation_growth_rate=3)
_small_prior = _reg("small_prior", n_ctx=8192, prior_width=1024, prior_depth=48, heads=1, c_res=1, attn_order=2,
                    blocks=64, init_scale=0.7)
_reg("small_labelled_prior", _small_prior, labels=True, labels_v3=True, y_bins=(10, 100), max_bow_genre_size=1,
     min_duration=60.0, max_duration=600.0, t_bins=64)
_reg("small_single_enc_dec_prior", n_ctx=6144, prior_width=1024, prior_depth=48, heads=2, attn_order=12, blocks=64,
     init_scale=0.7, c_res=1, prime_loss_fraction=0.4, single_enc_dec=True, labels=True, labels_v3=True,
     y_bins=(10, 100), max_bow_genre_size=1, min_duration=60.0, max_duration=600.0, t_bins=64, use_tokens=True,
     n_tokens=384, n_vocab=79)
_reg("small_upsampler", n_ctx=8192, prior_width=1024, prior_depth=48, heads=1, c_res=1, attn_order=2, blocks=64,
     init_scale=0.7, cond_width=512, cond_depth=16, cond_dilation_growth_rate=3, cond_dilation_cycle=8, cond_c_res=1)
_reg("all_fp16", fp16=True, fp16_params=True, fp16_opt=True, fp16_scale_window=250)

# ---- defaults ---------------------------------------------------------------------------------------------
DEFAULTS["script"] = Hyperparams(name="", local_path="", local_logdir="logs", seed=0, prior=False, save=True)
DEFAULTS["data"] = Hyperparams(bs=1, bs_sample=1, min_duration=None, max_duration=None, n_tokens=0, n_vocab=0,
                               use_tokens=False)
DEFAULTS["vqvae"] = Hyperparams(restore_vqvae="", levels=2, downs_t=(1, 1), strides_t=(2, 2), hvqvae_multipliers=None,
                                revival_threshold=1.0, emb_width=64, l_bins=512, l_mu=0.99, commit=1.0, spectral=0.0,
                                multispectral=1.0, loss_fn="l2", linf_k=2048, lmix_l1=0.0, lmix_l2=0.0, lmix_linf=0.0,
                                use_bottleneck=True)
DEFAULTS["vqvae_conv_block"] = Hyperparams(depth=3, width=128, m_conv=1.0, dilation_growth_rate=1, dilation_cycle=None,
                                           vqvae_reverse_decoder_dilation=True)
DEFAULTS["prior"] = Hyperparams(restore_prior="", restore_prior_ddp=False, max_bow_genre_size=None, y_bins=0, level=0,
                                cond_levels=None, t_bins=64, y_cond_as_bias=False, copy_input=False,
                                merged_decoder=False, single_enc_dec=False, alignment_layer=None, alignment_head=None)
DEFAULTS["prior_attn_block"] = Hyperparams(n_ctx=1024, prior_depth=3, prior_width=128, heads=1, attn_order=0,
                                           blocks=None, spread=None, attn_dropout=0.0, resid_dropout=0.0,
                                           emb_dropout=0.0, zero_out=False, res_scale=False, pos_init=False,
                                           init_scale=1.0, m_attn=0.25, m_mlp=1.0, c_res=0, c_attn=0, c_mlp=0)
DEFAULTS["cond_conv_block"] = Hyperparams(cond_depth=3, cond_width=128, cond_m_conv=1.0, cond_zero_out=False,
                                          cond_res_scale=False, cond_dilation_growth_rate=1, cond_dilation_cycle=None,
                                          cond_c_res=0)
DEFAULTS["sample"] = Hyperparams(primed_chunk_size=None, selected_artists="", temp_top=1.0, temp_rest=0.99,
                                 sample_length_in_seconds=24, total_sample_length_in_seconds=240)
DEFAULTS["prime"] = Hyperparams(prime_loss_fraction=0.1, restore_decoder="")
DEFAULTS["prime_attn_block"] = Hyperparams(prime_depth=3, prime_width=128, prime_heads=1, prime_attn_order=0,
                                           prime_blocks=None, prime_spread=None, prime_attn_dropout=0.0,
                                           prime_resid_dropout=0.0, prime_emb_dropout=0.0, prime_zero_out=False,
                                           prime_res_scale=False, prime_pos_init=False, prime_init_scale=1.0,
                                           prime_m_attn=0.25, prime_m_mlp=1.0, prime_c_res=0, prime_c_attn=0,
                                           prime_c_mlp=0, prime_rel_attn=False, prime_posemb_timescale=10000)
DEFAULTS["opt"] = Hyperparams(beta2=0.999)
DEFAULTS["fp16"] = Hyperparams(fp16=False, fp16_params=False, fp16_loss_scale=None, fp16_scale_window=1000.0,
                               fp16_opt=False)
DEFAULTS["train_test_eval"] = Hyperparams(labels=True, labels_v3=False, train=False, test=False, sample=False,
                                          sampler="ancestral", codes_logdir="")
DEFAULTS["audio"] = Hyperparams(sr=44100, channels=2, sample_length=0)
DEFAULTS["distributed"] = Hyperparams(bucket=128)
