"""Golden vectors for the teacher-forced evaluation pass: SimplePrior.z_forward
(jukebox/prior/prior.py:312-347) and ConditionalAutoregressive2D.forward (autoregressive.py:114-175) of the unmodified
reference on the tiny models whose weights are already committed in priors.npz / prior_sep.npz.

    python tests/golden/gen_forward_golden.py          (build container only: needs /root/reference)

Writes tests/golden/forward.npz: per model the loss, the metrics (bpd, prime_loss, gen_loss) and the logits."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
import torch as t  # noqa: E402
from jukebox.hparams import HPARAMS_REGISTRY, Hyperparams, setup_hparams  # noqa: E402
from jukebox.make_models import make_prior, make_vqvae  # noqa: E402


def main():
    tiny = json.load(open(os.path.join(HERE, "tiny_hps.json")))
    for nm, h in tiny.items():
        HPARAMS_REGISTRY[nm] = Hyperparams({k: (tuple(v) if isinstance(v, list) else v) for k, v in h.items()})
    gv = np.load(os.path.join(HERE, "vqvae.npz"))
    gp = np.load(os.path.join(HERE, "priors.npz"))
    gs = np.load(os.path.join(HERE, "prior_sep.npz"))
    vq = make_vqvae(setup_hparams("tiny_vqvae", dict(restore_vqvae="")), "cpu")
    vq.load_state_dict({k[3:]: t.from_numpy(gv[k]) for k in gv.files if k.startswith("sd.")}, strict=True)
    out = {}

    def run(tag, prior, z, z_conds, y):
        with t.no_grad():
            loss, metrics = prior.z_forward(t.from_numpy(z), [t.from_numpy(c) for c in z_conds], t.from_numpy(y), fp16=False,
                                            get_preds=True)
        out[f"{tag}.loss"] = np.float32(loss.item())
        for k in ("bpd", "prime_loss", "gen_loss"):
            out[f"{tag}.{k}"] = np.float32(metrics[k].item())
        out[f"{tag}.preds"] = metrics["preds"].numpy()
        print(tag, float(loss), {k: float(metrics[k]) for k in ("bpd", "prime_loss", "gen_loss")}, metrics["preds"].shape)

    for i, nm in enumerate(("tiny_up0", "tiny_up1", "tiny_top")):
        p = make_prior(setup_hparams(nm, dict(restore_prior="")), vq, "cpu")
        p.load_state_dict({k[3:]: t.from_numpy(gp[k]) for k in gp.files
                           if k.startswith(f"p{i}.") and not k[3:].startswith(("labels_y", "full_tokens"))}, strict=True)
        if nm == "tiny_top":
            run("top", p, gp["top.z_ancestral"], [], gp["top.y0"])
        else:
            tag = "up0" if nm == "tiny_up0" else "up1"
            run(tag, p, gp[f"{tag}.z"], [gp[f"{tag}.z_cond"]], gp[f"{tag}.y"])
    sep = make_prior(setup_hparams("tiny_sep", dict(restore_prior="")), vq, "cpu")
    sep.load_state_dict({k[3:]: t.from_numpy(gs[k]) for k in gs.files if k.startswith("sd.")}, strict=True)
    # get_prime_loss (prior.py:303-310) calls .view(-1) on the lyric slice of y, which only works for one sample
    run("sep", sep, gs["z_ancestral"][:1], [], gs["y0"][:1])
    np.savez_compressed(os.path.join(HERE, "forward.npz"), **out)


if __name__ == "__main__":
    main()
