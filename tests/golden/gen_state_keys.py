"""Parameter trees of the RELEASED model configurations, taken from the unmodified reference.

    python tests/golden/gen_state_keys.py        (build container only: needs /root/reference)

Instantiates the reference's `make_vqvae` / `make_prior` for every entry of MODELS['1b_lyrics' | '5b' | '5b_lyrics']
(jukebox/make_models.py:17-22) on the meta device (no storage, so the 5-billion-parameter priors cost nothing) and
writes, per model, the ordered list of (state_dict key, shape, dtype) to tests/golden/state_keys.json.
tests/test_host_cpu.py::test_released_parameter_trees compares the mirror's modules against it: that is the
"checkpoints load unchanged" contract at the real sizes, checked without weights."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
import torch  # noqa: E402
from jukebox.hparams import setup_hparams  # noqa: E402
from jukebox.make_models import MODELS, make_prior, make_vqvae  # noqa: E402


def tree(module):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in module.state_dict().items()]


def main():
    out = {}
    with torch.device("meta"):
        vq_name = MODELS["5b"][0]
        vq = make_vqvae(setup_hparams(vq_name, dict(sample_length=1048576, restore_vqvae="")), "meta")
        out[vq_name] = tree(vq)
        names = []
        for model in ("1b_lyrics", "5b", "5b_lyrics"):
            for nm in MODELS[model][1:]:
                if nm not in names:
                    names.append(nm)
        for nm in names:
            prior = make_prior(setup_hparams(nm, dict(restore_prior="")), vq, "meta")
            out[nm] = tree(prior)
            print(nm, len(out[nm]), "tensors", sum(int(torch.Size(s).numel()) for _, s, _ in out[nm]) / 1e6, "M elements")
    json.dump(out, open(os.path.join(HERE, "state_keys.json"), "w"), separators=(",", ":"))


if __name__ == "__main__":
    main()
