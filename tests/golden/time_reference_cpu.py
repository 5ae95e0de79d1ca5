"""Decode-step time of the UNMODIFIED reference on this container's CPU, next to the numpy oracle on the same cores.

    python tests/golden/time_reference_cpu.py [steps]        (build container only: needs /root/reference)

bench.py's cpu_baseline times the oracle ("port") on the GPU box, where the reference tree does not exist; this script
calibrates that port against the real thing: the reference's ConditionalAutoregressive2D.sample for the level-0
upsampler (72 layers, width 1920, batch 16, fp32, torch CPU kernels on all cores) and oracle.Transformer on the same
weights, same batch, same positions."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refshim  # noqa: E402

refshim.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from jukebox.hparams import setup_hparams  # noqa: E402
from jukebox.make_models import MODELS, make_prior, make_vqvae  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    vq = make_vqvae(setup_hparams(MODELS["1b_lyrics"][0], dict(sample_length=1048576, restore_vqvae="")), "cpu")
    prior = make_prior(setup_hparams("upsampler_level_0", dict(restore_prior="")), vq, "cpu")
    ar = prior.prior
    N, T, W = 16, ar.input_dims, ar.width
    x_cond = torch.zeros(N, T, W)
    y_cond = torch.zeros(N, 1, W)
    with torch.no_grad():
        ar.sample(N, x_cond, y_cond, None, fp16=False, temp=0.99, sample_tokens=2)          # warm-up
        t0 = time.perf_counter()
        ar.sample(N, x_cond, y_cond, None, fp16=False, temp=0.99, sample_tokens=steps)
        ref = (time.perf_counter() - t0) / steps
    print(f"reference (torch {torch.__version__} CPU, {torch.get_num_threads()} threads): {ref * 1e3:.1f} ms per decode step "
          f"(batch {N}, positions 0..{steps - 1}, incl. embedding / logits / sampling)")

    from oracle.transformer import Transformer
    from threadpoolctl import threadpool_limits
    sd = {k[len("transformer."):]: v.numpy() for k, v in ar.state_dict().items() if k.startswith("transformer.")}
    tr = Transformer(sd, "", W, T, 1, 72, attn_order=2, blocks=128)       # upsampler_level_0 (hparams.py)
    x = np.random.default_rng(0).standard_normal((N, 1, W)).astype(np.float32)
    with threadpool_limits(limits=os.cpu_count()):
        tr.forward(x)
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.forward(x)
        port = (time.perf_counter() - t0) / steps
    print(f"oracle port (numpy {np.__version__}, {os.cpu_count()} threads): {port * 1e3:.1f} ms per decode step (transformer only)")
    from oracle.torch_port import TorchDecodeStack
    tp = TorchDecodeStack(sd, "", W, T, 1, 72, attn_order=2, blocks=128, n_batch=N)
    tp.forward(x)
    t0 = time.perf_counter()
    for _ in range(steps):
        tp.forward(x)
    tport = (time.perf_counter() - t0) / steps
    print(f"torch port (oracle/torch_port.py, {torch.get_num_threads()} threads): {tport * 1e3:.1f} ms per decode step (transformer only)")
    print(f"numpy port / reference = {port / ref:.2f}   torch port / reference = {tport / ref:.2f}")


if __name__ == "__main__":
    main()
