"""Fixture for the HTML viewer: runs the UNMODIFIED reference `jukebox.save_html.save_html` (build container only, under
refshim) on a small seeded batch and stores what it wrote -- lyrics.json, align.json, the decoded align.png and the audio
samples of every item -- in tests/golden/save_html.npz, next to the inputs.

    python tests/golden/gen_save_html_golden.py
"""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402
from scipy.io import wavfile  # noqa: E402
from jukebox.save_html import save_html  # noqa: E402


def main():
    rng = np.random.default_rng(5)
    bs, total_length, sr = 2, 160, 8000
    lyr = ["Hello, world!\nla la la.....", "one two three four five six"]
    x = torch.from_numpy(rng.uniform(-1, 1, (bs, 2000, 1)).astype(np.float32))
    zs = [torch.zeros(bs, total_length * 16, dtype=torch.long), torch.zeros(bs, total_length * 4, dtype=torch.long),
          torch.zeros(bs, total_length, dtype=torch.long)]
    info = [dict(artist=f"artist {i}", genre=f"genre {i}", lyrics=lyr[i], full_tokens=list(range(len(lyr[i])))) for i in range(bs)]
    aligns = []
    for i in range(bs):
        a = rng.uniform(0, 1, (total_length, len(lyr[i]))) ** 4
        a[:, len(lyr[i]) - (5 if i == 0 else 1):] = 0.0            # trailing lyric columns nobody attended to
        aligns.append(a)
    hps = type("H", (), dict(levels=3, sr=sr))()
    out = dict(x=x.numpy(), total_length=np.int64(total_length), sr=np.int64(sr))
    with tempfile.TemporaryDirectory() as d:
        save_html(d, x, zs, dict(info=info), aligns, hps)
        for i in range(bs):
            out[f"lyrics{i}"] = np.frombuffer(lyr[i].encode(), np.uint8)
            out[f"align{i}"] = aligns[i]
            out[f"shown{i}"] = np.frombuffer(open(f"{d}/item_{i}/lyrics.json", "rb").read(), np.uint8)
            out[f"align_json{i}"] = np.asarray(json.load(open(f"{d}/item_{i}/align.json")), np.uint8)
            png = np.asarray(Image.open(f"{d}/item_{i}/align.png"))
            assert png.shape == (512, 1024) and png.dtype == np.uint8
            out[f"align_png_sha{i}"] = np.frombuffer(hashlib.sha256(png.tobytes()).digest(), np.uint8)   # 0.5 MB each: keep the digest
            out[f"align_png_sub{i}"] = png[::16, ::16].copy()                                            # ... and a coarse view
            rate, wav = wavfile.read(f"{d}/item_{i}/audio.wav")
            assert rate == sr
            out[f"wav{i}"] = wav
        out["index_iframes"] = np.int64(open(f"{d}/index.html").read().count("<iframe"))
    np.savez_compressed(os.path.join(HERE, "save_html.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
