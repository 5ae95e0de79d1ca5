"""Golden label vectors from the unmodified reference's Labeller (jukebox/data/labels.py:22-87) for named artists / genres /
lyrics, v2 (5b models, upsamplers) and v3 (1b_lyrics top level) vocabularies.

    python tests/golden/gen_labels_golden.py         (build container only: needs /root/reference)

Writes tests/golden/labels.npz.  tests/test_host_cpu.py::test_labeller_names_match_reference replays the same metas through
jukebox_amd.data.labels.Labeller with JUKEBOX_IDS_DIR pointing at the reference's id tables (skipped where absent)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
from jukebox.data.labels import Labeller  # noqa: E402

METAS = [
    dict(artist="Alan Jackson", genre="Country", lyrics="I met a traveller from an antique land,\nWho said: Two vast & trunkless legs"),
    dict(artist="Joe Bonamassa", genre="Blues Rock", lyrics="It's 5 o'clock -- somewhere; [chorus] (yeah!) \"quoted\" + plus"),
    dict(artist="Frank Sinatra", genre="Classic Pop", lyrics=""),
    dict(artist="Ella Fitzgerald", genre="Jazz", lyrics="x" * 700),
    dict(artist="Céline Dion", genre="Pop", lyrics="Café — déjà vu\\second line\ttab"),
    dict(artist="nobody in particular", genre="Not A Genre At All", lyrics="unknown artist and genre words map to id 0"),
    dict(artist="The Beatles", genre="Psychedelic Rock Pop", lyrics="bag of words: three genre words in v2"),
]


def main():
    out = {}
    for tag, v3, words, n_tok in (("v2", False, 5, 512), ("v3", True, 1, 384)):
        lab = Labeller(words, n_tok, 1048576, v3=v3)
        for i, m in enumerate(METAS):
            r = lab.get_label(total_length=180 * 44100, offset=(i % 3) * 1048576, **m)
            out[f"{tag}.y{i}"] = np.asarray(r["y"], dtype=np.int64)
            out[f"{tag}.full_tokens{i}"] = np.asarray(r["info"]["full_tokens"], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "labels.npz"), **out)
    print({k: v.shape for k, v in list(out.items())[:4]})


if __name__ == "__main__":
    main()
