"""Label vectors `y` for the priors (jukebox/data/labels.py).  y = [total_length, offset, sample_length,
artist_id, genre_ids..., lyric_tokens...] (int64).

Artist / genre / character vocabularies live in the reference's `jukebox/data/ids/*.txt`, which are data files
of the reference distribution and are not copied here: `Labeller.get_label` needs `JUKEBOX_IDS_DIR` to point at
them; `get_y_from_ids` / `get_batch_labels_from_ids` (what the benchmark and the tests use) need nothing."""
import os
import re

import numpy as np
import torch as t


def get_relevant_lyric_tokens(full_tokens, n_tokens, total_length, offset, duration, f32=False):
    """labels.py:7-20: a window of n_tokens lyric characters centred on the audio window's midpoint (left-padded
    with 0 when the lyrics are short).  f32: the per-window re-labelling (set_y_lyric_tokens) feeds 0-dim int64
    tensors, so the reference evaluates the midpoint in float32 -- mirrored here for sample-exact windows."""
    if len(full_tokens) < n_tokens:
        tokens = [0] * (n_tokens - len(full_tokens)) + full_tokens
        indices = [-1] * (n_tokens - len(full_tokens)) + list(range(0, len(full_tokens)))
    else:
        assert 0 <= offset < total_length
        if f32:
            f = np.float32
            midpoint = int(f(len(full_tokens)) * (f(offset) + f(duration) / f(2.0)) / f(total_length))
        else:
            midpoint = int(len(full_tokens) * (offset + duration / 2.0) / total_length)
        midpoint = min(max(midpoint, n_tokens // 2), len(full_tokens) - n_tokens // 2)
        tokens = full_tokens[midpoint - n_tokens // 2:midpoint + n_tokens // 2]
        indices = list(range(midpoint - n_tokens // 2, midpoint + n_tokens // 2))
    assert len(tokens) == n_tokens and len(indices) == n_tokens
    return tokens, indices


# where unidecode's tables (x000, x001, x020) differ from "compatibility-decompose and drop what is not ASCII": letters with a
# stroke (no decomposition), ligatures, Latin-1 symbols, dashes and quotes
_PUNCT = {"\u2014": "--", "\u2013": "-", "\u2018": "'", "\u2019": "'", "\u201c": '"', "\u201d": '"', "\u2026": "...",
          "\u00df": "ss", "\u00e6": "ae", "\u00c6": "AE", "\u00f8": "o", "\u00d8": "O", "\u0153": "oe", "\u0152": "OE",
          "\u00d0": "D", "\u00f0": "d", "\u00de": "Th", "\u00fe": "th", "\u0110": "D", "\u0111": "d", "\u0126": "H", "\u0127": "h",
          "\u0131": "i", "\u0138": "k", "\u0141": "L", "\u0142": "l", "\u0149": "'n", "\u014a": "NG", "\u014b": "ng", "\u0166": "T",
          "\u0167": "t", "\u00a1": "!", "\u00a2": "C/", "\u00a3": "PS", "\u00a4": "$?", "\u00a5": "Y=", "\u00a6": "|", "\u00a7": "SS",
          "\u00a8": '"', "\u00a9": "(c)", "\u00ab": "<<", "\u00ac": "!", "\u00ae": "(r)", "\u00af": "-", "\u00b0": "deg",
          "\u00b1": "+-", "\u00b4": "'", "\u00b5": "u", "\u00b6": "P", "\u00b7": "*", "\u00b8": ",", "\u00bb": ">>", "\u00bc": "1/4",
          "\u00bd": "1/2", "\u00be": "3/4", "\u00bf": "?", "\u00d7": "x", "\u00f7": "/", "\u2010": "-", "\u2011": "-", "\u2012": "-",
          "\u2015": "--", "\u201a": ",", "\u201b": "'", "\u201e": ",,", "\u201f": '"', "\u2022": "*", "\u2032": "'", "\u2033": '"',
          "\u2039": "<", "\u203a": ">"}


def to_ascii(text):
    """Stand-in for `unidecode` (data/text_processor.py:12, not installed here): compatibility-decompose and drop the
    combining marks (e-acute -> e), plus the handful of punctuation marks / ligatures lyric sheets actually contain.
    Identical to unidecode on ASCII, Latin-1 Supplement, Latin Extended-A and the dashes / quotes of General Punctuation
    (tests/test_oracle_golden.py compares it, character by character, with an independent restatement of unidecode's tables);
    other scripts are dropped instead of romanised."""
    import unicodedata
    text = "".join(_PUNCT.get(c, c) for c in text)
    return unicodedata.normalize("NFKD", text).encode("ascii", "ignore").decode()


class EmptyLabeller:
    def get_label(self, artist=None, genre=None, lyrics=None, total_length=None, offset=None):
        return dict(y=np.array([], dtype=np.int64), info=dict(artist="n/a", genre="n/a", lyrics=[], full_tokens=[]))

    def get_batch_labels(self, metas, device="cpu"):
        ys = t.zeros((len(metas), 0), dtype=t.long, device=device)
        return dict(y=ys, info=[self.get_label()["info"] for _ in metas])


class TextProcessor:
    """data/text_processor.py: character vocabulary (v3: 79 symbols incl. <unk>=0)."""

    def __init__(self, v3=False):
        extra = "" if v3 else "+"
        vocab = ("ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789.,:;!?-" + extra + "'\"()[] \t\n")
        self.not_vocab = re.compile("[^A-Za-z0-9.,:;!?\\-" + ("" if v3 else "+") + "'\"()\\[\\] \t\n]+")
        self.vocab = {c: i + 1 for i, c in enumerate(vocab)}
        self.vocab["<unk>"] = 0
        self.n_vocab = len(vocab) + 1
        self.tokens = {v: k for k, v in self.vocab.items()}
        self.tokens[0] = ""

    def clean(self, text):
        text = to_ascii(text)
        text = text.replace("\\", "\n")
        return self.not_vocab.sub("", text)

    def tokenise(self, text):
        return [self.vocab[c] for c in text]

    def textise(self, tokens):
        return "".join(self.tokens[tok] for tok in tokens)


class ArtistGenreProcessor:
    """data/artist_genre_processor.py:27-100: name -> id tables read from the reference's `ids/v{2,3}_{artist,genre}_ids.txt`
    (data files of the released checkpoints, not redistributed here): `ids_dir`, else $JUKEBOX_IDS_DIR, else an `ids/`
    directory next to this file, else $JUKEBOX_REFERENCE/jukebox/data/ids.  Looking a NAME up without tables raises -- the reference cannot run without them
    either, and silently conditioning every sample on id 0 would be wrong audio without a warning.  With tables, an unknown
    name maps to id 0 with the reference's message (artist_genre_processor.py:45-47,57-60).  Ids can always be given
    directly (Labeller.get_batch_labels_from_ids)."""

    def __init__(self, v3=False, ids_dir=None):
        self.v3 = v3
        here_ids = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ids")
        ref_ids = os.path.join(os.environ.get("JUKEBOX_REFERENCE", ""), "jukebox", "data", "ids")     # a checkout of the reference
        ids_dir = ids_dir or os.environ.get("JUKEBOX_IDS_DIR") or next((d for d in (here_ids, ref_ids) if os.path.isdir(d)), None)
        self.ids_dir = ids_dir
        self.artist_ids, self.genre_ids = {}, {}
        ver = "v3" if v3 else "v2"
        self.artist_id_file = os.path.join(ids_dir or "<ids dir>", f"{ver}_artist_ids.txt")
        self.genre_id_file = os.path.join(ids_dir or "<ids dir>", f"{ver}_genre_ids.txt")
        if ids_dir:
            for attr, fn in (("artist_ids", self.artist_id_file), ("genre_ids", self.genre_id_file)):
                with open(fn, encoding="utf-8") as f:
                    for line in f:
                        name, idx = line.strip().split(";")
                        getattr(self, attr)[name.lower()] = int(idx)

    @staticmethod
    def _norm(s):
        s = "".join(c if c.isascii() and c.isalnum() else "_" for c in s.lower())
        return re.sub(r"_+", "_", s).strip("_")

    def _require_tables(self, what):
        if not self.ids_dir:
            raise RuntimeError(f"cannot map {what} to an id: the artist / genre id tables are not loaded.  Point "
                               "JUKEBOX_IDS_DIR (or ArtistGenreProcessor(ids_dir=...)) at the reference's jukebox/data/ids "
                               "directory, or label the batch with ids (Labeller.get_batch_labels_from_ids)")

    def get_artist_id(self, artist):
        self._require_tables(f"artist {artist!r}")
        key = artist.lower() if self.v3 else self._norm(artist)
        if key not in self.artist_ids:
            print(f"Input artist {artist} maps to {key}, which is not present in {self.artist_id_file}. "
                  f"Defaulting to (artist_id, artist) = (0, unknown), if that seems wrong please format artist correctly")
        return self.artist_ids.get(key, 0)

    def get_genre_ids(self, genre):
        self._require_tables(f"genre {genre!r}")
        genres = [genre.lower()] if self.v3 else self._norm(genre).split("_")
        for word in genres:
            if word not in self.genre_ids:
                print(f"Input genre {genre} maps to the list {genres}. {word} is not present in {self.genre_id_file}. "
                      f"Defaulting to (word_id, word) = (0, unknown), if that seems wrong please format genre correctly")
        return [self.genre_ids.get(w, 0) for w in genres]


class Labeller:
    def __init__(self, max_genre_words, n_tokens, sample_length, v3=False):
        self.ag_processor = ArtistGenreProcessor(v3)
        self.text_processor = TextProcessor(v3)
        self.n_tokens, self.max_genre_words, self.sample_length = n_tokens, max_genre_words, sample_length
        self.label_shape = (4 + self.max_genre_words + self.n_tokens,)

    def get_label(self, artist, genre, lyrics, total_length, offset):
        artist_id = self.ag_processor.get_artist_id(artist)
        genre_ids = self.ag_processor.get_genre_ids(genre)
        lyrics = self.text_processor.clean(lyrics)
        full_tokens = self.text_processor.tokenise(lyrics)
        tokens, _ = get_relevant_lyric_tokens(full_tokens, self.n_tokens, total_length, offset, self.sample_length)
        y = self.get_y_from_ids(artist_id, genre_ids, tokens, total_length, offset)
        return dict(y=y, info=dict(artist=artist, genre=genre, lyrics=lyrics, full_tokens=full_tokens))

    def get_y_from_ids(self, artist_id, genre_ids, lyric_tokens, total_length, offset):
        assert len(genre_ids) <= self.max_genre_words
        genre_ids = list(genre_ids) + [-1] * (self.max_genre_words - len(genre_ids))
        if self.n_tokens > 0:
            assert len(lyric_tokens) == self.n_tokens
        else:
            lyric_tokens = []
        y = np.array([total_length, offset, self.sample_length, artist_id, *genre_ids, *lyric_tokens], dtype=np.int64)
        assert y.shape == self.label_shape, f"Expected {self.label_shape}, got {y.shape}"
        return y

    def get_batch_labels(self, metas, device="cpu"):
        labels = [self.get_label(**meta) for meta in metas]
        ys = t.stack([t.from_numpy(l["y"]) for l in labels], dim=0).to(device).long()
        return dict(y=ys, info=[l["info"] for l in labels])

    def get_batch_labels_from_ids(self, items, device="cpu"):
        """items: dicts with artist_id, genre_ids, full_tokens, total_length, offset."""
        ys, infos = [], []
        for it in items:
            tokens, _ = get_relevant_lyric_tokens(list(it["full_tokens"]), self.n_tokens, it["total_length"],
                                                  it["offset"], self.sample_length) if self.n_tokens > 0 else ([], [])
            ys.append(self.get_y_from_ids(it["artist_id"], it["genre_ids"], tokens, it["total_length"], it["offset"]))
            infos.append(dict(artist="n/a", genre="n/a", lyrics="", full_tokens=list(it["full_tokens"])))
        return dict(y=t.from_numpy(np.stack(ys)).to(device).long(), info=infos)

    def set_y_lyric_tokens(self, ys, labels):
        """labels.py:89-105: re-window the lyric tokens for the (already updated) offset in ys."""
        info = labels["info"]
        assert ys.shape[0] == len(info)
        if self.n_tokens == 0:
            return None
        tokens_list, indices_list = [], []
        for i in range(ys.shape[0]):
            total_length, offset, duration = int(ys[i, 0]), int(ys[i, 1]), int(ys[i, 2])
            tokens, indices = get_relevant_lyric_tokens(info[i]["full_tokens"], self.n_tokens, total_length, offset,
                                                        duration, f32=True)
            tokens_list.append(tokens)
            indices_list.append(indices)
        ys[:, -self.n_tokens:] = t.tensor(tokens_list, dtype=t.long, device=ys.device)
        return indices_list
