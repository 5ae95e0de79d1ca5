"""Static HTML viewer of a sampled batch -- the files `jukebox/save_html.py:7-130` writes next to `data.pth.tar`:

    <logdir>/index.html                      one <iframe> per item
    <logdir>/item_<i>/index.html             audio player, artist / genre, the lyric characters as <span id='<i>/<k>'>
    <logdir>/item_<i>/audio.wav              the item's audio
    <logdir>/item_<i>/lyrics.json            the displayed characters (shifted by one, as the model sees them)
    <logdir>/item_<i>/align.png              lyric-to-music alignment picture (512 x 1024, time running left to right)
    <logdir>/item_<i>/align.json             the alignment, one row per 16 top-level tokens, blurred, for the highlight

Same file names, same array contents (tests/test_host_cpu.py pins align.json / lyrics.json / the picture to a fixture
produced by the reference), own implementation: the alignment post-processing is three small array functions and the page
comes from one template; the karaoke highlight is driven by the audio element's `timeupdate` / animation frames."""
import json
import os

import numpy as np

_PAGE_STYLE = ("font-family: sans-serif; font-size: 1.4em; font-weight: bold; text-align: center; max-width:1024px; "
               "width: 100%; margin: auto;")
_HEAD = "<html><head><title>{title}</title></head><body style='" + _PAGE_STYLE + "'>\n<link rel='icon' href='data:;base64,iVBORw0KGgo='>\n"

# Highlight script: row r of align.json belongs to the r-th of `rows` equal slices of the track; character k is tinted by
# its weight in the current row (0 = white-ish text colour ... 23+ = full red), refreshed while the audio plays.
_SCRIPT = """<script>
(function () {{
  const audio = document.getElementById('{wav}'), track = '{track}', rows = {rows};
  let table = null;
  function paint() {{
    if (table === null || audio.paused) return;
    const r = Math.floor(audio.currentTime * rows / audio.duration);
    if (r > 0 && r < rows) {{
      const weights = table[r];
      for (let k = 0; k < weights.length; k++) {{
        const c = Math.max(230 - 10 * weights[k], 0);
        document.getElementById(track + '/' + k).style.color = 'rgb(255,' + c + ',' + c + ')';
      }}
    }}
    window.setTimeout(paint, 50);
  }}
  audio.addEventListener('play', function () {{
    const go = () => paint();
    if (table !== null) return go();
    fetch('{src}').then(resp => resp.json()).then(data => {{ table = data; go(); }}).catch(err => console.log(err.message));
  }});
}})();
</script>
"""


def trim_alignment(alignment, lyrics):
    """Drop the lyric columns after the last one that received any attention (save_html.py:47-57); at least one column
    is kept.  Returns (alignment[:, :n], lyrics[:n])."""
    total_tokens = alignment.shape[1]
    assert len(lyrics) == total_tokens, f"Total_tokens: {total_tokens}, Lyrics Len: {len(lyrics)}. Lyrics: {lyrics}"
    used = np.nonzero(alignment.max(axis=0) > 0)[0]
    n = int(used[-1]) + 1 if len(used) else 1
    return alignment[:, :n], lyrics[:n]


def alignment_picture(alignment):
    """The 8-bit alignment resized to 512 x 1024 and turned so that time runs along x (save_html.py:60-63)."""
    from PIL import Image
    return Image.fromarray(np.uint8(alignment * 255)).resize((512, 1024)).transpose(Image.ROTATE_90)


def alignment_rows(alignment, total_length):
    """The table behind the highlight: one row per 16 music tokens, one column per character, Gaussian-blurred
    (radius 1.5), as nested lists of 0..255 (save_html.py:65-72)."""
    from PIL import Image, ImageFilter
    rows = total_length // 16
    im = Image.fromarray(np.uint8(alignment * 255)).resize((alignment.shape[1], rows))
    return np.asarray(im.filter(ImageFilter.GaussianBlur(radius=1.5))).tolist(), rows


def write_wav(path, wav, sr):
    """soundfile.write(..., format='wav') of the reference (save_html.py:75): float32 samples as they are."""
    from scipy.io import wavfile
    wavfile.write(path, sr, np.asarray(wav, dtype=np.float32))


def save_item(item_dir, item_id, wav, sr, info, total_length, alignment=None):
    """One item's directory (save_html.py:27-130)."""
    os.makedirs(item_dir, exist_ok=True)
    lyrics = info["lyrics"]
    parts = [_HEAD.format(title=item_id)]
    rows = None
    if alignment is not None:
        assert alignment.shape == (total_length, len(info["full_tokens"]))
        alignment, lyrics = trim_alignment(np.asarray(alignment), lyrics)
        alignment_picture(alignment).save(os.path.join(item_dir, "align.png"))
        parts.append("<img id='align.png' src='align.png' \\>\n")
        table, rows = alignment_rows(alignment, total_length)
        with open(os.path.join(item_dir, "align.json"), "w") as f:
            json.dump(table, f)
    write_wav(os.path.join(item_dir, "audio.wav"), wav, sr)
    parts.append("<audio id='audio.wav' src='audio.wav' style='width: 100%;' controls></audio>\n")
    shown = [""] + list(lyrics)[:-1]                         # the model reads the lyrics shifted by one position
    spans = "".join(f"<span id='{item_id}/{k}'>{c}</span>" for k, c in enumerate(shown))
    parts.append(f"<pre style='white-space: pre-wrap;'><div>Artist {info['artist']}, Genre {info['genre']}</div>\n{spans}</pre>\n")
    with open(os.path.join(item_dir, "lyrics.json"), "w") as f:
        json.dump(shown, f)
    if rows is not None:
        parts.append(_SCRIPT.format(wav="audio.wav", track=item_id, rows=rows, src="align.json"))
    parts.append("</body></html>\n")
    with open(os.path.join(item_dir, "index.html"), "w") as f:
        f.write("".join(parts))


def save_html(logdir, x, zs, labels, alignments, hps):
    """save_html.py:7-25: `x` (n, T, 1) audio, `zs` the codes of all levels (the top level gives the track length in tokens),
    `labels` the top level's label dict, `alignments` per-item (total_length, n_lyric_tokens) arrays or None."""
    level = hps.levels - 1
    bs, total_length = int(zs[level].shape[0]), int(zs[level].shape[1])
    os.makedirs(logdir, exist_ok=True)
    frames = []
    for item in range(bs):
        wav = x[item].detach().cpu().numpy() if hasattr(x[item], "detach") else np.asarray(x[item])
        save_item(os.path.join(logdir, f"item_{item}"), item, wav, hps.sr, labels["info"][item], total_length,
                  alignments[item] if alignments is not None else None)
        frames.append(f"<iframe style='height: 100%; width: 100%;' frameborder='0' scrolling='no' src='item_{item}/index.html'></iframe>\n")
    with open(os.path.join(logdir, "index.html"), "w") as f:
        f.write(_HEAD.format(title=logdir) + "".join(frames) + "</body></html>\n")
