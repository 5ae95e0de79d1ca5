"""A miniature of the benchmark job that touches every kernel of the path with realistic shapes -- for ONE
`rocprofv3 --kernel-trace --stats` summary (the full job is 3e7 launches: no trace survives that).  1b_lyrics, 16 samples:
  level 2: lyric prefill (384 tokens) + 96 decode steps;   level 1 / level 0: conditioner, prefill of 4096 primed tokens
  (2 chunks of 2048 positions), 128 decode steps each;   VQ-VAE decode of the three levels' codes (1 s of audio).
Usage: rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python tools/profile_job.py"""
import sys
import torch
sys.path.insert(0, ".")
import bench

dev = torch.device("cuda:0")
sr, N = 44100, 16
sample_length = 6 * sr // 128 * 128
vq, priors = bench.build_models("1b_lyrics", sample_length, dev)
labels = bench.synthetic_labels(priors, N, 180 * sr, dev)
torch.manual_seed(1)
zs = [torch.randint(0, 2048, (N, sample_length // p.raw_to_tokens), device=dev) for p in priors]
kw = dict(fp16=True, temp=0.99, seed=0)
top, up1, up0 = priors[2], priors[1], priors[0]
y = lambda p: p.get_y(labels[p.level], 0)
z2 = top.sample(N, z=torch.zeros(N, 0, dtype=torch.long, device=dev), y=y(top), sample_tokens=96, **kw)
for p in (up1, up0):
    zc = zs[p.level + 1][:, :p.n_ctx // 4].contiguous()                 # the upper level's codes of one window
    zp = zs[p.level][:, :4096].contiguous()
    z = p.sample(N, z=zp, z_conds=[zc], y=y(p), sample_tokens=4096 + 128, **kw)
x = vq.decode([zs[0][:, :5504], zs[1][:, :1376], zs[2][:, :344]], start_level=0, bs_chunks=N)
torch.cuda.synchronize()
print("done", tuple(z2.shape), tuple(z.shape), tuple(x.shape))
