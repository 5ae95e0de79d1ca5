"""CPU port of the decode step on torch's CPU kernels (fp32) -- test infrastructure / bench.py's cpu_baseline only.

Same algorithm as oracle.transformer.Transformer.forward at q_l == 1 (Transformer.forward(sample=True),
jukebox/transformer/transformer.py:169-192; ResAttnBlock :62-66,82-86; FactoredAttention :289-301 with the closed-form
key sets of SURVEY.md Appendix B), but on the BLAS / vector kernels the reference itself runs on when it is on a CPU
(torch.addmm, F.layer_norm, softmax), with preallocated k/v caches instead of the reference's per-step t.cat.  It is
therefore a slightly FASTER stand-in for the unmodified reference (measured in the build container on 8 cores:
reference 185 ms, this port ~ the same BLAS time without the Python / cat overhead, numpy oracle 470 ms per decode step
of the level-0 upsampler at batch 16; tests/golden/time_reference_cpu.py)."""
import numpy as np
import torch
import torch.nn.functional as F

from .transformer import attn_func_of_layer, decode_key_index, rounded_prime_len


class TorchDecodeStack:
    def __init__(self, sd, prefix, n_in, n_ctx, n_head, n_depth, attn_order=0, blocks=None, m_attn=0.25, prime_len=None,
                 n_batch=1, encoder_kv=None, device="cpu"):
        """device: where torch runs the port.  "cpu" is the CPU baseline's form; the GPU suite's full-size cases pass "cuda" --
        torch's own fp32 kernels (rocBLAS / ATen), still an implementation independent of libjukebox_hip -- because 64 steps
        of a 72-layer model at batch 16 cost minutes on the lease's host cores."""
        self.device = torch.device(device)
        g = lambda name: torch.as_tensor(np.asarray(sd[prefix + name], dtype=np.float32)).to(self.device)
        self.n_in, self.n_ctx, self.H, self.L = n_in, n_ctx, n_head, n_depth
        self.S = int(m_attn * n_in)
        self.bc = n_ctx // blocks if blocks else None
        self.prime_r = rounded_prime_len(prime_len, blocks) if prime_len else None
        self.funcs = [attn_func_of_layer(attn_order, d) for d in range(n_depth)]
        self.layers = []
        for d in range(n_depth):
            p = f"_attn_mods.{d}."
            self.layers.append({k: g(p + v) for k, v in dict(
                wa="attn.c_attn.w", ba="attn.c_attn.b", wp="attn.c_proj.w", bp="attn.c_proj.b", g0="ln_0.weight",
                b0="ln_0.bias", wf="mlp.c_fc.w", bf="mlp.c_fc.b", w2="mlp.c_proj.w", b2="mlp.c_proj.b", g1="ln_1.weight",
                b1="ln_1.bias").items()})
        cap = lambda d: self.prime_r if self.funcs[d] == 7 else (0 if self.funcs[d] == 6 else n_ctx)
        self.K = [torch.zeros(n_batch, cap(d), self.S, device=self.device) for d in range(n_depth)]
        self.V = [torch.zeros(n_batch, cap(d), self.S, device=self.device) for d in range(n_depth)]
        # cross-attention layers (attn_func 6): key / value = c_enc_kv(encoder_kv), once (decode_qkv, factored_attention.py:273-280)
        if 6 in self.funcs:
            ekv = torch.as_tensor(np.asarray(encoder_kv, dtype=np.float32)).to(self.device)
            for d in range(n_depth):
                if self.funcs[d] == 6:
                    kv = torch.matmul(ekv, g(f"_attn_mods.{d}.attn.c_enc_kv.w")) + g(f"_attn_mods.{d}.attn.c_enc_kv.b")
                    self.K[d], self.V[d] = kv[..., :self.S].contiguous(), kv[..., self.S:].contiguous()
        self.t = 0

    @torch.no_grad()
    def forward(self, x):
        """x: (N, 1, n_in) float32 at position self.t; returns (N, 1, n_in)."""
        x = torch.as_tensor(x, dtype=torch.float32).to(self.device).reshape(-1, self.n_in)
        N, S, H, t = x.shape[0], self.S, self.H, self.t
        d_head = S // H
        scale2 = 1.0 / np.sqrt(d_head)                       # (d^-1/4)^2, factored_attention.py:84-92
        for d, p in enumerate(self.layers):
            func = self.funcs[d]
            h = F.layer_norm(x, (self.n_in,), p["g0"], p["b0"], 1e-5)
            qkv = torch.addmm(p["ba"], h, p["wa"])
            if func == 6:                                # the query only; every encoder position is a key (decode_attn :226-228)
                q, idx = qkv, np.arange(self.K[d].shape[1])
            else:
                q, k, v = qkv[:, :S], qkv[:, S:2 * S], qkv[:, 2 * S:]
                if t < self.K[d].shape[1]:
                    self.K[d][:, t], self.V[d][:, t] = k, v
                idx = decode_key_index(func, t, self.bc, self.prime_r)
            if idx is None:
                a = torch.zeros(N, S, device=self.device)
            else:
                sl = slice(int(idx[0]), int(idx[-1]) + 1, int(idx[1] - idx[0]) if len(idx) > 1 else 1)
                Ks = self.K[d][:, sl].reshape(N, -1, H, d_head).transpose(1, 2)      # (N, H, kl, d)
                Vs = self.V[d][:, sl].reshape(N, -1, H, d_head).transpose(1, 2)
                w = torch.matmul(q.reshape(N, H, 1, d_head), Ks.transpose(-1, -2)) * scale2
                a = torch.matmul(torch.softmax(w, dim=-1), Vs).reshape(N, S)
            xb = x + torch.addmm(p["bp"], a, p["wp"])
            h1 = F.layer_norm(xb, (self.n_in,), p["g1"], p["b1"], 1e-5)
            m = torch.addmm(p["bf"], h1, p["wf"])
            m = m * torch.sigmoid(1.702 * m)
            x = xb + torch.addmm(p["b2"], m, p["w2"])
        self.t = t + 1
        return x.reshape(N, 1, self.n_in)
