"""N1 -- CLIP text side at init time: byte-level BPE tokenizer and the text transformer on the HIP kernels.

The reference builds its class-name tables once in ``CNN_OTAM_CLIPFSAR.__init__`` (few_shot.py:2714-2728):
``tokenize(["a photo of {c}" ...])`` (:393-429, SimpleTokenizer :110-180) then ``CLIP.encode_text`` (:793-806).
Nothing here runs per episode.

Tokenizer: the published OpenAI-CLIP byte-level BPE (lower-cased, whitespace-collapsed text; merges ranked by their
position in ``bpe_simple_vocab_16e6.txt.gz``; ids = 256 byte symbols, 256 end-of-word byte symbols, one per merge,
then <|startoftext|>, <|endoftext|>).  The merges file is a third-party data file that is NOT vendored in this
repository: pass ``bpe_path`` or set ``CLIP_BPE_PATH`` (the reference keeps it at models/base/bpe_simple_vocab_16e6.txt.gz).
"""
from __future__ import annotations

import gzip
import html
import os
from collections import OrderedDict

import numpy as np
import torch

from . import synth

CONTEXT_LENGTH = 77
N_MERGES = 49152 - 256 - 2


def _byte_symbols():
    """Printable stand-ins for the 256 byte values (GPT-2 convention): printable latin-1 bytes map to themselves, the
    remaining 68 to code points 256, 257, ..."""
    keep = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class ClipBpeTokenizer:
    def __init__(self, bpe_path: str | None = None):
        import regex  # \\p{L} / \\p{N} classes
        bpe_path = bpe_path or os.environ.get("CLIP_BPE_PATH")
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError("CLIP BPE merges file not found (pass bpe_path or set CLIP_BPE_PATH to "
                                    "bpe_simple_vocab_16e6.txt.gz)")
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(l.split()) for l in lines[1:N_MERGES + 1]]
        self.byte_sym = _byte_symbols()
        base = [self.byte_sym[b] for b in sorted(self.byte_sym, key=lambda b: (b not in _PRINTABLE_ORDER, _ORDER_KEY(b)))]
        vocab = base + [s + "</w>" for s in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.ids = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.ids["<|startoftext|>"], self.ids["<|endoftext|>"]
        self._memo = {}
        self._split = regex.compile(
            r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)

    def _merge_word(self, symbols):
        """Greedy lowest-rank-first pair merging of one word (last symbol carries the '</w>' marker)."""
        word = list(symbols)
        while len(word) > 1:
            best, best_rank = None, None
            for a, b in zip(word, word[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                    merged.append(best[0] + best[1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        return word

    def encode(self, text: str):
        text = html.unescape(html.unescape(text)).strip()          # ftfy.fix_text is the identity on clean ASCII names
        text = " ".join(text.split()).lower()
        out = []
        for piece in self._split.findall(text):
            key = "".join(self.byte_sym[b] for b in piece.encode("utf-8"))
            toks = self._memo.get(key)
            if toks is None:
                if key in ("<|startoftext|>", "<|endoftext|>"):
                    toks = [key]
                else:
                    toks = self._merge_word(list(key[:-1]) + [key[-1] + "</w>"])
                self._memo[key] = toks
            out.extend(self.ids[t] for t in toks)
        return out

    def tokenize(self, texts, context_length: int = CONTEXT_LENGTH, truncate: bool = False) -> np.ndarray:
        """[n, context_length] int32: <sot> ids <eot> then zero padding (few_shot.py:393-429)."""
        if isinstance(texts, str):
            texts = [texts]
        res = np.zeros((len(texts), context_length), np.int32)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError("Input %s is too long for context length %d" % (t, context_length))
                ids = ids[:context_length]
                ids[-1] = self.eot
            res[i, :len(ids)] = ids
        return res


# byte-symbol ordering of the published vocabulary: the 188 printable bytes first (in byte order), then the 68 others
_PRINTABLE_ORDER = set(list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256)))
_ORDER_KEY = lambda b: b  # noqa: E731


def prompts(class_names, template: str | None = None):
    """``TEST.PROMPT.format(c)`` if given else "a photo of {c}" (few_shot.py:2715-2718)."""
    tpl = template or "a photo of {}"
    return [tpl.format(c) for c in class_names]


# ------------------------------------------------------------------------------------------------- text tower weights
def text_tower_state_dict(width: int = 512, layers: int = 12, embed: int = 512, vocab: int = 49408,
                          context: int = CONTEXT_LENGTH, seed: int = 18):
    """Random-init CLIP text tower with the reference's parameter names (CLIP.__init__ few_shot.py:731-747) at the
    scales of ``initialize_parameters`` (:749-776) -- deterministic (synth.tensor), for tests and synthetic runs."""
    sd = OrderedDict()

    def put(key, shape, std, mean=0.0):
        sd[key] = synth.tensor(shape, "texttower/%d/%d/%s" % (width, layers, key), std=std, mean=mean, seed=seed)

    put("token_embedding.weight", (vocab, width), 0.02)
    put("positional_embedding", (context, width), 0.01)
    attn_std, proj_std, fc_std = width ** -0.5, (width ** -0.5) * ((2 * layers) ** -0.5), (2 * width) ** -0.5
    for i in range(layers):
        b = "transformer.resblocks.%d." % i
        put(b + "ln_1.weight", (width,), 0.05, 1.0)
        put(b + "ln_1.bias", (width,), 0.05)
        put(b + "attn.in_proj_weight", (3 * width, width), attn_std)
        put(b + "attn.in_proj_bias", (3 * width,), 0.05)
        put(b + "attn.out_proj.weight", (width, width), proj_std)
        put(b + "attn.out_proj.bias", (width,), 0.05)
        put(b + "ln_2.weight", (width,), 0.05, 1.0)
        put(b + "ln_2.bias", (width,), 0.05)
        put(b + "mlp.c_fc.weight", (4 * width, width), fc_std)
        put(b + "mlp.c_fc.bias", (4 * width,), 0.05)
        put(b + "mlp.c_proj.weight", (width, 4 * width), proj_std)
        put(b + "mlp.c_proj.bias", (width,), 0.05)
    put("ln_final.weight", (width,), 0.05, 1.0)
    put("ln_final.bias", (width,), 0.05)
    put("text_projection", (width, embed), width ** -0.5)
    return sd


class HipTextEncoder:
    """CLIP.encode_text (few_shot.py:793-806) in fp32 on the HIP kernels: token + positional embedding, causal
    pre-LN transformer (ResidualAttentionBlock :619-640 with the additive -inf mask :778-784), ln_final, EOT-token
    pooling (position of the largest token id), ``@ text_projection``."""

    def __init__(self, sd: dict, device="cuda"):
        from . import hip
        self.hip = hip
        self.dev = torch.device(device)

        def g(name):
            t = sd[name]
            if not isinstance(t, torch.Tensor):
                t = torch.from_numpy(np.asarray(t))
            return t.detach().to(device=self.dev, dtype=torch.float32).contiguous()

        self.table, self.pos = g("token_embedding.weight"), g("positional_embedding")
        self.W = self.table.shape[1]
        if self.W % 64:
            raise ValueError("text width must be a multiple of 64 (heads = width // 64, few_shot.py:867)")
        self.heads = self.W // 64
        self.ln_final = (g("ln_final.weight"), g("ln_final.bias"))
        self.proj_t = g("text_projection").t().contiguous()
        n_layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
        self.blocks = []
        for i in range(n_layers):
            b = "transformer.resblocks.%d." % i
            self.blocks.append(dict(ln1=(g(b + "ln_1.weight"), g(b + "ln_1.bias")), ln2=(g(b + "ln_2.weight"), g(b + "ln_2.bias")),
                                    w_qkv=g(b + "attn.in_proj_weight"), b_qkv=g(b + "attn.in_proj_bias"),
                                    w_out=g(b + "attn.out_proj.weight"), b_out=g(b + "attn.out_proj.bias"),
                                    w_fc=g(b + "mlp.c_fc.weight"), b_fc=g(b + "mlp.c_fc.bias"),
                                    w_pr=g(b + "mlp.c_proj.weight"), b_pr=g(b + "mlp.c_proj.bias")))

    def encode(self, tokens) -> torch.Tensor:
        """tokens [n, L] int (numpy or tensor) -> [n, embed] fp32 on the device."""
        hip = self.hip
        tok = torch.as_tensor(np.asarray(tokens)).to(torch.int32)
        n, L = tok.shape
        W, dev = self.W, self.dev
        rows = n * L
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        x, h, qkv, o, u = f(rows, W), f(rows, W), f(rows, 3 * W), f(rows, W), f(rows, 4 * W)
        hip.embed_tokens(tok.to(dev).contiguous(), self.table, self.pos[:L].contiguous(), x)
        for b in self.blocks:
            hip.layernorm(x, h, b["ln1"][0], b["ln1"][1], rows, W)
            hip.gemm(h, b["w_qkv"], qkv, bias=b["b_qkv"])
            hip.seq_attention(qkv, o, n, L, 0, 0, self.heads, 64, 0.125, causal=True)
            hip.gemm(o, b["w_out"], x, bias=b["b_out"], residual=x)
            hip.layernorm(x, h, b["ln2"][0], b["ln2"][1], rows, W)
            hip.gemm(h, b["w_fc"], u, bias=b["b_fc"], act=hip.ACT_QUICKGELU)
            hip.gemm(u, b["w_pr"], x, bias=b["b_pr"], residual=x)
        eot = (torch.arange(n) * L + tok.long().argmax(dim=1)).to(torch.int32).to(dev)      # text.argmax(-1) (:804)
        pooled, normed = f(n, W), f(n, W)
        hip.gather_rows(x, eot, pooled)
        hip.layernorm(pooled, normed, self.ln_final[0], self.ln_final[1], n, W)             # LN is per row: gather first
        out = f(n, self.proj_t.shape[0])
        hip.gemm(normed, self.proj_t, out)
        return out


def encode_class_names(class_names, text_sd, template=None, bpe_path=None, device="cuda") -> torch.Tensor:
    """The reference's init-time table: encode_text(tokenize(prompts)) -> [n_classes, embed] fp32 (few_shot.py:2714-2728)."""
    tok = ClipBpeTokenizer(bpe_path).tokenize(prompts(class_names, template))
    return HipTextEncoder(text_sd, device=device).encode(tok)
