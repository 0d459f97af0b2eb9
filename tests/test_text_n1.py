"""N1: CLIP text side.  CPU: tokenizer ids == the reference's tokenize() (fixture generated from the real
reference; needs the third-party BPE merges file, so it is skipped where that file is absent) and the oracle's
encode_text == the reference's CLIP.encode_text.  GPU: the HIP text encoder == the reference outputs."""
import json
import os

import numpy as np
import pytest
import torch

import clip_fsar_amd.text as ctext
import clipfsar_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden", "text_cases.npz")
# the third-party merges file is never committed; dev runs keep a git-ignored copy under tests/_local/
_CANDIDATES = [os.environ.get("CLIP_BPE_PATH", ""), "/root/reference/models/base/bpe_simple_vocab_16e6.txt.gz",
               os.path.join(os.path.dirname(__file__), "_local", "bpe_simple_vocab_16e6.txt.gz")]
BPE = next((p for p in _CANDIDATES if p and os.path.exists(p)), _CANDIDATES[1])


def _gold():
    z = np.load(GOLD)
    return {k: z[k] for k in z.files}


@pytest.mark.skipif(not os.path.exists(BPE), reason="CLIP BPE merges file not available")
def test_tokenizer_matches_reference_ids():
    g = _gold()
    texts = json.loads(str(g["texts"]))
    tk = ctext.ClipBpeTokenizer(BPE)
    assert np.array_equal(tk.tokenize(texts), g["tokens"])
    # SURVEY.md section 4 known answers
    assert tk.tokenize("a photo of blasting sand")[0, :8].tolist() == [49406, 320, 1125, 539, 26178, 5094, 49407, 0]
    assert tk.tokenize("a photo of hurling (sport)")[0, :10].tolist() == [49406, 320, 1125, 539, 24738, 263, 2364, 264, 49407, 0]
    with pytest.raises(RuntimeError):
        tk.tokenize("word " * 100)
    assert tk.tokenize("word " * 100, truncate=True)[0, -1] == tk.eot
    assert ctext.prompts(["x"], "a video of {}") == ["a video of x"]


def test_tokenizer_requires_the_merges_file():
    with pytest.raises(FileNotFoundError):
        ctext.ClipBpeTokenizer("/nonexistent/bpe.txt.gz")


@pytest.mark.parametrize("tag,width,layers,embed", [("small", 128, 2, 64)])
def test_oracle_encode_text_matches_reference(tag, width, layers, embed):
    g = _gold()
    seed = json.loads(str(g["meta"]))["seed"]
    sd = {k: torch.from_numpy(v) for k, v in ctext.text_tower_state_dict(width, layers, embed, seed=seed).items()}
    with torch.no_grad():
        out = orc.encode_text(torch.from_numpy(g["tokens"]), sd)
    assert float((out - torch.from_numpy(g["feats_" + tag])).abs().max()) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag,width,layers,embed", [("small", 128, 2, 64), ("b16", 512, 12, 512)])
def test_hip_text_encoder_matches_reference(tag, width, layers, embed):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = _gold()
    seed = json.loads(str(g["meta"]))["seed"]
    sd = ctext.text_tower_state_dict(width, layers, embed, seed=seed)
    enc = ctext.HipTextEncoder(sd)
    out = enc.encode(g["tokens"]).cpu()
    torch.cuda.synchronize()
    ref = torch.from_numpy(g["feats_" + tag])
    assert float((out - ref).abs().max()) < 1e-3, float((out - ref).abs().max())


@pytest.mark.gpu
def test_causal_seq_attention_kernel():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from clip_fsar_amd import hip
    heads, hd, n, L = 2, 64, 3, 77
    inner = heads * hd
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(n * L, 3 * inner, generator=g)
    out = torch.empty(n * L, inner, device="cuda")
    hip.seq_attention(qkv.cuda(), out, n, L, 0, 0, heads, hd, hd ** -0.5, causal=True)
    mask = torch.full((L, L), float("-inf")).triu_(1)
    for s in range(n):
        q, k, v = [t.reshape(L, heads, hd).transpose(0, 1) for t in qkv[s * L:(s + 1) * L].split(inner, dim=1)]
        a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5 + mask, dim=-1)
        ref = (a @ v).transpose(0, 1).reshape(L, inner)
        assert float((out.cpu()[s * L:(s + 1) * L] - ref).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BPE), reason="CLIP BPE merges file not available")
def test_head_builds_text_tables_from_class_names():
    """The head's init-time path (few_shot.py:2714-2728): class names -> prompts -> tokenizer -> HIP text encoder."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from types import SimpleNamespace as NS
    import clip_fsar_amd.models.base  # noqa: F401
    from clip_fsar_amd.models.base.builder import build_model
    import clip_fsar_amd.synth as synth
    names_tr, names_te = ["air drumming", "bowling", "cheerleading", "zumba", "yoga"], ["busking", "unboxing", "ice skating", "side kick", "tap dancing"]
    cfg = NS(VIDEO=NS(HEAD=NS(NAME="CNN_OTAM_CLIPFSAR", BACKBONE_NAME="ViT-B/16", PRECISION="bf16", TEXT_TOWER="synthetic",
                              BPE_PATH=BPE), BACKBONE=NS(META_ARCH="Identity")),
             TRAIN=NS(CLASS_NAME=names_tr, WAY=5), TEST=NS(CLASS_NAME=names_te), DATA=NS(NUM_INPUT_FRAMES=2),
             MODEL=NS(NAME="BaseVideoModel", EMA=NS(ENABLE=False)), BN=NS(FREEZE=False), NUM_GPUS=1, NUM_SHARDS=1,
             RANDOM_SEED=18)
    model, _ = build_model(cfg)
    model.eval()
    ep = synth.make_episode(way=5, shot=1, query_per_class=1, frames=2, res=224, n_test_classes=5, episode=0)
    task = {k: torch.from_numpy(v).cuda() for k, v in ep.items()}
    with torch.no_grad():
        out = model(task)
    assert out["logits"].shape == (5, 5) and torch.isfinite(out["logits"]).all()
    head = model.head
    assert head.text_features_test.shape == (5, 512) and head.text_features_train.shape == (5, 512)
    # the table equals the oracle's encode_text on the same tokens / weights
    tsd = {k: torch.from_numpy(v) for k, v in ctext.text_tower_state_dict(512, 12, 512, seed=18).items()}
    tok = ctext.ClipBpeTokenizer(BPE).tokenize(ctext.prompts(names_te))
    with torch.no_grad():
        ref = orc.encode_text(torch.from_numpy(tok), tsd)
    assert float((head.text_features_test - ref).abs().max()) < 1e-3
