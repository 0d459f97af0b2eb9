"""Shared helpers for the parity tests: rebuild inputs/weights of a golden case from synth.py, run the oracle,
run the HIP engine."""
import json
import os

import numpy as np
import torch

import clip_fsar_amd.synth as synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# trained-CLIP-like activation statistics (two ln_pre channels at |x| ~ 100, row mean away from zero): the LayerNorm-folded GEMMs'
# cancellation regime.  The outlier channels swamp the tiny tower's class signal (top-1 at chance in the reference
# itself): these are numerics fixtures, not accuracy fixtures.
OUTLIER_CASES = ["t_outlier_5w1s_T8", "t197_outlier_5w1s_T2"]
SMALL_CASES = ["t_5w1s_T8", "t_5w5s_T8_mb", "t_5w5s_q2_T8", "t_5w3s_T16_mb_d2", "t_5w2s_T4_sd", "t197_5w1s_T2",
               "t257_5w1s_T2", "rn_t_5w2s_T4"]

# max |logits - reference| measured per golden case (MI355X, tools/parity_report.py -> profiles/r04_parity_table.md; bf16: the larger of
# the round-3 and round-4 tables, fp16: round 4's mode -- fp32 residual add, two-word stream, split QKV / out_proj / c_proj weights).  The
# 16-bit modes' regression bounds are 2 x these (a bound 10 x the measured value lets a 10 x regression through); fp32 and the
# full-size fp16 cases (cfg2 / cfg3 / cfg4) are held to the north-star 1e-3 in tests/test_gpu_e2e.py.
MEASURED_DLOGITS = {
    "t_5w1s_T8": {"bf16": 0.01404, "fp16": 0.00205},
    "t_5w5s_T8_mb": {"bf16": 0.0145, "fp16": 0.002639},
    "t_5w5s_q2_T8": {"bf16": 0.0135, "fp16": 0.002728},
    "t_5w3s_T16_mb_d2": {"bf16": 0.02286, "fp16": 0.003726},
    "t_5w2s_T4_sd": {"bf16": 0.004334, "fp16": 0.0009908},
    "t197_5w1s_T2": {"bf16": 0.008597, "fp16": 0.00097},
    "t257_5w1s_T2": {"bf16": 0.003436, "fp16": 0.0007384},
    "rn_t_5w2s_T4": {"bf16": 0.001986, "fp16": 0.00027},
    "t_outlier_5w1s_T8": {"bf16": 0.0007324, "fp16": 0.000144},
    "t197_outlier_5w1s_T2": {"bf16": 0.0002892, "fp16": 0.0001543},
    "cfg2_B16_5w1s_T8": {"bf16": 0.003892, "fp16": 0.0006766},
    "cfg3_B16_5w5s_T8_mb": {"bf16": 0.003086, "fp16": 0.0005069},
    "cfg4_L14_5w1s_T16": {"bf16": 0.00545, "fp16": 0.0004578},
    "rn50_5w1s_T2": {"bf16": 0.008433, "fp16": 0.000374},
}


def bound(name, precision):
    """Regression bound of a 16-bit mode on a golden case: 2 x the measured deviation (never below 1e-3)."""
    return max(2.0 * MEASURED_DLOGITS[name][precision], 1e-3)


def load_golden(name):
    z = np.load(os.path.join(GOLD, "head_%s.npz" % name))
    out = {k: z[k] for k in z.files}
    out["meta"] = json.loads(str(out["meta"]))
    return out


def case_inputs(meta, episode=None):
    """(arch dict, head state dict (torch cpu), text_train, text_test, episode dict (torch cpu))"""
    a = synth.ARCHS[meta["arch"]]
    depth = meta.get("depth", 1)
    sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(meta["arch"], seed=meta["seed"], depth=depth,
                                                                    outliers=meta.get("outliers")).items()}
    tt = torch.from_numpy(synth.text_features(meta["n_train"], a["embed"], "train", meta["seed"]))
    te = torch.from_numpy(synth.text_features(meta["n_test"], a["embed"], "test", meta["seed"]))
    ep = synth.make_episode(way=meta["way"], shot=meta["shot"], query_per_class=meta["q"], frames=meta["T"],
                            res=a["res"], n_test_classes=meta["n_test"],
                            episode=meta["episode"] if episode is None else episode, seed=meta["seed"],
                            lowfreq=meta.get("lowfreq", 0.0))
    ep = {k: torch.from_numpy(v) for k, v in ep.items()}
    return a, sd, tt, te, ep


_DEV_ENV = {"CFSAR_FP16_WIDE": "fp16_wide", "CFSAR_FP16_LO": "fp16_lo", "CFSAR_FP16_RAWMEANS": "fp16_rawmeans",
            "CFSAR_FUSED_UMEANS": "fused_umeans", "CFSAR_FUSED_OMEANS": "fused_omeans", "CFSAR_FUSE_STATS": "fuse_stats"}


def vit_options_from_env():
    """HipViT's developer options from the environment names the tools and ablation tests set (value "0" = off).  Test / tool
    infrastructure: the engine itself takes them as the `vit_options` constructor argument and reads none of these variables."""
    return {opt: os.environ[env] != "0" for env, opt in _DEV_ENV.items() if env in os.environ}


def run_engine(meta, a, sd, tt, te, eps, precision, taps=None, vit_options=None):
    """eps: list of episode dicts (cpu) -> (logits [B,Q,way], class_logits) on cpu."""
    from clip_fsar_amd.engine import ClipFsarEngine
    eng = ClipFsarEngine(a, sd, tt, te, depth=meta.get("depth", 1), precision=precision, device="cuda",
                         vit_options=vit_options_from_env() if vit_options is None else vit_options)
    dev = torch.device("cuda")
    sup = torch.stack([e["support_set"] for e in eps]).to(dev)
    tgt = torch.stack([e["target_set"] for e in eps]).to(dev)
    sl = torch.stack([e["support_labels"] for e in eps]).to(dev)
    rl = torch.stack([e["real_support_labels"] for e in eps]).to(dev)
    logits, cl = eng.forward(sup, tgt, sl, rl, way=meta["way"], T=meta["T"],
                             merge_before=meta.get("merge_before", False),
                             single_direct=meta.get("single_direct", False), taps=taps)
    torch.cuda.synchronize()
    return logits.cpu(), cl.cpu()


def maxdiff(a, b):
    a = torch.as_tensor(a).float()
    b = torch.as_tensor(b).float()
    return float((a - b).abs().max())


# ---- multi-episode reference goldens (round 5; oracle/make_golden.py --multi): name -> file tests/golden/multi_<name>.npz
MULTI_CASES = ("mc_cfg2_B16_5w1s_T8", "hc_cfg2_B16_5w1s_T8", "hc_cfg3_B16_5w5s_T8_mb", "hc_cfg4_L14_5w1s_T16", "mc_cfg4_L14_5w1s_T16",
               "hc_rn50_5w1s_T8", "oc_cfg2_B16_5w1s_T8")


_MULTI_CACHE = {}


def load_multi(name):
    z = np.load(os.path.join(GOLD, "multi_%s.npz" % name))
    return {"meta": json.loads(str(z["meta"])), "logits": z["logits"], "class_logits": z["class_logits"]}


def multi_case_stats(name, precision, chunk=None):
    """Run every episode of a multi-episode golden through the engine in `precision`; deviation statistics of the logits against the
    REFERENCE's (the generating script imported /root/reference)."""
    g = load_multi(name)
    m = g["meta"]
    a = synth.ARCHS[m["arch"]]
    E = m["episodes"]
    # weights and episodes of the LAST case are kept: the three modes of a case (and the two contrast sets of an architecture) share them, and
    # generating the ViT-L/14 state dict takes half a minute of numpy
    wkey, ekey = (m["arch"], m["seed"], m["n_train"], m["n_test"], json.dumps(m.get("outliers"), sort_keys=True)), (name,)
    if _MULTI_CACHE.get("wkey") != wkey:
        _MULTI_CACHE.clear()
        _MULTI_CACHE.update(wkey=wkey, w=({k: torch.from_numpy(v) for k, v in synth.head_state_dict(m["arch"], seed=m["seed"], outliers=m.get("outliers")).items()},
                                          torch.from_numpy(synth.text_features(m["n_train"], a["embed"], "train", m["seed"])),
                                          torch.from_numpy(synth.text_features(m["n_test"], a["embed"], "test", m["seed"]))))
    if _MULTI_CACHE.get("ekey") != ekey:
        _MULTI_CACHE.update(ekey=ekey, eps=[{k: torch.from_numpy(v) for k, v in synth.make_episode(
            way=m["way"], shot=m["shot"], query_per_class=m["q"], frames=m["T"], res=a["res"], n_test_classes=m["n_test"], episode=e, seed=m["seed"],
            lowfreq=m.get("lowfreq", 0.0)).items()} for e in range(E)])
    sd, tt, te = _MULTI_CACHE["w"]
    eps = _MULTI_CACHE["eps"]
    chunk = chunk or (4 if m["arch"] == "ViT-L/14" else 8)
    lg = torch.cat([run_engine(m, a, sd, tt, te, eps[i:i + chunk], precision)[0] for i in range(0, E, chunk)])
    ref = torch.from_numpy(g["logits"])
    d = (lg - ref).abs()
    per_ep = d.reshape(E, -1).max(1).values
    spread = (ref.reshape(E, -1).max(1).values - ref.reshape(E, -1).min(1).values)
    flat = d.flatten().sort().values
    return {"episodes": E, "rows": int(ref.shape[0] * ref.shape[1]), "mean_spread": float(spread.mean()),
            "rms": float(d.pow(2).mean().sqrt()), "p99": float(flat[min(len(flat) - 1, int(0.99 * len(flat)))]), "max": float(d.max()),
            "worst_episode_max": float(per_ep.max()), "episodes_over_1e-3": int((per_ep > 1e-3).sum()),
            "max_rel_spread": float((per_ep / spread).max()), "rms_rel_spread": float((d.reshape(E, -1).pow(2).mean(1).sqrt() / spread).mean()),
            "argmax_equal": int((lg.argmax(-1) == ref.argmax(-1)).sum())}
