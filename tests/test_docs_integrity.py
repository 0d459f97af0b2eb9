"""The documents cite files (profiles, tools, tests, sources) as evidence: every plain path they name must exist in the tree."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["README.md", "DESIGN.md", "INTEGRATION.md", "docs/history/r05.md", "docs/history/r06.md", "profiles/r05_gemm_forms.md", "profiles/r05_parity_table.md",
        "profiles/r05_attention.md", "profiles/r06_strict_eval.md", "profiles/r06_strict_budget.md", "tools/README.md"]
PREFIXES = ("profiles/", "tools/", "tests/", "docs/", "oracle/", "include/", "clip-fsar_amd/", "csrc/")


def test_cited_files_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for tok in re.findall(r"`([^`\s]+)`", text):
            path = tok.split("::")[0]
            if not path.startswith(PREFIXES) or any(c in path for c in "*{}<>[]|$") or path.endswith(("/", "_")):
                continue
            if path.startswith("csrc/"):
                path = "clip-fsar_amd/" + path
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, tok))
    assert not missing, missing


def test_quoted_headline_numbers_are_the_committed_bench_line():
    """the headline figures README.md and DESIGN.md quote are the ones of the committed bench line (profiles/r06_bench.json: the default
    `python bench.py` run of the round's evidence session) -- episodes/s of the bf16 headline, of the fp16 and fp16_strict legs (one decimal) and the GEMM
    roofline fraction (three decimals)."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    want = {"bf16 episodes/s": "%.1f" % d["value"], "fp16 leg": "%.1f" % d["fp16_mode"]["value"], "fp16_strict leg": "%.1f" % d["strict_mode"]["value"],
            "GEMM roofline frac": "%.3f" % d["roofline"]["frac"]}
    for doc in ("README.md", "DESIGN.md"):
        text = open(os.path.join(ROOT, doc)).read()
        missing = {k: v for k, v in want.items() if v not in text}
        assert not missing, (doc, missing)
    assert d["config"]["episodes_per_step_per_gpu"] == 36 and d["strict_mode"]["parity"]["meets_north_star"]

