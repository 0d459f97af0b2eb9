"""The documents cite files (profiles, tools, tests, sources) as evidence: every plain path they name must exist in the tree."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["README.md", "DESIGN.md", "INTEGRATION.md", "docs/history/r05.md", "profiles/r05_gemm_forms.md", "profiles/r05_parity_table.md",
        "profiles/r05_attention.md"]
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
