#!/bin/bash
# GPU box: everything the bench line's roofline object cites, for the bench command itself.
#   1. rocprofv3 --kernel-trace --stats                      -> kernel_summary.txt / kernel_stats.csv
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, SEPARATE passes (TCC slots) -> gemm_traffic.json (bytes per launch per kernel)
#   3. --pmc matrix-pipe / LDS groups                        -> pmc_by_kernel.json (MFMA busy fraction, LDS conflict rate)
# Summaries land in gpurun_out/prof_<tag>/ and are copied to profiles/ by hand.   usage: tools/collect_profiles.sh <tag>
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
EPS=${EPS:-36}
CMD="python bench.py --steps 4 --warmup 2 --episodes-per-step $EPS --no-cpu-baseline --no-kernel-events --no-fp16-leg --no-config-legs ${CMD_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python tools/trace_summary.py $OUT/trace/t_kernel_trace.csv 0 > $OUT/kernel_summary.txt
cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o p -- $CMD > $OUT/mfma.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -o p -- $CMD > $OUT/tcc.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $OUT/lds -o p -- $CMD > $OUT/lds.log 2>&1
python - <<PY
import csv, json, collections, re, os
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:96]
def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path): return d
    for r in csv.DictReader(open(path)):
        d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d
f = agg("$OUT/fetch/p_counter_collection.csv")
w = agg("$OUT/write/p_counter_collection.csv")
traffic = {}
tot = n = 0
for k in f:
    if "gemm" not in k and "patch_embed" not in k: continue
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM)
    fv, wv = f[k]["FETCH_SIZE"], w.get(k, {}).get("WRITE_SIZE", [0.0])
    fb, wb = sum(fv) * 1024 * 2, sum(wv) * 1024
    traffic[k] = {"launches": len(fv), "fetch_bytes_per_launch": fb / len(fv), "write_bytes_per_launch": wb / max(len(wv), 1)}
    if "vit_gemm" in k or "gemm_kernel_p" in k or "patch_embed_kernel" in k:
        tot += fb + wb; n += len(fv)
import hashlib, socket
h = hashlib.sha256()
for nm in ("common.h", "gemm.hip", "gemm_vit.h", "gemm_vit_epi.h", "gemm_vit.hip"):          # bench.py::csrc_fingerprint
    h.update(open(os.path.join("clip-fsar_amd", "csrc", nm), "rb").read())
traffic["_all_bf16_gemm"] = {"launches": n, "hbm_bytes_per_launch": tot / max(n, 1),
                             "csrc_sha16": h.hexdigest()[:16], "commit": os.environ.get("COMMIT", "unknown (no .git on the GPU box)"),
                             "box": socket.gethostname(),
                             "note": "FETCH_SIZE*1024*2 (gfx950 correction) + WRITE_SIZE*1024, separate --pmc passes, bench.py --episodes-per-step $EPS"}
json.dump(traffic, open("$OUT/gemm_traffic.json", "w"), indent=1)
m, l = agg("$OUT/mfma/p_counter_collection.csv"), agg("$OUT/lds/p_counter_collection.csv")
t = agg("$OUT/tcc/p_counter_collection.csv")
pmc = {}
for k in m:
    c = {x: sum(v) / len(v) for x, v in m[k].items()}
    c.update({x: sum(v) / len(v) for x, v in l.get(k, {}).items()})
    c.update({x: sum(v) / len(v) for x, v in t.get(k, {}).items()})
    gui = c.get("GRBM_GUI_ACTIVE", 0)
    e = {"launches": len(next(iter(m[k].values()))), "counters_per_launch": c}
    if gui > 0:
        # GRBM_GUI_ACTIVE sums the 8 XCDs; a CU has 4 SIMDs, each with one matrix pipe: busy fraction = MFMA busy cycles
        # / (cycles x 256 CUs x 4 SIMDs)
        e["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 256 * 4)
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        e["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
    pmc[k] = e
json.dump(pmc, open("$OUT/pmc_by_kernel.json", "w"), indent=1)
print(json.dumps(traffic["_all_bf16_gemm"]))
for k, e in sorted(pmc.items(), key=lambda kv: -kv[1]["counters_per_launch"].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"])[:10]:
    print("%-90s mfma_busy %.3f  l2_hit %s  lds_conflict %s" % (k[:90], e.get("mfma_busy_frac", -1), e.get("l2_hit_rate"), e.get("lds_bank_conflict_frac")))
PY
cat $OUT/kernel_summary.txt
