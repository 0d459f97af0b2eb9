"""Dev tool (CPU): gpurun_out/parity_multi.json (tools/parity_multi.py) -> the table of profiles/rNN_parity_table.md on stdout.
usage: python tools/parity_multi_table.py [json]"""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_multi.json"))
names = {"mc_cfg2_B16_5w1s_T8": "cfg2 (ViT-B/16, 5-way 1-shot, 8 frames), standard contrast", "hc_cfg2_B16_5w1s_T8": "cfg2, high contrast",
         "hc_cfg3_B16_5w5s_T8_mb": "cfg3 (5-way 5-shot, MERGE_BEFORE), high contrast", "hc_cfg4_L14_5w1s_T16": "cfg4 (ViT-L/14, 16 frames), high contrast",
         "mc_cfg4_L14_5w1s_T16": "cfg4, standard contrast", "hc_rn50_5w1s_T8": "CLIP RN50 tower (N3), 5-way 1-shot, 8 frames, lowfreq 2.0",
         "oc_cfg2_B16_5w1s_T8": "cfg2 with trained-CLIP-like outlier channels (|x| ~ 100 in two ln_pre channels), high contrast"}
print("| configuration (13 reference episodes = 65 logit rows each) | mean logits spread | mode | rms | p99 | max | episodes whose largest deviation > 1e-3 | max / spread | argmax equal |")
print("|---|---|---|---|---|---|---|---|---|")
for n in names:
    if n not in d:
        continue
    for mode in ("fp32", "fp16", "bf16"):
        s = d[n].get(mode)
        if not s:
            continue
        print("| %s | %.2f | %s | %.2e | %.2e | %.2e | %d of %d | %.1e | %d / %d |" % (
            names[n], s["mean_spread"], mode, s["rms"], s["p99"], s["max"], s["episodes_over_1e-3"], s["episodes"], s["max_rel_spread"], s["argmax_equal"], s["rows"]))
