"""Dev tool (GPU): feature error of the RN50 tower's 16-bit modes against its fp32 mode by number of frames (= by which kernels serve the
launches: small launches take the explicit gather + generic GEMM, batch-scale ones the implicit / direct convs and the 256 x 256 kernels),
with the CPU model's prediction (tools/numerics_lab_rn.py) for the same frames."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
synth = importlib.import_module("clip-fsar_amd.synth")
engine = importlib.import_module("clip-fsar_amd.engine")

torch.set_grad_enabled(False)
arch = sys.argv[1] if len(sys.argv) > 1 else "RN50"
a = synth.ARCHS[arch]
sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(arch, 18).items()}
eps = [synth.make_episode(5, 1, 1, 8, a["res"], 24, 200 + e, 18) for e in range(8)]
frames = torch.cat([torch.cat([torch.from_numpy(e["support_set"]), torch.from_numpy(e["target_set"])]) for e in eps]).cuda()
print("frames", tuple(frames.shape))
towers = {p: engine.HipResNet(a, sd, prefix="backbone.", precision=p) for p in ("fp32", "fp16", "bf16")}
for n in (20, 80, 640):
    ref = towers["fp32"].forward(frames[:n]).clone()
    for p in ("fp16", "bf16"):
        for tiles in (128, 10 ** 9):
            towers[p].IMPLICIT_MIN_TILES = tiles
            f = towers[p].forward(frames[:n]).clone()
            d = f - ref
            print("n=%4d %s implicit_min_tiles=%-10d feature rms err %.3e (rms %.3f)  err of the first 20 frames %.3e" % (
                n, p, tiles, float(d.pow(2).mean().sqrt()), float(ref.pow(2).mean().sqrt()), float(d[:20].pow(2).mean().sqrt())), flush=True)
if "--cpu" in sys.argv:
    import numerics_lab_rn as L
    import clipfsar_oracle as orc
    fr = frames[:20].cpu()
    ref = orc.resnet_forward(fr, sd, a)
    for sch in ("act=f16,w=f16", "act=bf16,w=bf16"):
        s = dict(kv.split("=") for kv in sch.split(","))
        d = L.make_tower(s)(fr, sd, a) - ref
        print("CPU model %-18s feature rms err %.3e" % (sch, float(d.pow(2).mean().sqrt())), flush=True)
