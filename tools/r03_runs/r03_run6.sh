mkdir -p gpurun_out/r3f
python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
for v in 26 30; do TRACE_DBG=32 timeout 300 python tools/vit_trace.py 16 qkv $v 0 >> gpurun_out/r3f/trace.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/r3f/trace.txt
