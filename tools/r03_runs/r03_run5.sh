mkdir -p gpurun_out/r3e
python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
for v in 26 30; do for d in 0 16; do TRACE_DBG=$d timeout 300 python tools/vit_trace.py 16 qkv $v 0 >> gpurun_out/r3e/trace.txt 2>&1; done; done
for d in 0 16; do TRACE_DBG=$d timeout 300 python tools/vit_trace.py 16 fc 30 0 >> gpurun_out/r3e/trace.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/r3e/trace.txt
