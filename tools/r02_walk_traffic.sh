for d in 0 256 2816 512 768 3584 3840 2048; do
  AB_VARIANT=26:$d PMC_GROUPS="4" PMC_MATCH=vit_gemm bash tools/pmc_gemm.sh walk_qkv_$d 252160 2304 768 > /dev/null 2>&1
  echo "== qkv dbg $d"; grep -E "RDREQ|GUI|HIT|MISS" gpurun_out/pmc_walk_qkv_$d/summary.txt
done
for d in 0 2816 3584; do
  AB_VARIANT=26:$d PMC_GROUPS="4" PMC_MATCH=vit_gemm bash tools/pmc_gemm.sh walk_fc_$d 252160 3072 768 gelu > /dev/null 2>&1
  echo "== fc dbg $d"; grep -E "RDREQ|GUI|HIT|MISS" gpurun_out/pmc_walk_fc_$d/summary.txt
done
