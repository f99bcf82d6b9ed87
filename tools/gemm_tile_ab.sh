R8=r8.s1.qkv,r8.s1.proj,r8.s1.fc1,r8.s1.fc2,r8.s2.qkv,r8.s2.proj,r8.s2.fc1,r8.s2.fc2
for t in big mid small; do echo "== AURORA_GEMM_TILE=$t"; AURORA_GEMM_TILE=$t AURORA_GEMM_VARIANT=1 timeout 200 python tools/gemm_bench.py bf16 $R8,r8.s0.qkv,r8.s0.proj,r8.s0.fc1,r8.s0.fc2,s0.fc1,s1.fc1,sq8192; done
echo "== default dispatch"; timeout 200 python tools/gemm_bench.py bf16 $R8,r8.s0.qkv,r8.s0.proj,r8.s0.fc1,r8.s0.fc2
