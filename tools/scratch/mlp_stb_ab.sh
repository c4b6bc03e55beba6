#!/bin/bash
# same-box A/B of the storing forward product's ring depth (mlp_gemm_hidden_stb): trees .ab_pfbase (stores between the
# barriers), .ab_pf6 / 7 / 8; tools/mlp_kernel_time.py prints forward / data-gradient / weight-gradient times per head
for i in 1 2; do
  for d in .ab_pfbase .ab_pf6 .ab_pf7 .ab_pf8; do
    echo "== $d"
    (cd $d && python tools/mlp_kernel_time.py 2>/dev/null | grep -E "forward")
  done
done
